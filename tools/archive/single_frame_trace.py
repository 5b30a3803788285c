"""Single device-resident frame, 200 launches: the workload behind tools/trace_gaps.py."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import conftest, torch
torch.cuda.init()
import pwpp_hip
h = pwpp_hip.Handle()
a = conftest.load_kitti(0); t = torch.from_numpy(a).cuda()
b = h.make_device_batch([t.data_ptr()], [a.shape[0]])
for i in range(200):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
