"""One stateful stream, device-resident frames, 260 steps (histories full after ~70): the steady-state
workload behind tools/trace_gaps.py (skip the first 150 runs)."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import conftest, torch
torch.cuda.init()
import pwpp_hip
h = pwpp_hip.Handle()
h.set_num_streams(1)
src = [torch.from_numpy(conftest.load_kitti(i)).cuda() for i in range(6)]
bs = [h.make_device_batch([s.data_ptr()], [s.shape[0]]) for s in src]
ts = []
for i in range(260):
    h.launch_device_batch(bs[i % 6], cols=4, mode=pwpp_hip.MODE_STREAMS); h.synchronize(); ts.append(h.time_us())
print("gpu_us per step: first 6 median %.1f, last 60 median %.1f" % (sorted(ts[:6])[3], sorted(ts[-60:])[30]))

