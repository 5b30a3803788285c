"""Single-frame latency of each KITTI sample frame (GPU time per call, median of 30) and the size of its largest patches."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, torch
torch.cuda.init()
import pwpp_hip
for k in range(6):
    a = conftest.load_kitti(k)
    t = torch.from_numpy(a).cuda()
    h = pwpp_hip.Handle()
    b = h.make_device_batch([t.data_ptr()], [a.shape[0]])
    ts = []
    for i in range(35):
        h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize(); ts.append(h.time_us())
    recs = h.patch_records(0)
    n = np.sort(recs["n_points"])[::-1]
    print("kitti %d: %6d points  %6.1f us   largest patches %s" % (k, a.shape[0], sorted(ts[5:])[15], n[:5].tolist()))
    h.close()
