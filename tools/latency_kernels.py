"""Single frame (BASELINE.json configs[1]): per-kernel GPU time from the library's HIP events, medians over 60 calls.
   run on the GPU box:  python tools/latency_kernels.py"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, torch
torch.cuda.init()
import pwpp_hip
h = pwpp_hip.Handle()
a = conftest.load_kitti(int(sys.argv[1]) if len(sys.argv) > 1 else 0); t = torch.from_numpy(a).cuda()
b = h.make_device_batch([t.data_ptr()], [a.shape[0]])
for i in range(10):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
ts = []
for i in range(60):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize(); ts.append(h.time_us())
print("whole call between events: median %.1f us  min %.1f us" % (sorted(ts)[30], min(ts)))
h.set_profiling(True); h.reset_kernel_profile()
for i in range(60):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
p = h.kernel_profile()
print({k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in p.items() if v[0] > 0}, "us per launch (with per-kernel events)")
