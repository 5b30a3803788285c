import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, torch
torch.cuda.init()
import conftest, pwpp_hip
dev = torch.device("cuda", 0)
src = [conftest.load_kitti(i) for i in range(6)]
F = 1024
ns = [src[i % 6].shape[0] for i in range(F)]
offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
big = torch.empty((int(offs[-1]), 4), dtype=torch.float32, device=dev)
sd = [torch.from_numpy(s).to(dev) for s in src]
for i in range(F):
    big[offs[i]:offs[i + 1]].copy_(sd[i % 6])
torch.cuda.synchronize()
ptrs = [big.data_ptr() + int(offs[i]) * 16 for i in range(F)]
h = pwpp_hip.Handle()
b = h.make_device_batch(ptrs, ns)
for _ in range(2):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
h.set_profiling(True); h.reset_kernel_profile()
for _ in range(8):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
prof = h.kernel_profile()
print("flags=%s:" % os.environ.get("PWPP_DEBUG_FLAGS", "0"), {k: round(v[0] / max(v[1], 1), 3) for k, v in prof.items() if k.startswith("k_czm")})
