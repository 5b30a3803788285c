"""kernel times of the reference-order mode on a 1024-frame batch (run under rocprofv3 --kernel-trace --stats)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'patchwork-plusplus_amd/python', ''):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
torch.cuda.init()
import bench, pwpp_hip
dev = torch.device("cuda", 0)
src, _ = bench.load_source_frames("kitti")
F = 1024
ns = [src[i % 6].shape[0] for i in range(F)]
offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
big = torch.empty((int(offs[-1]), 4), dtype=torch.float32, device=dev)
sd = [torch.from_numpy(a).to(dev) for a in src]
for i in range(F):
    big[offs[i]:offs[i + 1]].copy_(sd[i % 6])
torch.cuda.synchronize()
h = pwpp_hip.Handle()
h.set_output_order(True)
b = h.make_device_batch([big.data_ptr() + int(offs[i]) * 16 for i in range(F)], ns)
for _ in range(8):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
