"""Why did bench.py's `synchronous` leg read 3.18 ms after a pipelined timed region?  sync -> pipelined -> sync on the same handles."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python'); sys.path.insert(0, '.')
import numpy as np, torch
torch.cuda.init()
import bench, pwpp_hip
dev = torch.device("cuda", 0)
src, _ = bench.load_source_frames("kitti")
F = 1024
ns = [src[i % 6].shape[0] for i in range(F)]
offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
sd = [torch.from_numpy(a).to(dev) for a in src]
def make_input():
    big = torch.empty((int(offs[-1]), 4), dtype=torch.float32, device=dev)
    for i in range(F):
        big[offs[i]:offs[i + 1]].copy_(sd[i % 6])
    return big
ins = [make_input() for _ in range(2)]
hs = [pwpp_hip.Handle() for _ in range(2)]
bs = [hs[d].make_device_batch([ins[d].data_ptr() + int(offs[i]) * 16 for i in range(F)], ns) for d in range(2)]
def sync_steps(n, tag):
    hs[0].set_overlap(True)
    for _ in range(3):
        hs[0].launch_device_batch(bs[0], cols=4, mode=pwpp_hip.MODE_FRESH); hs[0].synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        hs[0].launch_device_batch(bs[0], cols=4, mode=pwpp_hip.MODE_FRESH); hs[0].synchronize()
    print("%-40s %.3f ms" % (tag, 1e3 * (time.perf_counter() - t0) / n))
def piped(n, tag):
    for h in hs: h.set_overlap(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        d = k % 2
        if k >= 2: hs[d].synchronize()
        hs[d].launch_device_batch(bs[d], cols=4, mode=pwpp_hip.MODE_FRESH)
    for h in hs: h.synchronize()
    print("%-40s %.3f ms" % (tag, 1e3 * (time.perf_counter() - t0) / n))
sync_steps(10, "sync, fresh handle")
piped(10, "pipelined (first: allocations)")
piped(60, "pipelined")
sync_steps(10, "sync right after")
sync_steps(10, "sync again")
hs[1].trim_workspace()
sync_steps(10, "sync after trimming the other handle")
time.sleep(1.0)
sync_steps(10, "sync after 1 s idle")
