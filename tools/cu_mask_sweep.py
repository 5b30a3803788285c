"""the overlap schedule with the memory stream (K1', K2, K6) and the fit streams on DISJOINT sets of CUs
(hipExtStreamCreateWithCUMask, option "cu_split"), against the default (both kinds of stream on all 256 CUs).
   run on the GPU box:  python tools/cu_mask_sweep.py  > gpurun_out/r05_cu_mask_sweep.txt"""
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
big = torch.empty((int(offs[-1]), 4), dtype=torch.float32, device=dev)
sd = [torch.from_numpy(a).to(dev) for a in src]
for i in range(F):
    big[offs[i]:offs[i + 1]].copy_(sd[i % 6])
torch.cuda.synchronize()
h = pwpp_hip.Handle()
batch = h.make_device_batch([big.data_ptr() + int(offs[i]) * 16 for i in range(F)], ns)

def run(steps=20):
    for _ in range(3):
        h.launch_device_batch(batch, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        h.launch_device_batch(batch, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
    return (time.perf_counter() - t0) / steps

h.launch_device_batch(batch, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
ref = h.all_counts().copy()
base = run()
print("default (no CU masks, 2 ranges, 2 fit streams): %.3f ms per 1024-frame step, %.0f frames/s" % (1e3 * base, F / base))
best = (base, "default")
for ranges in (2, 4, 8):
    h.set_option("overlap_ranges", ranges)
    h.set_option("cu_split", "0")
    t = run()
    print("ranges %d, no masks                      : %.3f ms (%+.1f %%)" % (ranges, 1e3 * t, 100 * (base / t - 1)))
    for mode in (0, 1):
        for n in (32, 64, 96, 128, 160, 192):
            try:
                h.set_option("cu_split", "%d:%d" % (n, mode))
            except Exception as e:
                print("cu_split %d:%d refused: %s" % (n, mode, e)); continue
            t = run()
            ok = np.array_equal(h.all_counts(), ref)
            print("ranges %d, memory stream on %3d CUs mode %d : %.3f ms (%+.1f %% vs default)%s" % (ranges, n, mode, 1e3 * t, 100 * (base / t - 1), "" if ok else "  RESULTS DIFFER"))
            if t < best[0]: best = (t, "ranges %d cu_split %d:%d" % (ranges, n, mode))
h.set_option("cu_split", "0"); h.set_option("overlap_ranges", 2)
t = run()
print("default again: %.3f ms" % (1e3 * t))
print("best: %s %.3f ms (%+.1f %% vs default)" % (best[1], 1e3 * best[0], 100 * (base / best[0] - 1)))
