"""Throughput with batches in flight back to back: two (three) handles, each with its own 1024-frame batch and workspace; batch k + 1 is
enqueued before batch k is waited for, so the ramp-up of one batch (binning with no fits to overlap) runs under the ramp-down of the
one before (last fits, index lists).  Against the synchronous step of bench.py (launch, wait, launch, wait).
   run on the GPU box:  python tools/pipelined_batches.py"""
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
ORDERED = len(sys.argv) > 1 and sys.argv[1] == "--reference-order"   # PWPP_ORDER_REFERENCE: every sub-list in the reference's z-sorted order
for depth in (1, 2, 3):
    for overlap in ((False,) if ORDERED else (True, False)):
        ins = [make_input() for _ in range(depth)]
        hs = [pwpp_hip.Handle() for _ in range(depth)]
        for h in hs:
            h.set_overlap(overlap)
            h.set_output_order(ORDERED)
        bs = [hs[d].make_device_batch([ins[d].data_ptr() + int(offs[i]) * 16 for i in range(F)], ns) for d in range(depth)]
        torch.cuda.synchronize()
        def run(steps):
            for k in range(steps):
                d = k % depth
                if k >= depth: hs[d].synchronize()      # the batch this handle launched `depth` steps ago
                hs[d].launch_device_batch(bs[d], cols=4, mode=pwpp_hip.MODE_FRESH)
            for h in hs: h.synchronize()
        run(6)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        steps = 30
        run(steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        c = hs[0].all_counts()
        print("%d batch(es) in flight, overlap mode %s: %.3f ms per 1024-frame batch, %.0f frames/s  (ground points of frame 0: %d)" % (depth, "on " if overlap else "off", 1e3 * dt, F / dt, c[0, 0]))
        for h in hs: h.close()
        del ins, bs
