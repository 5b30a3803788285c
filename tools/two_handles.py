"""Does the chip overlap the memory-bound stages (K1', K6) of one batch with the VALU-bound fit kernels of
another?  1024 device-resident frames as one batch on one handle vs two half-batches on two handles
(own streams) launched back to back.  run on the GPU box: python tools/two_handles.py"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import conftest, torch
torch.cuda.init()
import pwpp_hip
src = [torch.from_numpy(conftest.load_kitti(i)).cuda() for i in range(6)]
F = 1024
big = []
for i in range(F):  # distinct buffers, as bench.py
    big.append(src[i % 6].clone())
torch.cuda.synchronize()
def batch(h, lo, hi):
    return h.make_device_batch([big[i].data_ptr() for i in range(lo, hi)], [big[i].shape[0] for i in range(lo, hi)])
for parts in (1, 0, 2, 4):  # 0 = one handle in overlap mode
    overlap = parts == 0
    parts = max(parts, 1)
    hs = [pwpp_hip.Handle() for _ in range(parts)]
    if overlap: hs[0].set_overlap(True)
    per = F // parts
    bs = [batch(hs[p], p * per, (p + 1) * per) for p in range(parts)]
    def step():
        for p in range(parts):
            hs[p].launch_device_batch(bs[p], cols=4, mode=pwpp_hip.MODE_FRESH)
        for p in range(parts):
            hs[p].synchronize()
    for _ in range(3): step()
    t0 = time.perf_counter()
    for _ in range(10): step()
    dt = (time.perf_counter() - t0) / 10
    print("%d handle(s) x %4d frames%s: %.3f ms per 1024 frames, %.0f frames/s" % (parts, per, " (overlap mode)" if overlap else "", dt * 1e3, F / dt))
    for h in hs: h.close()
