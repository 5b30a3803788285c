"""Double-buffered ingest: frames in page-locked host memory, two handles = two HIP
streams; while handle A's index lists travel device-to-host, handle B's frames travel host-to-device
and its kernels run.  Compared with the same work on one handle, one chunk after the other.
   run on the GPU box:  python tools/ingest_pipeline.py [frames per chunk] [chunks]"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, torch
torch.cuda.init()
import conftest, pwpp_hip

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
src = [conftest.load_kitti(i) for i in range(6)]   # stands for np.fromfile(path, np.float32).reshape(-1, 4)
bufs, outs = [], []
SLAB = len(sys.argv) > 3 and sys.argv[3] == "slab"   # frames of a chunk back to back in ONE pinned slab
for b in range(2):
    fr = []
    if SLAB:
        rows = sum(src[i % 6].shape[0] for i in range(C))
        slab = pwpp_hip.pinned_empty((rows, 4)); at = 0
        for i in range(C):
            n = src[i % 6].shape[0]; slab[at:at + n] = src[i % 6]; fr.append(slab[at:at + n]); at += n
    else:
        for i in range(C):
            a = pwpp_hip.pinned_empty(src[i % 6].shape); a[:] = src[i % 6]; fr.append(a)
    bufs.append(fr)
    outs.append(pwpp_hip.pinned_empty((sum(f.shape[0] for f in fr),), np.int32))
mb_in = sum(f.nbytes for f in bufs[0]) / 1e6

def run(overlap):
    H = [pwpp_hip.Handle(), pwpp_hip.Handle()] if overlap else [pwpp_hip.Handle()]
    for h in H:   # warm-up: allocations
        h.submit_pinned_batch(bufs[0]); h.all_indices(outs[0])
    t0 = time.perf_counter()
    checksum = 0
    for k in range(K):
        h = H[k % len(H)]
        if overlap:
            h.submit_pinned_batch(bufs[k % 2])
            if k > 0:
                idx, base, counts = H[(k - 1) % 2].all_indices(outs[(k - 1) % 2]); checksum += int(counts[:, 0].sum())
        else:
            h.submit_pinned_batch(bufs[k % 2])
            idx, base, counts = h.all_indices(outs[k % 2]); checksum += int(counts[:, 0].sum())
    if overlap:
        idx, base, counts = H[(K - 1) % 2].all_indices(outs[(K - 1) % 2]); checksum += int(counts[:, 0].sum())
    dt = time.perf_counter() - t0
    for h in H: h.close()
    return dt, checksum

for overlap in (False, True, False, True):
    dt, cs = run(overlap)
    print("%s: %d chunks x %d frames in %.1f ms -> %.0f frames/s end to end (%.1f GB/s in), ground points %d" %
          ("two handles, overlapped" if overlap else "one handle, serial     ", K, C, dt * 1e3, K * C / dt, K * mb_in / 1e3 / dt, cs))
