"""Fit plans against batch size in ONE process: GPU time per call (median of 15 after 4 warm-up calls) of fresh batches of
device-resident KITTI frames under every plan given.   FRAMES=32,64,... python tools/plans_by_frames.py "<plan>" "<plan>" ...
("" = the library's default plan for that batch size; distinct frame buffers, as bench.py)"""
import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, torch
torch.cuda.init()
import pwpp_hip
src = [conftest.load_kitti(i) for i in range(6)]
frames = [int(x) for x in os.environ.get('FRAMES', '32,64,128,256,512').split(',')]
plans = sys.argv[1:] or [""]
Fm = max(frames)
ns = [src[i % 6].shape[0] for i in range(Fm)]
offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
big = torch.empty((int(offs[-1]), 4), dtype=torch.float32, device="cuda")
sd = [torch.from_numpy(s).cuda() for s in src]
for i in range(Fm):
    big[offs[i]:offs[i + 1]].copy_(sd[i % 6])
torch.cuda.synchronize()
ptrs = [big.data_ptr() + int(offs[i]) * 16 for i in range(Fm)]
for F in frames:
    line = []
    for plan in plans:
        h = pwpp_hip.Handle()
        if plan:
            h.set_option("fit_plan", plan)
        b = h.make_device_batch(ptrs[:F], ns[:F])
        ts = []
        for i in range(19):
            h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize(); ts.append(h.time_us())
        m = sorted(ts[4:])[7]
        line.append("%-26s %8.1f us %7.0f f/s" % (plan or "(default)", m, F * 1e6 / m))
        h.close()
    print("%4d frames | " % F + " | ".join(line), flush=True)
