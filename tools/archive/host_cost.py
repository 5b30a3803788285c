import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, torch
torch.cuda.init()
import conftest, pwpp_hip
dev = torch.device("cuda", 0)
src = [torch.from_numpy(conftest.load_kitti(i)).to(dev) for i in range(6)]
for F in (1, 64, 1024):
    h = pwpp_hip.Handle()
    b = h.make_device_batch([src[i % 6].data_ptr() for i in range(F)], [src[i % 6].shape[0] for i in range(F)])
    for _ in range(3):
        h.launch_device_batch(b); h.synchronize()
    tl, ts = [], []
    for _ in range(20):
        t0 = time.perf_counter(); h.launch_device_batch(b); t1 = time.perf_counter(); h.synchronize(); t2 = time.perf_counter()
        tl.append(t1 - t0); ts.append(t2 - t0)
    print("F=%d: launch call returns after %.1f us (host work), whole step %.1f us, GPU %.1f us" % (F, 1e6 * sorted(tl)[10], 1e6 * sorted(ts)[10], h.time_us()))
