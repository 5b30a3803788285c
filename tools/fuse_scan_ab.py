"""Single fresh frame with the part scan inside the binning kernel (fuse_scan = 1, rounds 4-5's default) and as a kernel of its own (0):
GPU us per KITTI sample, the two handles measured alternately, median of 60 calls each.   run on the GPU box: python tools/fuse_scan_ab.py"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, torch
torch.cuda.init()
import pwpp_hip
hs = []
for fz in (0, 1):
    h = pwpp_hip.Handle(); h.set_option("fuse_scan", fz); hs.append(h)
for k in range(6):
    a = conftest.load_kitti(k)
    t = torch.from_numpy(a).cuda()
    bs = [h.make_device_batch([t.data_ptr()], [a.shape[0]]) for h in hs]
    ts = [[], []]
    for i in range(70):
        for v in (0, 1):
            hs[v].launch_device_batch(bs[v], cols=4, mode=pwpp_hip.MODE_FRESH); hs[v].synchronize()
            if i >= 10: ts[v].append(hs[v].time_us())
    print("kitti %d: own kernel %6.1f us   inside the binning kernel %6.1f us" % (k, sorted(ts[0])[30], sorted(ts[1])[30]))
