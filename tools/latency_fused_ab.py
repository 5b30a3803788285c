"""single fresh frame and one stateful stream: K2 fused into K1' (default for fewer than 8 frames) against the two kernels (debug_flags 128)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'patchwork-plusplus_amd/python', ''):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
torch.cuda.init()
import conftest, pwpp_hip
src = [torch.from_numpy(conftest.load_kitti(k)).cuda() for k in range(6)]
for flags in (0, 256, 0, 256):
    row = []
    for k in range(6):
        h = pwpp_hip.Handle()
        h.set_option("debug_flags", flags)
        b = h.make_device_batch([src[k].data_ptr()], [src[k].shape[0]])
        ts = []
        for i in range(45):
            h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize(); ts.append(h.time_us())
        row.append(sorted(ts[5:])[20])
        h.close()
    h = pwpp_hip.Handle(); h.set_option("debug_flags", flags); h.set_num_streams(1)
    bs = [h.make_device_batch([src[k].data_ptr()], [src[k].shape[0]]) for k in range(6)]
    ts = []
    for i in range(300):
        h.launch_device_batch(bs[i % 6], cols=4, mode=pwpp_hip.MODE_STREAMS); h.synchronize(); ts.append(h.time_us())
    st = sorted(ts[150:])[75]
    h.close()
    print("%s: fresh frames %s us (median %.1f); one stream in steady state %.1f us" % ("fused K1'+K2" if flags == 0 else "two kernels ", " ".join("%.1f" % t for t in row), sorted(row)[3], st))
