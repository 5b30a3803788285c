"""single fresh frame (each KITTI sample) and one stateful stream under two builds of the library: PWPP_LIB_PATH selects the build"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'patchwork-plusplus_amd/python', ''):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
torch.cuda.init()
import conftest, pwpp_hip
src = [torch.from_numpy(conftest.load_kitti(k)).cuda() for k in range(6)]
row = []
for k in range(6):
    h = pwpp_hip.Handle()
    b = h.make_device_batch([src[k].data_ptr()], [src[k].shape[0]])
    ts = []
    for i in range(65):
        h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize(); ts.append(h.time_us())
    row.append(sorted(ts[5:])[30])
    h.close()
h = pwpp_hip.Handle(); h.set_num_streams(1)
bs = [h.make_device_batch([src[k].data_ptr()], [src[k].shape[0]]) for k in range(6)]
ts = []
for i in range(300):
    h.launch_device_batch(bs[i % 6], cols=4, mode=pwpp_hip.MODE_STREAMS); h.synchronize(); ts.append(h.time_us())
print("%-14s fresh frames %s us (median %.1f); one stream in steady state %.1f us" % (os.path.basename(os.environ.get("PWPP_LIB_PATH", "default")), " ".join("%.1f" % t for t in row), sorted(row)[3], sorted(ts[150:])[75]))
