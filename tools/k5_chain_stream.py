"""k_gle_tgr's chain (debug_flags 8 probes) for ONE STATEFUL STREAM in steady state (histories full), cf. tools/k5_chain.py."""
import sys, ctypes
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, torch
torch.cuda.init()
import pwpp_hip
h = pwpp_hip.Handle()
h.set_option("debug_flags", 8)
h.set_num_streams(1)
src = [torch.from_numpy(conftest.load_kitti(i)).cuda() for i in range(6)]
bs = [h.make_device_batch([s.data_ptr()], [s.shape[0]]) for s in src]
names = ["start", "next counters cleared", "records and counts in", "decisions, first scan, centres written", "pushes, ring statistics",
         "TGR, second scan", "list offsets written", "threshold statistics", "state written", "end"]
acc = None
N = 0
for i in range(260):
    h.launch_device_batch(bs[i % 6], cols=4, mode=pwpp_hip.MODE_STREAMS); h.synchronize()
    if i < 200:
        continue
    out = (ctypes.c_ulonglong * 64)()
    h._L.pwpp_debug_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
    h._check(h._L.pwpp_debug_read(h._h, out))
    v = [out[32 + k] for k in range(len(names))]
    d = [(v[k] - v[k - 1]) / 100.0 for k in range(1, len(names))]
    acc = d if acc is None else [x + y for x, y in zip(acc, d)]
    N += 1
for k in range(1, len(names)):
    print("%-42s +%6.2f us" % (names[k], acc[k - 1] / N))
print("total %.2f us" % (sum(acc) / N))
