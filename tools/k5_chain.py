"""Where the time of k_gle_tgr goes for a single fresh frame: timestamps of thread 0 along the kernel (debug_flags & 8).
Run on the GPU box: python tools/k5_chain.py"""
import sys, ctypes
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, torch
torch.cuda.init()
import pwpp_hip
h = pwpp_hip.Handle()
h.set_option("debug_flags", 8)
a = conftest.load_kitti(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t = torch.from_numpy(a).cuda()
b = h.make_device_batch([t.data_ptr()], [a.shape[0]])
names = ["start", "next counters cleared", "records and counts in", "decisions, first scan, centres written", "pushes, ring statistics",
         "TGR, second scan", "list offsets written", "threshold statistics", "state written", "end"]
acc = None
for i in range(30):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
    out = (ctypes.c_ulonglong * 64)()
    h._L.pwpp_debug_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
    h._check(h._L.pwpp_debug_read(h._h, out))
    v = [out[32 + k] for k in range(len(names))]
    d = [(v[k] - v[k - 1]) / 100.0 for k in range(1, len(names))]
    if i >= 10:
        acc = d if acc is None else [x + y for x, y in zip(acc, d)]
for k in range(1, len(names)):
    print("%-42s +%6.2f us" % (names[k], acc[k - 1] / 20.0))
print("total %.2f us" % (sum(acc) / 20.0))
v = [out[16 + k] for k in range(5)]
print("k_czm_scan (last call):", " ".join("+%.2f" % ((v[k] - v[k - 1]) / 100.0) for k in range(1, 5)), "us  [counts+offsets | maxima, bin counts | bucket histogram | lists]")
v = [out[k] for k in range(5)]
print("k_czm_bin_scatter, workgroup 0 (last call):", " ".join("+%.2f" % ((v[k] - v[k - 1]) / 100.0) for k in range(1, 5)),
      "us  [tables in LDS | points in, codes, ranks | ranges reserved | stores issued]; last workgroup ends %.2f us after workgroup 0 starts; scan starts %.2f us after that" % ((out[8] - v[0]) / 100.0, (out[16] - out[8]) / 100.0))
