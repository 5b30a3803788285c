"""Where the time of the single-frame fit stage goes: stage-by-stage timestamps of the LARGEST patch
(k_fit_brows probes, PWPP_DEBUG_FLAGS=4).  codes: 1 start, 2 LPR done, 3 points pass done,
4 totals known, 5 plane(s) solved, 6 stage closed (strip included).  Run: PWPP_DEBUG_FLAGS=4 python tools/brows_chain.py"""
import sys, ctypes
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, torch
torch.cuda.init()
import pwpp_hip
h = pwpp_hip.Handle()
a = conftest.load_kitti(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t = torch.from_numpy(a).cuda()
b = h.make_device_batch([t.data_ptr()], [a.shape[0]])
for i in range(20):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
out = (ctypes.c_ulonglong * 64)()
h._L.pwpp_debug_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
h._check(h._L.pwpp_debug_read(h._h, out))
names = {1: "start", 2: "lpr", 3: "pass", 4: "totals", 5: "solve", 6: "stage end"}
prev = None
print("largest patch: %d points" % out[62])
first = out[0] & ((1 << 56) - 1)
print("first workgroup started %.2f us before the largest patch's; last workgroup (a patch of %d points) ended %.2f us after that start" % ((first - out[60]) / 100.0, out[61] & 0xFFFFFF, ((out[61] >> 24) - (first & ((1 << 40) - 1))) / 100.0))
for v in list(out)[:62]:
    if v == 0: break
    code, tick = v >> 56, v & ((1 << 56) - 1)
    print("%-10s +%6.2f us" % (names.get(code, code), 0.0 if prev is None else (tick - prev) / 100.0))
    prev = tick
