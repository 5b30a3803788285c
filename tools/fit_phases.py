"""Where the wave-time of the two big-batch fit kernels goes: shader-clock cycles per phase of k_fit_w64's loop, summed over all waves
(build with -DPWPP_PHASE_PROBE: tools/ab_build.sh probe -DPWPP_PHASE_PROBE; run: PWPP_LIB_PATH=ab/probe.so python tools/fit_phases.py [frames] [exact_moments])."""
import sys, ctypes
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, torch
torch.cuda.init()
import pwpp_hip
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h = pwpp_hip.Handle()
if len(sys.argv) > 2:
    h.set_option("exact_moments", sys.argv[2])
src = [conftest.load_kitti(i) for i in range(6)]
dev = [torch.from_numpy(a).cuda() for a in src]
b = h.make_device_batch([dev[i % 6].data_ptr() for i in range(F)], [src[i % 6].shape[0] for i in range(F)])
for i in range(3):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
h.set_option("debug_flags", "4")
h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
out = (ctypes.c_ulonglong * 64)()
h._L.pwpp_debug_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
h._check(h._L.pwpp_debug_read(h._h, out))
names = ["set-up", "lowest points", "publish", "points phase", "solve", "R-VPF strip", "state step"]
for base, kern in ((16, "k_fit_w64<16,64>"), (24, "k_fit_w64<64,p>")):
    tot = sum(out[base + k] for k in range(7))
    waves = out[base + 16]
    print("%s: %d waves, %.0f cycles per wave" % (kern, waves, tot / max(waves, 1)))
    for k in range(7):
        print("   %-14s %5.1f %%  (%8.0f cycles per wave)" % (names[k], 100.0 * out[base + k] / max(tot, 1), out[base + k] / max(waves, 1)))
for base, kern in ((56, "k_fit_w64<16,64>"), (48, "k_fit_w64<64,p>")):
    v = [out[base + k] / max(out[32 + (8 if base == 48 else 0)], 1) for k in range(5)]
    print("%s solve phase in detail (lane 0's view): counts / tiny fits %.0f, totals -> mean, covariance %.0f, Jacobi + plane %.0f, rest %.0f cycles per wave"
          % (kern, v[0] + v[1], v[2], v[3], v[4]))
