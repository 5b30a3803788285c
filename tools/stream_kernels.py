"""One stateful stream in steady state (histories full): GPU time per call and per kernel (library events), medians."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import conftest, torch, time
torch.cuda.init()
import pwpp_hip
h = pwpp_hip.Handle()
h.set_num_streams(1)
src = [torch.from_numpy(conftest.load_kitti(i)).cuda() for i in range(6)]
bs = [h.make_device_batch([s.data_ptr()], [s.shape[0]]) for s in src]
ts, wall = [], []
for i in range(240):
    t0 = time.perf_counter()
    h.launch_device_batch(bs[i % 6], cols=4, mode=pwpp_hip.MODE_STREAMS); h.synchronize()
    wall.append(time.perf_counter() - t0); ts.append(h.time_us())
for k in range(6):
    sel = [ts[i] for i in range(120, 240) if i % 6 == k]
    print("kitti %d as a stream frame: %.1f us" % (k, sorted(sel)[len(sel) // 2]))
print("gpu_us per step: last 120 median %.1f; wall per call %.1f us" % (sorted(ts[-120:])[60], 1e6 * sorted(wall[-120:])[60]))
h.set_profiling(True); h.reset_kernel_profile()
for i in range(120):
    h.launch_device_batch(bs[0], cols=4, mode=pwpp_hip.MODE_STREAMS); h.synchronize()
p = h.kernel_profile()
print("frame 0 repeated, per kernel (us, events around every launch):", " ".join("%s=%.1f" % (k, 1000 * v[0] / max(v[1], 1)) for k, v in p.items() if v[1]))
