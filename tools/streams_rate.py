"""Stateful sequence mode: S long-lived streams stepped in lock-step, device-resident frames.
   run on the GPU box:  python tools/streams_rate.py"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, torch
torch.cuda.init()
import conftest, pwpp_hip
dev = torch.device("cuda", 0)
src = [torch.from_numpy(conftest.load_kitti(i)).to(dev) for i in range(6)]
for S in (1, 8, 64, 256):
    h = pwpp_hip.Handle()
    h.set_num_streams(S)
    # stream s sees the six frames in the order s, s+1, ...
    batches = []
    for t in range(6):
        ptrs = [src[(s + t) % 6].data_ptr() for s in range(S)]
        ns = [src[(s + t) % 6].shape[0] for s in range(S)]
        batches.append(h.make_device_batch(ptrs, ns))
    steps = 30
    for t in range(6):
        h.launch_device_batch(batches[t], cols=4, mode=pwpp_hip.MODE_STREAMS); h.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        h.launch_device_batch(batches[t % 6], cols=4, mode=pwpp_hip.MODE_STREAMS); h.synchronize()
    dt = time.perf_counter() - t0
    print("%4d streams: %.3f ms per lock-step (one frame per stream), %.0f frames/s, sensor height of stream 0 now %.4f" % (S, 1e3 * dt / steps, S * steps / dt, h.state(0).sensor_height))
    h.close()
