"""ONE stateful stream in steady state (A-GLE histories full): GPU time of a frame on the handle's main stream (first kernel -> index lists
written, pwpp_get_time_us) and host time per launch + synchronize, with K5 as one kernel (split_k5 = 0), in two launches (1, the default for up to 64 streams: the
statistics over the histories run on the handle's second stream, under K6 and the host's turn-around) and with the second launch held back
until the lists are written (2).
   run on the GPU box:  python tools/stream_latency.py [steps] [fuse_scan]"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, torch
torch.cuda.init()
import pwpp_hip
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 240
src = [torch.from_numpy(conftest.load_kitti(i)).cuda() for i in range(6)]
FUSE = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for split in (0, 1, 2, 0, 1, 2):
    h = pwpp_hip.Handle()
    h.set_option("split_k5", split)
    h.set_option("fuse_scan", FUSE)
    h.set_num_streams(1)
    bs = [h.make_device_batch([s.data_ptr()], [s.shape[0]]) for s in src]
    for i in range(200):
        h.launch_device_batch(bs[i % 6], cols=4, mode=pwpp_hip.MODE_STREAMS); h.synchronize()
    gpu, t0 = [], time.perf_counter()
    for i in range(STEPS):
        h.launch_device_batch(bs[i % 6], cols=4, mode=pwpp_hip.MODE_STREAMS); h.synchronize()
        gpu.append(h.time_us())
    wall = (time.perf_counter() - t0) / STEPS * 1e6
    gpu.sort()
    print("split_k5 %d: GPU us per frame min / median / max %.1f / %.1f / %.1f, host us per launch + synchronize %.1f, history entries %s"
          % (split, gpu[0], gpu[len(gpu) // 2], gpu[-1], wall, [int(len(h.history(0, 0, r))) for r in range(4)]))
    h.close()
