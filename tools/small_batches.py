"""Fresh batches of 1..8 device-resident KITTI frames: GPU time per call (median of 40), default plans."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import conftest, torch
torch.cuda.init()
import pwpp_hip
src = [torch.from_numpy(conftest.load_kitti(i)).cuda() for i in range(6)]
import os
for F in [int(x) for x in os.environ.get('FRAMES', '1,2,3,4,5,6,8,12,16').split(',')]:
    h = pwpp_hip.Handle()
    b = h.make_device_batch([src[i % 6].data_ptr() for i in range(F)], [src[i % 6].shape[0] for i in range(F)])
    ts = []
    for i in range(45):
        h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize(); ts.append(h.time_us())
    m = sorted(ts[5:])[20]
    print("%2d frames: %7.1f us per call, %6.0f frames/s" % (F, m, F * 1e6 / m))
    h.close()
