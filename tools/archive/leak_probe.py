import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, torch
torch.cuda.init()
import conftest, pwpp_hip
k = [conftest.load_kitti(i) for i in range(6)]
def free_gb():
    f, t = torch.cuda.mem_get_info(); return f / 1e9
f0 = free_gb()
for r in range(60):
    if r == 30: print("   after 30 cycles: %.2f GB free" % free_gb())
    h = pwpp_hip.Handle()
    h.estimate_ground_batch([k[i % 6] for i in range(12)], mode=pwpp_hip.MODE_FRESH)
    h.set_num_streams(4)
    h.estimate_ground_batch(k[:4], mode=pwpp_hip.MODE_STREAMS)
    h.set_output_order(True)
    h.estimate_ground(k[0])
    _ = h.ground(0)
    h.close()
f1 = free_gb()
print("free HBM before %.2f GB, after 60 create/use/destroy cycles %.2f GB (delta %.3f GB)" % (f0, f1, f0 - f1))
