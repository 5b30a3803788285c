import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'patchwork-plusplus_amd/python')
import pwpp_hip
from conftest import load_kitti
k=[load_kitti(i) for i in range(6)]
for F, scale, ov in ((256, 0.05, 0), (256, 0.05, 1), (1024, 0.05, 0), (1024, 0.05, 1)):
    h=pwpp_hip.Handle()
    h.set_overlap(bool(ov))
    h.set_option("one_pass_scale", scale)
    frames=[k[i%6] for i in range(F)]
    for rep in range(2):
        h.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
        c=h.all_counts()
        bad=[i for i in range(F) if tuple(c[i,:3])!=tuple(c[i%6,:3])]
        print(F, scale, 'overlap', ov, rep, h.one_pass_stats(), 'bad frames', len(bad), bad[:8], flush=True)
