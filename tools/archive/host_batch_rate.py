import sys, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'patchwork-plusplus_amd/python')
import numpy as np, conftest, pwpp_hip
src=[conftest.load_kitti(i) for i in range(6)]
h=pwpp_hip.Handle()
for pinned in (False, True):
    for n in (64,256):
        if pinned:
            frames=[]
            for i in range(n):
                a=pwpp_hip.pinned_empty(src[i%6].shape); a[:]=src[i%6]; frames.append(a)
        else:
            frames=[src[i%6].copy() for i in range(n)]
        h.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
        best=1e9
        for rep in range(3):
            t0=time.perf_counter(); h.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH); best=min(best,time.perf_counter()-t0)
        out=pwpp_hip.pinned_empty((sum(f.shape[0] for f in frames),), np.int32) if pinned else None
        t0=time.perf_counter(); idx,base,counts=h.all_indices(out); d2h=time.perf_counter()-t0
        print("%s host memory, %3d frames: H2D+pipeline %.1f ms -> %.0f frames/s (%.1f GB/s in); all index lists D2H %.1f ms; end to end %.0f frames/s"%("pinned  " if pinned else "pageable",n,best*1e3,n/best,sum(f.nbytes for f in frames)/best/1e9,d2h*1e3,n/(best+d2h)))
        if pinned:
            for a in frames: pwpp_hip.pinned_free(a)
            pwpp_hip.pinned_free(out)
