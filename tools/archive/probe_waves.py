import sys, os, ctypes
os.environ["PWPP_DEBUG_FLAGS"]="4"
sys.path.insert(0,'tests'); sys.path.insert(0,'patchwork-plusplus_amd/python')
import numpy as np, conftest, pwpp_hip
frames=[conftest.load_kitti(i%6) for i in range(256)]
h=pwpp_hip.Handle()
for rep in range(2):
    h.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
out=(ctypes.c_ulonglong*64)()
pwpp_hip.load().pwpp_debug_read(h._h, out)
for c in range(3):
    mx,sm,n=out[c*4],out[c*4+1],out[c*4+2]
    print("class",c,"waves",n,"max_us",mx/100.0,"avg_us",(sm/max(n,1))/100.0)
