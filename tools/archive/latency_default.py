import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'patchwork-plusplus_amd/python')
import numpy as np, conftest, torch
torch.cuda.init()
import pwpp_hip
h=pwpp_hip.Handle()
for k in (0,3):
    a=conftest.load_kitti(k); t=torch.from_numpy(a).cuda()
    b=h.make_device_batch([t.data_ptr()],[a.shape[0]])
    ts=[]
    for i in range(60):
        h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize(); ts.append(h.time_us())
    print("frame",k,"device-resident single frame median gpu_us=%.1f min=%.1f"%(sorted(ts)[len(ts)//2],min(ts)))
    ts=[]
    import time
    for i in range(30):
        t0=time.perf_counter(); h.estimate_ground(a); ts.append((time.perf_counter()-t0)*1e6)
    print("   host path (H2D + pipeline + sync) median wall_us=%.1f"%sorted(ts)[len(ts)//2])
