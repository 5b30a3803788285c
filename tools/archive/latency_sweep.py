import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'patchwork-plusplus_amd/python')
import numpy as np, conftest
import torch
torch.cuda.init()
plans=sys.argv[1:]
for plan in plans:
    os.environ["PWPP_FIT_PLAN"]=plan
    import importlib, pwpp_hip
    h=pwpp_hip.Handle()
    a=conftest.load_kitti(0)
    t=torch.from_numpy(a).cuda()
    b=h.make_device_batch([t.data_ptr()],[a.shape[0]])
    ts=[]
    for i in range(40):
        h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize(); ts.append(h.time_us())
    h.set_profiling(True); h.reset_kernel_profile()
    for i in range(20):
        h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
    prof=h.kernel_profile()
    print(plan, "median_us=%.1f"%sorted(ts)[len(ts)//2], {k:round(1000*v[0]/max(v[1],1)) for k,v in prof.items()})
    h.close()
