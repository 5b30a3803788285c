import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import conftest, torch
torch.cuda.init()
import pwpp_hip
F = int(os.environ.get('FRAMES', '3'))
src = [torch.from_numpy(conftest.load_kitti(i)).cuda() for i in range(6)]
h = pwpp_hip.Handle()
b = h.make_device_batch([src[i % 6].data_ptr() for i in range(F)], [src[i % 6].shape[0] for i in range(F)])
h.set_profiling(True)
for i in range(30):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
h.reset_kernel_profile()
for i in range(30):
    h.launch_device_batch(b, cols=4, mode=pwpp_hip.MODE_FRESH); h.synchronize()
p = h.kernel_profile()
print(F, "frames:", " ".join("%s=%.1f" % (k, 1000 * v[0] / max(v[1], 1)) for k, v in p.items() if v[1]))
