import sys, ctypes
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, pwpp_hip
import oracle_lib as ol
from test_gpu_parity import to_oracle_params, assert_frame_equal
oracle = ol.restatement()
kitti = [conftest.load_kitti(i) for i in range(6)]
p = pwpp_hip.default_params(); p.enable_RNR = 0
refs = [ol.Estimator(oracle, to_oracle_params(p), arith=ol.ARITH_FXP).run(k) for k in kitti]
h = pwpp_hip.Handle(p)
L = pwpp_hip.load()
for name, cols, layout, arrs in (
        ("row-major N x 3", 3, pwpp_hip.LAYOUT_ROW_MAJOR, [np.ascontiguousarray(k[:, :3]) for k in kitti]),
        ("col-major N x 4", 4, 1, [np.asfortranarray(k) for k in kitti]),
        ("col-major N x 3", 3, 1, [np.asfortranarray(k[:, :3]) for k in kitti])):
    ptrs = (ctypes.c_void_p * 6)(*[a.ctypes.data for a in arrs])
    ns = (ctypes.c_int32 * 6)(*[a.shape[0] for a in arrs])
    h._check(L.pwpp_estimate_ground_batch(h._h, ptrs, ns, 6, cols, layout, pwpp_hip.MEM_HOST, pwpp_hip.MODE_FRESH))
    for i in range(6):
        assert_frame_equal(h, i, refs[i], kitti[i].shape[0])
    print(name, "batch of 6: bit-exact; one-pass/redone", h.one_pass_stats())
