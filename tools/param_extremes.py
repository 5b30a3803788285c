"""Parameter sets at the edges of what the C-ABI accepts, as single frames and as batches, against the
oracle (bit-exact).  run on the GPU box:  python tools/param_extremes.py"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest, pwpp_hip, pwpp_synth
import oracle_lib as ol
from test_gpu_parity import to_oracle_params, assert_frame_equal
oracle = ol.restatement()
kitti = [conftest.load_kitti(i) for i in range(6)]
syn = pwpp_synth.add_edge_cases(pwpp_synth.make_cloud(5, beams=48, azimuth_steps=1500), 5)
variants = [
    dict(sectors=(128, 128, 128, 128), rings=(4, 4, 4, 4)),          # 2048 bins = PWPP_MAX_BINS
    dict(sectors=(1, 1, 1, 1), rings=(1, 1, 1, 1)),                  # 4 bins of 10-80 k points each
    dict(sectors=(2, 3, 5, 7), rings=(1, 2, 1, 3)),
    dict(max_range=200.0, min_range=0.3),                            # fixed-point shift 15
    dict(max_range=20.0, min_range=5.0),                             # most points out of range
    dict(num_lpr=64, num_iter=7, th_seeds=0.02, th_dist=0.02),
    dict(num_min_pts=0, enable_TGR=0), dict(num_min_pts=5000),
    dict(uprightness_thr=0.0), dict(uprightness_thr=1.0),
    dict(RNR_ver_angle_thr=10.0, RNR_intensity_thr=2.0),             # RNR removes almost everything below the sensor
    dict(sensor_height=0.1, adaptive_seed_selection_margin=-50.0), dict(adaptive_seed_selection_margin=1.2),
]
for v in variants:
    p = pwpp_hip.default_params()
    for k, val in v.items():
        if k == "sectors":
            for i in range(4): p.num_sectors_each_zone[i] = val[i]
        elif k == "rings":
            for i in range(4): p.num_rings_each_zone[i] = val[i]
        else:
            setattr(p, k, val)
    try:
        h = pwpp_hip.Handle(p)
    except pwpp_hip.PwppError as e:
        print(v, "-> rejected:", str(e)[:90]); continue
    frames = [kitti[1], syn, kitti[3], kitti[0], kitti[5], syn, kitti[2]]
    ok = True
    try:
        h.estimate_ground_batch([kitti[4]], mode=pwpp_hip.MODE_FRESH)
        assert_frame_equal(h, 0, ol.Estimator(oracle, to_oracle_params(p), arith=ol.ARITH_FXP).run(kitti[4]), kitti[4].shape[0])
        h.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
        for i, pts in enumerate(frames):
            assert_frame_equal(h, i, ol.Estimator(oracle, to_oracle_params(p), arith=ol.ARITH_FXP).run(pts), pts.shape[0])
    except AssertionError as e:
        ok = False; print("   MISMATCH:", str(e)[:120])
    print(v, "->", "bit-exact" if ok else "FAIL", "| one-pass/redone", h.one_pass_stats(), "| ground/nonground/patches of frame 0:", h.counts(0))
