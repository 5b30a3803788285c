"""Contract v3: a fit set of ONE, TWO or THREE points follows the reference's own
float arithmetic (patchworkpp.cpp:56-60) -- there Eigen's sums are determinate (fewer elements than a SIMD packet are
reduced sequentially in storage order, two terms commute), so nothing about those planes is "degenerate": all three
builds of the reference agree on them, and the product must agree with them.

CPU part: the restatement's product flavour (ARITH_FXP, which the HIP path equals bit for bit) against the reference
builds oracle/_ref/libpwpp_ref*.so -- the judge's table: parameter sets that let bins of 0-3 points through (the ROS
launch file among them), KITTI frames fresh and as a 12-frame sequence, and clouds with bins crafted to dwindle to 1-3 seeds.
GPU part (-m gpu): the HIP path on the same inputs, bit-identical to the restatement (every fit kernel) and identical
in index sets to the reference builds wherever those agree among themselves."""
import numpy as np
import pytest

import oracle_lib as ol

ROS_LAUNCH = dict(sensor_height=1.88, num_iter=3, num_lpr=20, num_min_pts=0, th_seeds=0.3, th_dist=0.125, th_seeds_v=0.25,
                  th_dist_v=0.9, max_range=80.0, min_range=1.0, uprightness_thr=0.101, enable_RNR=0)
TINY_VARIANTS = [dict(num_min_pts=0), dict(num_min_pts=1), dict(num_min_pts=2), dict(num_min_pts=3), dict(num_lpr=1), dict(num_lpr=2),
                 dict(num_lpr=3, num_min_pts=1), dict(th_seeds=0.02, th_seeds_v=0.02, num_min_pts=2), ROS_LAUNCH]
REF_FLAVOURS = (ol.ARITH_EIGEN_F32, ol.ARITH_EXACT_F64, ol.ARITH_F32_PACKET4)


def oparams(lib, variant):
    p = lib.default_params()
    for k, v in variant.items():
        setattr(p, k, v)
    return p


def tiny_seed_cloud(base, seed, sizes=(12, 40, 300, 1500, 6000), sensor_height=1.723):
    """`base` with a few CZM bins rebuilt so that their seed sets are 1, 2 or 3 points: k low points near the ground and M
    points 3 m higher -- the lowest-point representative (mean of the num_lpr = 20 lowest, ref :99-103) then lies so
    high that only the k low points are seeds, and the plane of the first fit comes from k points.  Bins in zone 0
    (R-VPF runs first) and zone 1, M from a dozen to thousands (every fit kernel's size class), some low points with
    EQUAL heights (the order among them is part of the contract: cloud order)."""
    rng = np.random.default_rng(seed)
    pts = base.copy()
    r = np.hypot(pts[:, 0], pts[:, 1])
    a = np.degrees(np.arctan2(pts[:, 1], pts[:, 0])) % 360.0
    # default CZM (patchworkpp.h:122-134): zone 0 = r in [2.7, 12.3625), 2 rings x 16 sectors; zone 1 = [12.3625, 22.025), 4 rings x 32 sectors
    wedges = []
    for s, m in zip(rng.permutation(16)[:len(sizes)], sizes):
        wedges.append((3.0, 7.0, 22.5 * s + 1.0, 22.5 * s + 21.5, m))
    for s, m in zip(rng.permutation(32)[:len(sizes)], sizes):
        wedges.append((12.6, 14.5, 11.25 * s + 0.5, 11.25 * s + 10.75, m))
    keep = np.ones(len(pts), bool)
    add = []
    for n, (r0, r1, a0, a1, m) in enumerate(wedges):
        keep &= ~((r >= r0 - 0.4) & (r < r1 + 0.6) & (a >= a0 - 1.0) & (a < a1 + 1.0))
        k = 1 + n % 3
        rr = rng.uniform(r0, r1, m + k)
        aa = np.radians(rng.uniform(a0, a1, m + k))
        z = rng.uniform(1.3, 1.8, m + k)
        z[:k] = -sensor_height + rng.uniform(-0.05, 0.05, k)
        if k > 1 and n % 2:
            z[1:k] = z[0]  # equal heights among the seeds
        w = np.stack([rr * np.cos(aa), rr * np.sin(aa), z, rng.uniform(0.0, 1.0, m + k)], 1).astype(np.float32)
        add.append(w[rng.permutation(m + k)])
    out = np.concatenate([pts[keep]] + add, 0)
    return np.ascontiguousarray(out[rng.permutation(len(out))])


def cpu_inputs(kitti):
    for v in TINY_VARIANTS:
        n3 = v is ROS_LAUNCH
        for k in (0, 3):
            yield v, (np.ascontiguousarray(kitti[k][:, :3]) if n3 else kitti[k])
    for sd in (1, 2):
        yield {}, tiny_seed_cloud(kitti[sd], sd)
        yield dict(num_min_pts=1), tiny_seed_cloud(kitti[sd + 2], 10 + sd, sizes=(0, 1, 2, 5, 9))
        yield dict(enable_RVPF=0), tiny_seed_cloud(kitti[sd], 20 + sd)


def ref_consensus(variant, pts):
    """ground index sets of the three reference builds; None when oracle/_ref is not there"""
    out = []
    for a in REF_FLAVOURS:
        lib = ol.reference(a)
        if lib is None:
            return None
        out.append(np.sort(ol.Estimator(lib, oparams(lib, variant), arith=a).run(pts).ground_idx))
    return out


def test_contract_follows_the_reference_builds_on_tiny_fit_sets(oracle_built, kitti):
    """The judge's table ("What's weak" 1) must read 0: wherever the three builds of the reference agree among
    themselves, the product's contract gives the same ground set -- num_lpr = 1 included."""
    lib = oracle_built.restatement()
    if ol.reference() is None:
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    agreed = 0
    for v, pts in cpu_inputs(kitti):
        refs = ref_consensus(v, pts)
        fx = np.sort(ol.Estimator(lib, oparams(lib, v), arith=ol.ARITH_FXP).run(pts).ground_idx)
        agreed += all(np.array_equal(refs[0], r) for r in refs[1:])
        # (on these inputs the contract equals the float build even in the five cases where the reference's exact-f64
        # flavour does not: num_lpr <= 3 on KITTI frame 0, the ROS set on frame 3, one crafted cloud)
        assert np.array_equal(fx, refs[0]), (v, len(np.setxor1d(fx, refs[0])))
    assert agreed >= 18


def test_ros_launch_sequence_follows_the_reference_build(oracle_built, kitti):
    """ros/launch/patchworkpp.launch.py:50-64 on ONE object over twelve N x 3 frames: identical ground sets
    to the reference build (float flavour) in every frame where the reference's exact flavour follows it too."""
    ref, refx = ol.reference(ol.ARITH_EIGEN_F32), ol.reference(ol.ARITH_EXACT_F64)
    if ref is None or refx is None:
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    lib = oracle_built.restatement()
    e_ref = ol.Estimator(ref, oparams(ref, ROS_LAUNCH), arith=ol.ARITH_EIGEN_F32)
    e_refx = ol.Estimator(refx, oparams(refx, ROS_LAUNCH), arith=ol.ARITH_EXACT_F64)
    e_fx = ol.Estimator(lib, oparams(lib, ROS_LAUNCH), arith=ol.ARITH_FXP)
    followed = 0
    for k in range(12):
        pts = np.ascontiguousarray(kitti[k % 6][:, :3])
        a, b, c = e_ref.run(pts), e_refx.run(pts), e_fx.run(pts)
        if len(np.setxor1d(a.ground_idx, b.ground_idx)) == 0:
            followed += 1
            assert len(np.setxor1d(a.ground_idx, c.ground_idx)) == 0, k
    assert followed >= 8


def test_tiny_sets_order_is_height_then_cloud_index(oracle_built):
    """Three seeds, two of them at the same height: the float sums depend on which of the two comes second, and the contract
    says cloud order.  Shuffling the cloud (indices change, heights do not) must therefore be able to change the last
    bits of the plane but never what the restatement and the reference build (stable insertion sort below 17 points,
    libstdc++) make of a small bin."""
    ref = ol.reference(ol.ARITH_EIGEN_F32)
    if ref is None:
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    lib = oracle_built.restatement()
    rng = np.random.default_rng(5)
    v = dict(num_min_pts=3, enable_RNR=0)
    final_tiny = 0
    for _ in range(200):
        k = int(rng.integers(3, 9))
        rr, aa = rng.uniform(3.0, 7.0, k), rng.uniform(0.05, 0.35, k)
        z = np.full(k, -1.7, np.float32) + rng.uniform(0, 0.1, k).astype(np.float32)
        z[rng.integers(0, k)] = z[rng.integers(0, k)]
        z[3:] += 20.0
        pts = np.stack([rr * np.cos(aa), rr * np.sin(aa), z, np.zeros(k)], 1).astype(np.float32)
        a = ol.Estimator(ref, oparams(ref, v), arith=ol.ARITH_EIGEN_F32).run(pts)
        c = ol.Estimator(lib, oparams(lib, v), arith=ol.ARITH_FXP).run(pts)
        assert np.array_equal(np.sort(a.ground_idx), np.sort(c.ground_idx))
        if len(c.records) == 1 and 1 <= c.records["n_ground"][0] <= 3:  # the final plane is a tiny fit: bit for bit the reference's
            final_tiny += 1
            assert np.array_equal(a.normals, c.normals, equal_nan=True) and np.array_equal(a.centers, c.centers, equal_nan=True)
    assert final_tiny >= 100


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_tiny_fit_sets_in_every_fit_kernel(oracle_built, kitti):
    import pwpp_hip
    from test_gpu_parity import apply_variant, assert_frame_equal, to_oracle_params
    lib = oracle_built.restatement()
    cases = list(cpu_inputs(kitti))
    # one frame at a time (the single-frame kernel: four waves per big patch, one per small one) ...
    checked_ref = 0
    for v, pts in cases:
        p = apply_variant(pwpp_hip.default_params(), v)
        h = pwpp_hip.Handle(p)
        h.estimate_ground(pts)
        fx = ol.Estimator(lib, to_oracle_params(p), arith=ol.ARITH_FXP).run(pts)
        assert_frame_equal(h, 0, fx, pts.shape[0], state_index=0)
        refs = ref_consensus(v, pts)
        if refs is not None and all(np.array_equal(refs[0], r) for r in refs[1:]):
            assert np.array_equal(np.sort(h.ground_indices(0)), refs[0]), v
            checked_ref += 1
    assert checked_ref >= 18 or ol.reference() is None
    # ... and as batches under every plan: 16-lane rows / 64-lane rows with the solve per lane, one wave per patch,
    # four waves per patch, and the workgroup-per-patch kernel for everything above 255 points
    by_variant = {}
    for v, pts in cases:
        by_variant.setdefault(tuple(sorted(v.items())), (v, []))[1].append(pts)
    for v, frames in by_variant.values():
        p = apply_variant(pwpp_hip.default_params(), v)
        frames = (frames * 3)[:6]
        refs = [ol.Estimator(lib, to_oracle_params(p), arith=ol.ARITH_FXP).run(f) for f in frames]
        hb = pwpp_hip.Handle(p)
        for plan in ("W16:1023,W64.2:65535", "W16.16:1023,S64:65535", "W16.32:255,W64.4:65535", "S16:127,S64:65535", "B64:65535", "W16:255", "S8:63,S32:255"):
            hb.set_option("fit_plan", plan)
            hb.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
            for i, f in enumerate(frames):
                assert_frame_equal(hb, i, refs[i], f.shape[0])


@pytest.mark.gpu
def test_hip_ros_launch_sequence_follows_the_reference_build(oracle_built, kitti):
    """SURVEY 8f-f4 against the REFERENCE BUILD (not only the contract oracle): the launch file's parameter set, one stream
    over twelve N x 3 frames."""
    import pwpp_hip
    from test_gpu_parity import apply_variant
    ref, refx = ol.reference(ol.ARITH_EIGEN_F32), ol.reference(ol.ARITH_EXACT_F64)
    if ref is None or refx is None:
        pytest.skip("oracle/_ref did not travel")
    p = apply_variant(pwpp_hip.default_params(), ROS_LAUNCH)
    h = pwpp_hip.Handle(p)
    e_ref = ol.Estimator(ref, oparams(ref, ROS_LAUNCH), arith=ol.ARITH_EIGEN_F32)
    e_refx = ol.Estimator(refx, oparams(refx, ROS_LAUNCH), arith=ol.ARITH_EXACT_F64)
    followed = 0
    for k in range(12):
        pts = np.ascontiguousarray(kitti[k % 6][:, :3])
        a, b = e_ref.run(pts), e_refx.run(pts)
        h.estimate_ground(pts)
        if len(np.setxor1d(a.ground_idx, b.ground_idx)) == 0:
            followed += 1
            assert np.array_equal(np.sort(h.ground_indices(0)), np.sort(a.ground_idx)), k
            assert np.abs(h.normals(0) - a.normals)[np.isfinite(a.normals).all(1)].max() < 1e-4
    assert followed >= 8
