"""CPU checks of the checker itself (no GPU): the restatement against the committed golden
outputs of the reference build, against the reference build directly when it is present,
and the primitives of the arithmetic contract."""
import ctypes
import hashlib
import math

import numpy as np
import pytest

import oracle_lib as ol
from conftest import ground_mask

# flavours the reference build exists in (oracle/Makefile); the fixed-point contract is the restatement's alone
FLAVOURS = [(ol.ARITH_EIGEN_F32, "f32"), (ol.ARITH_EXACT_F64, "exact"), (ol.ARITH_F32_PACKET4, "pk4")]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_against_golden(res, n, golden, key):
    assert np.array_equal(np.packbits(ground_mask(res.ground_idx, n)), golden[key + "ground_mask"])
    assert [len(res.ground_idx), len(res.nonground_idx), len(res.centers)] == list(golden[key + "counts"])
    assert np.array_equal(res.centers, golden[key + "centers"], equal_nan=True)
    assert np.array_equal(res.normals, golden[key + "normals"], equal_nan=True)
    state = np.concatenate([[res.sensor_height], res.elevation_thr, res.flatness_thr])
    assert np.array_equal(state, golden[key + "state"])
    # the reference's own output ORDER, too
    assert sha(res.ground_idx) == str(golden[key + "sha_ground_order"])
    assert sha(res.nonground_idx) == str(golden[key + "sha_nonground_order"])


@pytest.mark.parametrize("arith,name", FLAVOURS)
def test_restatement_matches_golden_fresh(oracle_built, kitti, golden, arith, name):
    lib = oracle_built.restatement()
    for k, pts in enumerate(kitti):
        res = ol.Estimator(lib, arith=arith).run(pts)
        check_against_golden(res, pts.shape[0], golden, "%s/fresh/%d/" % (name, k))


@pytest.mark.parametrize("arith,name", FLAVOURS)
def test_restatement_matches_golden_sequence(oracle_built, kitti, golden, arith, name):
    est = ol.Estimator(oracle_built.restatement(), arith=arith)
    for k, pts in enumerate(kitti):
        check_against_golden(est.run(pts), pts.shape[0], golden, "%s/seq/%d/" % (name, k))


def test_fixture_md5(kitti, golden):
    for k, pts in enumerate(kitti):
        assert hashlib.md5(pts.tobytes()).hexdigest() == str(golden["md5"][k])


def test_survey_anchor_counts(golden):
    # fresh-state ground counts of the six sample frames
    anchors = [(72665, 52003, 274), (72500, 52105, 271), (71413, 53065, 264), (70560, 53607, 261),
               (69315, 54654, 254), (68068, 55856, 250)]
    for k, a in enumerate(anchors):
        assert tuple(golden["f32/fresh/%d/counts" % k]) == a
        assert tuple(golden["exact/fresh/%d/counts" % k]) == a
    seq = [72665, 71848, 71263, 70535, 69095, 67614]
    assert [int(golden["f32/seq/%d/counts" % k][0]) for k in range(6)] == seq


def test_float_flavours_agree_with_the_exact_arbiter_on_kitti(golden):
    """On the reference's own data the summation arithmetic changes no decision: the two float flavours of
    the reference build and its exact-f64 flavour give the same ground masks, normals within 1e-4."""
    for mode in ("fresh", "seq"):
        for k in range(6):
            for name in ("f32", "pk4"):
                a, b = "%s/%s/%d/" % (name, mode, k), "exact/%s/%d/" % (mode, k)
                assert np.array_equal(golden[a + "ground_mask"], golden[b + "ground_mask"])
                assert np.abs(golden[a + "normals"] - golden[b + "normals"]).max() < 1e-4
                assert np.abs(golden[a + "centers"] - golden[b + "centers"]).max() < 1e-4


def test_fixed_point_contract_vs_the_exact_arbiter_on_kitti(oracle_built, kitti, golden):
    """The product's arithmetic contract (restatement, fxp flavour) against the reference build in exact
    arithmetic (golden): identical ground masks, centres within 1e-6, normals within 3e-5 -- closer to exact
    arithmetic than the reference's own float sums (test above / tests/test_arith_flavours.py)."""
    lib = oracle_built.restatement()
    for mode in ("fresh", "seq"):
        est = ol.Estimator(lib, arith=ol.ARITH_FXP)
        for k, pts in enumerate(kitti):
            res = est.run(pts) if mode == "seq" else ol.Estimator(lib, arith=ol.ARITH_FXP).run(pts)
            key = "exact/%s/%d/" % (mode, k)
            assert np.array_equal(np.packbits(ground_mask(res.ground_idx, pts.shape[0])), golden[key + "ground_mask"])
            assert np.abs(res.centers - golden[key + "centers"]).max() < 1e-6
            assert np.abs(res.normals - golden[key + "normals"]).max() < 3e-5
            state = np.concatenate([[res.sensor_height], res.elevation_thr, res.flatness_thr])
            assert np.abs(state - golden[key + "state"]).max() < 1e-6


@pytest.mark.parametrize("arith,name", FLAVOURS)
def test_restatement_equals_reference_build_bitwise(oracle_built, kitti, arith, name):
    ref = oracle_built.reference(arith)
    if ref is None:
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    import pwpp_synth
    frames = [kitti[0], kitti[3], pwpp_synth.add_edge_cases(pwpp_synth.make_cloud(11, beams=32, azimuth_steps=1000), 11)]
    a, b = ol.Estimator(ref, arith=arith), ol.Estimator(oracle_built.restatement(), arith=arith)
    for pts in frames + frames:  # second pass exercises the adaptive state
        ra, rb = a.run(pts), b.run(pts)
        for fld in ("ground_idx", "nonground_idx", "ground", "nonground", "centers", "normals",
                    "elevation_thr", "flatness_thr"):
            assert np.array_equal(getattr(ra, fld), getattr(rb, fld), equal_nan=True), fld
        assert ra.sensor_height == rb.sensor_height
        for r in range(4):
            assert np.array_equal(ra.hist_elev[r], rb.hist_elev[r])
            assert np.array_equal(ra.hist_flat[r], rb.hist_flat[r])


def test_invariants_partition_and_normals(oracle_built, kitti):
    # properties derived from the reference code
    res = ol.Estimator(oracle_built.restatement(), arith=ol.ARITH_FXP).run(kitti[1])
    n = kitti[1].shape[0]
    allidx = np.concatenate([res.ground_idx, res.nonground_idx])
    assert len(allidx) == n and np.array_equal(np.sort(allidx), np.arange(n))
    assert len(res.centers) == len(res.normals) == len(res.records)
    assert (res.normals[:, 2] >= 0).all()
    assert np.allclose(np.linalg.norm(res.normals, axis=1), 1.0, atol=1e-5)
    assert np.array_equal(res.ground, kitti[1][res.ground_idx, :3])


def test_three_column_input_without_rnr(oracle_built, kitti):
    lib = oracle_built.restatement()
    p = lib.default_params()
    p.enable_RNR = 0
    a = ol.Estimator(lib, p, arith=ol.ARITH_FXP).run(kitti[2][:, :3].copy())
    b = ol.Estimator(lib, p, arith=ol.ARITH_FXP).run(kitti[2])
    assert np.array_equal(a.ground_idx, b.ground_idx)


def test_glibc_atan2_special_values():
    """The HIP kernel answers axis/diagonal directions with these constants (czm_atan2)."""
    def bits(x):
        return np.float64(x).view(np.uint64)
    assert bits(math.atan2(1.0, 1.0)) == 0x3FE921FB54442D18
    assert bits(math.atan2(3.7, 3.7)) == 0x3FE921FB54442D18
    assert bits(math.atan2(1.0, -1.0)) == 0x4002D97C7F3321D2
    assert bits(math.atan2(-1.0, -1.0)) == 0xC002D97C7F3321D2
    assert bits(math.atan2(0.0, -1.0)) == 0x400921FB54442D18
    assert bits(math.atan2(1.0, 0.0)) == 0x3FF921FB54442D18
    assert math.atan2(0.0, 1.0) == 0.0 and math.copysign(1, math.atan2(-0.0, 1.0)) == -1


def test_jacobi_is_an_eigen_decomposition(oracle_built):
    oracle_built.restatement()
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.normal(size=(rng.integers(3, 50), 3)).astype(np.float32) * rng.uniform(0.01, 5, 3).astype(np.float32)
        cov = np.cov(a.T).astype(np.float32)
        u, sv = ol.jacobi(cov)
        w = np.linalg.eigvalsh(cov.astype(np.float64))[::-1]
        assert np.allclose(sv, w, rtol=2e-4, atol=1e-6)
        assert sv[0] >= sv[1] >= sv[2] >= 0
        assert np.allclose(u.T @ u, np.eye(3), atol=1e-5)
        assert np.allclose(cov @ u[:, 2], sv[2] * u[:, 2], atol=1e-4 * max(1.0, sv[0]))
    u, sv = ol.jacobi(np.full((3, 3), np.nan, np.float32))
    assert np.isnan(u).all() and np.isnan(sv).all()
    u, sv = ol.jacobi(np.zeros((3, 3), np.float32))
    assert np.array_equal(u, np.eye(3, dtype=np.float32)) and not sv.any()


def test_fxp_contract_primitives(oracle_built):
    """Shift, origins and quantisers of the fixed-point contract: v4 = a 2^-30 m grid, |Q| <= 2^35;
    the 2^-21 m grid of rounds 3-5 is still there as a witness (ARITH_FXP21)."""
    lib = oracle_built.restatement()
    L = lib.lib
    sh, zr, ox, oy = ol.Estimator(lib, arith=ol.ARITH_FXP).fxp_geometry()
    assert sh == 30 and zr == 32.0 and len(ox) == 504  # default CZM: every bin within 32 m of its origin
    sh3, zr3, ox3, oy3 = ol.Estimator(lib, arith=ol.ARITH_FXP21).fxp_geometry()
    assert sh3 == 21 and zr3 == 32.0 and np.array_equal(ox, ox3) and np.array_equal(oy, oy3)
    assert (np.abs(ox * 8 - np.rint(ox * 8)) == 0).all() and (np.abs(oy * 8 - np.rint(oy * 8)) == 0).all()
    p = lib.default_params()
    p.max_range = 500.0
    assert ol.Estimator(lib, p, arith=ol.ARITH_FXP).fxp_geometry()[0] == 29  # bigger bins, coarser grid
    for k in range(4):
        p.num_sectors_each_zone[k] = 1
    sh1, zr1, ox1, oy1 = ol.Estimator(lib, p, arith=ol.ARITH_FXP).fxp_geometry()
    assert sh1 == 26 and not ox1.any() and not oy1.any()  # one sector per ring: the sensor is the origin
    for s in (21, 30):
        q = lambda v, o: L.pwo_ext_quantise(ctypes.c_float(v), o, s)
        assert q(1.0, 0.0) == 1 << s and q(-1.0, 0.0) == -(1 << s) and q(10.125, 10.125) == 0
        assert q(1.5 / (1 << s), 0.0) == 2 and q(2.5 / (1 << s), 0.0) == 2 and q(0.5 / (1 << s), 0.0) == 0  # ties to even
        assert q(-1.5 / (1 << s), 0.0) == -2 and q(-2.5 / (1 << s), 0.0) == -2 and q(-0.5 / (1 << s), 0.0) == 0
        assert q(30.0, 12.5) == int(17.5 * (1 << s))
        # ONE rounding of the exact value: a hair above / below a tie next to a large origin (two roundings in double would tie)
        tie = np.float32(0.5 / (1 << s))
        up, dn = np.nextafter(tie, np.float32(1.0)), np.nextafter(tie, np.float32(0.0))
        assert q(float(tie), 12.5) == -int(12.5 * (1 << s)) and q(float(up), 12.5) == -int(12.5 * (1 << s)) + 1 and q(float(dn), 12.5) == -int(12.5 * (1 << s))
        assert q(float(np.float32(1e-30)), 3.0) == -(3 << s) and q(0.0, 3.0) == -(3 << s) and q(float(np.float32(-1e-42)), -3.0) == 3 << s
        qz = lambda v, z0: L.pwo_ext_quantise_z(ctypes.c_float(v), z0, s, 32.0)
        top = 32 << s
        assert qz(-1.75, -1.75) == 0 and qz(1e30, -1.75) == top and qz(-1e30, -1.75) == -top
        assert qz(float("inf"), 0.0) == top and qz(float("-inf"), 0.0) == -top and qz(float("nan"), 0.0) == -top
    # v4: every float of magnitude >= 2^-7 m is on the grid -- Q * 2^-30 + origin gives the float back exactly
    rng = np.random.default_rng(5)
    for v in np.concatenate([rng.uniform(-30, 30, 200), rng.uniform(-1, 1, 200) * 2.0 ** -6, [2.0 ** -7, -2.0 ** -7]]).astype(np.float32):
        if abs(v) >= 2.0 ** -7:
            assert L.pwo_ext_quantise(ctypes.c_float(float(v)), 1.625, 30) == int(round((float(v) - 1.625) * (1 << 30)))
            assert (L.pwo_ext_quantise(ctypes.c_float(float(v)), 1.625, 30) / (1 << 30)) + 1.625 == float(v)
    assert L.pwo_ext_z_origin(-1.73) == -1.75 and L.pwo_ext_z_origin(float("nan")) == 0.0
    assert L.pwo_ext_z_origin(float("inf")) == 0.0 and L.pwo_ext_z_origin(1e9) == 4096.0
