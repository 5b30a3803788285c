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

FLAVOURS = [(ol.ARITH_EIGEN_F32, "f32"), (ol.ARITH_FXP, "fxp")]


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
        assert tuple(golden["fxp/fresh/%d/counts" % k]) == a
    seq = [72665, 71848, 71263, 70535, 69095, 67614]
    assert [int(golden["f32/seq/%d/counts" % k][0]) for k in range(6)] == seq


def test_fxp_flavour_same_index_sets_as_eigen_f32_flavour(golden):
    """The fixed-point plane-fit arithmetic changes no decision on the reference's own data,
    and moves plane normals by far less than the 1e-4 tolerance of BASELINE.json."""
    for mode in ("fresh", "seq"):
        for k in range(6):
            a, b = "f32/%s/%d/" % (mode, k), "fxp/%s/%d/" % (mode, k)
            assert np.array_equal(golden[a + "ground_mask"], golden[b + "ground_mask"])
            assert np.abs(golden[a + "normals"] - golden[b + "normals"]).max() < 1e-4
            assert np.abs(golden[a + "centers"] - golden[b + "centers"]).max() < 1e-4


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


def test_fxp_quantiser(oracle_built):
    L = oracle_built.restatement().lib
    assert L.pwo_ext_fxp_shift(80.0) == 16 and L.pwo_ext_fxp_shift(120.0) == 16
    assert L.pwo_ext_fxp_shift(200.0) == 15 and L.pwo_ext_fxp_shift(5.0) == 20
    q = lambda v: L.pwo_ext_quantise(ctypes.c_float(v), 16)
    assert q(1.0) == 65536 and q(-1.0) == -65536 and q(0.0) == 0
    assert q(1.5 / 65536) == 2 and q(2.5 / 65536) == 2 and q(0.5 / 65536) == 0  # ties to even
    assert q(1e30) == 8388607 and q(-1e30) == -8388607 and q(float("nan")) == 0
