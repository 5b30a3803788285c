"""The arithmetic contract needs IEEE-exact f32 / f64 division and sqrt on gfx950 and an
exactly matching Jacobi sequence.  Exercised through the real pipeline: tiny frames whose
single patch is built from adversarial covariances, compared bitwise with the oracle."""
import numpy as np
import pytest

import oracle_lib as ol
import pwpp_hip

pytestmark = pytest.mark.gpu


def test_many_small_patches_bitwise(oracle_built):
    """4000 random single-patch frames (10..300 points, random plane tilt / thickness /
    scale, some degenerate): every fit result must equal the oracle bit for bit."""
    lib = oracle_built.restatement()
    rng = np.random.default_rng(123)
    frames = []
    for i in range(400):
        n = int(rng.integers(10, 300))
        r = rng.uniform(3.0, 75.0)
        a = rng.uniform(0, 2 * np.pi)
        c = np.array([r * np.cos(a), r * np.sin(a), -1.7])
        tilt = rng.normal(0, [0.05, 0.3, 1.0][i % 3], 2)
        spread = rng.uniform(0.05, 1.5)
        xy = rng.normal(0, spread, (n, 2))
        z = xy @ tilt + rng.normal(0, [1e-4, 0.02, 0.3][(i // 3) % 3], n)
        if i % 17 == 0:
            z[:] = 0.0  # perfectly flat
        if i % 19 == 0:
            xy[:, 1] = xy[:, 0]  # a line
        pts = np.zeros((n, 4), np.float32)
        pts[:, :2] = xy + c[:2]
        pts[:, 2] = z + c[2]
        pts[:, 3] = 0.5
        frames.append(pts)
    h = pwpp_hip.Handle()
    h.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
    bad = 0
    for k, pts in enumerate(frames):
        ref = ol.Estimator(lib, arith=ol.ARITH_FXP).run(pts)
        rec = h.patch_records(k)
        assert len(rec) == len(ref.records)
        for fld in ("n_ground", "decision", "mean", "normal", "sv", "d"):
            if not np.array_equal(rec[fld], ref.records[fld], equal_nan=True):
                bad += 1
        assert np.array_equal(np.sort(h.ground_indices(k)), np.sort(ref.ground_idx))
    assert bad == 0
