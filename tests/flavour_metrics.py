"""How far two evaluations of the same frame are apart (TEST INFRASTRUCTURE): index sets, plane centres,
plane normals against the conditioning of the patch.  Used by tests/test_arith_flavours.py (CPU: the oracle's
flavours among themselves; GPU: the HIP path against the exact-f64 arbiter and the float reference)."""
import numpy as np

EPS32 = 2.0 ** -24


def patch_condition(sv):
    """sigma_0 / (sigma_1 - sigma_2): how strongly a relative perturbation of the covariance turns the normal
    (the normal is the singular vector of the SMALLEST singular value, reference patchworkpp.cpp:66)."""
    sv = np.asarray(sv, np.float64)
    return sv[:, 0] / np.maximum(sv[:, 1] - sv[:, 2], 1e-300)


def normal_distance(a, b):
    """max-abs difference of unit normals; a normal with |n_z| < 1e-3 may come out with either sign (the
    reference flips on n_z < 0, :68), which is not a deviation."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b).max(axis=1)
    flip = np.abs(a + b).max(axis=1)
    amb = np.minimum(np.abs(a[:, 2]), np.abs(b[:, 2])) < 1e-3
    return np.where(amb, np.minimum(d, flip), d)


def compare(ground_a, recs_a, ground_b, recs_b, min_points=0, min_ground=0):
    """a against b (b = the arbiter: its singular values define the conditioning).  recs_*: structured arrays
    with mean / normal / sv per processed patch, same patches in the same order.
    Returns dict(symdiff, iou, dc, dn, dn_well, cond_of_worst, excess) where excess = max over the patches of
    dn / (3e-5 + 4e-10 * cond): <= 1 means every patch is within the bound of DESIGN.md section 3.4.
    min_points: patches with fewer points are left out of the plane statistics (two or three points do not
    define a plane: their normal is whatever the last bit of the covariance says, in every arithmetic).
    min_ground: patches whose FINAL fit set (a's ground points) is smaller are left out of the plane statistics -- the
    product evaluates sets of 1-3 points in the reference's float arithmetic (contract v3), which is not the arbiter's."""
    ga, gb = np.asarray(ground_a), np.asarray(ground_b)
    sym = len(np.setxor1d(ga, gb))
    union = len(np.union1d(ga, gb))
    out = {"symdiff": sym, "iou": 1.0 - sym / union if union else 1.0, "patches_differ": len(recs_a) != len(recs_b)}
    if out["patches_differ"] or len(recs_a) == 0:
        v = np.nan if out["patches_differ"] else 0.0  # (no patch at all: nothing to compare)
        out.update(dc=v, dn=v, dn_well=v, cond_of_worst=v, excess=v)
        return out
    ok = np.isfinite(recs_b["normal"]).all(axis=1) & np.isfinite(recs_a["normal"]).all(axis=1) & (recs_b["n_points"] >= min_points) & (recs_a["n_ground"] >= min_ground)
    cond = np.nan_to_num(patch_condition(recs_b["sv"]), nan=np.inf, posinf=1e300)
    dn = np.where(ok, normal_distance(recs_a["normal"], recs_b["normal"]), 0.0)
    dc = np.where(ok, np.abs(recs_a["mean"].astype(np.float64) - recs_b["mean"].astype(np.float64)).max(axis=1), 0.0)
    i = int(np.argmax(dn))
    out.update(dc=float(dc.max()), dn=float(dn[i]), dn_well=float(np.where(cond < 100.0, dn, 0.0).max()),
               cond_of_worst=float(cond[i]), excess=float((dn / (3e-5 + 4e-10 * cond)).max()))
    return out
