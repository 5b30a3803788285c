#!/usr/bin/env python3
"""Details of a failing case of tools/fuzz_parity.py: python tools/fuzz_triage.py <seed> -- which patches differ, and what their points look like."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fuzz_parity as fz
from fuzz_parity import ol, pwpp_hip, to_oracle_params

orig = fz.assert_frame_equal


def verbose_equal(h, frame, ref, n_points, state_index=None, check_state=True):
    try:
        orig(h, frame, ref, n_points, state_index, check_state)
    except AssertionError as e:
        print("frame", frame, "differs:", str(e)[:100], "|", fz.LAST, "| one-pass stats", h.one_pass_stats())
        print("counts hip", h.counts(frame), "ref", (len(ref.ground_idx), len(ref.nonground_idx), len(ref.centers)))
        rec = h.patch_records(frame)
        rr = ref.records
        if len(rec) != len(rr):
            print("patch count", len(rec), len(rr))
            hb, rb = set(rec["bin"].tolist()), set(rr["bin"].tolist())
            print("only hip", sorted(hb - rb)[:20], "only ref", sorted(rb - hb)[:20])
            hn = dict(zip(rec["bin"].tolist(), rec["n_points"].tolist())); rn = dict(zip(rr["bin"].tolist(), rr["n_points"].tolist()))
            common = sorted(hb & rb)
            print("common bins", len(common), "with equal n_points", sum(hn[b] == rn[b] for b in common), [(b, hn[b], rn[b]) for b in common[:12]])
            print("points in patches hip", sum(hn.values()), "ref", sum(rn.values()), "frame points", n_points)
        else:
            bad = [i for i in range(len(rec)) if any(not np.array_equal(rec[f][i], rr[f][i], equal_nan=True) for f in ("n_points", "n_ground", "mean", "normal", "d", "decision"))]
            print(len(bad), "patches differ of", len(rec))
            for i in bad[:6]:
                print(" bin", rec["bin"][i], "ring", rec["concentric_idx"][i], "n", rec["n_points"][i], rr["n_points"][i], "ng", rec["n_ground"][i], rr["n_ground"][i],
                      "decision", rec["decision"][i], rr["decision"][i])
                print("   hip mean", rec["mean"][i], "normal", rec["normal"][i], "d", rec["d"][i])
                print("   ref mean", rr["mean"][i], "normal", rr["normal"][i], "d", rr["d"][i])
                cl = [c for c in CLOUDS if c.shape[0] == n_points][-1]
                zb = np.sort(cl[bins_of(cl, PARAMS) == rec["bin"][i], 2])
                print("   z of the bin's points (numpy binning): n", len(zb), "lowest", zb[:24], "highest", zb[-4:])
        st = h.state(frame if state_index is None else state_index)
        print("sensor_height hip", st.sensor_height, "ref", ref.sensor_height)
        raise


def bins_of(pts, p):
    x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    r = np.sqrt(x * x + y * y)
    th = np.arctan2(y, x)
    th = np.where(th > 0, th, th + 2 * np.pi)
    mn, mx = p.min_range, p.max_range
    mr = [mn, (7 * mn + mx) / 8, (3 * mn + mx) / 4, (mn + mx) / 2, mx]
    rings, sect = list(p.num_rings_each_zone), list(p.num_sectors_each_zone)
    base = np.cumsum([0] + [a * b for a, b in zip(rings, sect)])
    ok = (r > mn) & (r <= mx)
    k = np.digitize(r, mr[1:4])
    code = np.full(len(r), -1)
    for z in range(4):
        m = ok & (k == z)
        ring = np.minimum(((r[m] - mr[z]) / ((mr[z + 1] - mr[z]) / rings[z])).astype(int), rings[z] - 1)
        sec = np.minimum((th[m] / (2 * np.pi / sect[z])).astype(int), sect[z] - 1)
        code[m] = base[z] + ring * sect[z] + sec
    return code


CLOUDS = []
orig_cloud = fz.random_cloud


def keep_cloud(rng, sh):
    c = orig_cloud(rng, sh)
    CLOUDS.append(c)
    return c


fz.random_cloud = keep_cloud
fz.assert_frame_equal = verbose_equal
seed = int(sys.argv[1])
ol.build()
rng = np.random.default_rng(seed)
p = fz.random_params(rng)
PARAMS = p
print("params:", {n: (list(getattr(p, n)) if hasattr(getattr(p, n), "__len__") else getattr(p, n)) for n, _ in ol.Params._fields_})
try:
    print(fz.one_case(seed, ol.restatement()))
except AssertionError:
    pass
