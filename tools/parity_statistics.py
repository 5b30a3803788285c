"""How often does the product's arithmetic contract (exact integer moments on a 2^-30 m grid, contract v4; the HIP path equals its CPU
restatement bit for bit -- asserted by the GPU suite) give exactly the ground set of the reference's own patchworkpp.cpp?  CPU only.
(The 2^-21 m grid of rounds 3-5, contract v3, rides along as a witness: `contract_v3_*` in the report.)

Three populations, each frame through the restatement (ARITH_FXP) and through the THREE builds of the reference under oracle/_ref
(float sums in storage order, float sums in a 4-lane order, double sums rounded once):
  fresh     N varied 64-beam frames (pwpp_synth.varied_frame), a fresh object per frame
  stateful  S sequences of L frames each through ONE long-lived object per build (adaptive thresholds, histories, sensor height drift)
  dense     M dense 128-beam ~480 k-point frames, 36-sector CZM (BASELINE.json configs[4])
A frame on which the three reference builds agree has "a reference result"; the contract is held against it.  Where they differ among
themselves the frame is counted as split and the contract is compared with the exact build.  Rates come with 95 % Wilson intervals.

   python tools/parity_statistics.py [--fresh 2000] [--seqs 10] [--seq-len 200] [--dense 200] [--workers 8] [--out profiles/r05_parity_statistics.json]"""
import argparse, json, math, os, sys, time
import multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "patchwork-plusplus_amd", "python"))
import numpy as np
import oracle_lib as ol
import pwpp_synth

FLAV = (("eigen_f32", ol.ARITH_EIGEN_F32), ("f32_packet4", ol.ARITH_F32_PACKET4), ("exact_f64", ol.ARITH_EXACT_F64))


def wilson(k, n, z=1.96):
    if n == 0:
        return [0.0, 1.0]
    p = k / n
    d = 1 + z * z / n
    c = p + z * z / (2 * n)
    h = z * math.sqrt(p * (1 - p) / n + z * z / (4 * n * n))
    return [max(0.0, (c - h) / d), min(1.0, (c + h) / d)]


def dense_params(lib):
    p = lib.default_params()
    for k in range(4):
        p.num_sectors_each_zone[k] = 36
    return p


def judge_frame(sets, mine):
    ref = sets
    agree = np.array_equal(ref["eigen_f32"], ref["f32_packet4"]) and np.array_equal(ref["eigen_f32"], ref["exact_f64"])
    return {"agree": bool(agree), "vs_exact": int(np.setxor1d(mine, ref["exact_f64"]).size), "vs_f32": int(np.setxor1d(mine, ref["eigen_f32"]).size),
            "v3_vs_exact": int(np.setxor1d(ref["fxp21"], ref["exact_f64"]).size), "vs_pk4": int(np.setxor1d(mine, ref["f32_packet4"]).size),
            "f32_vs_exact": int(np.setxor1d(ref["eigen_f32"], ref["exact_f64"]).size), "pk4_vs_exact": int(np.setxor1d(ref["f32_packet4"], ref["exact_f64"]).size)}


def fresh_job(args):
    kind, i = args
    pts = pwpp_synth.varied_frame(i) if kind == "fresh" else pwpp_synth.make_dense_cloud(5000 + i)
    rs = ol.restatement()
    prm = (lambda lib: dense_params(lib)) if kind == "dense" else (lambda lib: None)
    mine = np.sort(ol.Estimator(rs, prm(rs), arith=ol.ARITH_FXP).run(pts).ground_idx)
    sets = {"fxp21": np.sort(ol.Estimator(rs, prm(rs), arith=ol.ARITH_FXP21).run(pts).ground_idx)}  # (rounds 3-5's grid: a witness)
    for name, a in FLAV:
        lib = ol.reference(a)
        sets[name] = np.sort(ol.Estimator(lib, prm(lib), arith=a).run(pts).ground_idx)
    r = judge_frame(sets, mine)
    r.update(frame=i, points=int(pts.shape[0]))
    return r


def seq_job(args):
    s, length = args
    rs = ol.restatement()
    est = {"mine": ol.Estimator(rs, arith=ol.ARITH_FXP), "fxp21": ol.Estimator(rs, arith=ol.ARITH_FXP21)}
    for name, a in FLAV:
        est[name] = ol.Estimator(ol.reference(a), arith=a)
    rows, hmax = [], 0.0
    for t in range(length):
        pts = pwpp_synth.varied_frame(100_000 + 1000 * s + t)
        out = {k: e.run(pts) for k, e in est.items()}
        sets = {k: np.sort(out[k].ground_idx) for k in [n for n, _ in FLAV] + ["fxp21"]}
        r = judge_frame(sets, np.sort(out["mine"].ground_idx))
        r.update(seq=s, t=t, points=int(pts.shape[0]))
        hmax = max(hmax, abs(out["mine"].sensor_height - out["exact_f64"].sensor_height))
        r["height_vs_exact"] = abs(out["mine"].sensor_height - out["exact_f64"].sensor_height)
        r["height_f32_vs_exact"] = abs(out["eigen_f32"].sensor_height - out["exact_f64"].sensor_height)
        rows.append(r)
    return rows


def summarise(rows, what):
    cons = [r for r in rows if r["agree"]]
    split = [r for r in rows if not r["agree"]]
    eq = sum(1 for r in cons if r["vs_exact"] == 0)
    miss = [{k: r[k] for k in r if k in ("frame", "seq", "t", "points", "vs_exact")} for r in cons if r["vs_exact"] != 0]
    out = {"what": what, "frames": len(rows), "points": int(sum(r["points"] for r in rows)),
           "reference_builds_unanimous": len(cons), "reference_builds_split": len(split),
           "split_rate": len(split) / max(len(rows), 1), "split_rate_ci95": wilson(len(split), len(rows)),
           "contract_equals_unanimous_reference": eq, "rate": eq / max(len(cons), 1), "rate_ci95": wilson(eq, len(cons)),
           "misses": miss, "largest_miss_indices": max([m["vs_exact"] for m in miss], default=0),
           "contract_v3_2e-21_grid_equals_unanimous_reference": sum(1 for r in cons if r["v3_vs_exact"] == 0),
           "contract_v3_equals_exact_on_split": sum(1 for r in split if r["v3_vs_exact"] == 0),
           "contract_equals_exact_on_split": sum(1 for r in split if r["vs_exact"] == 0),
           # (fit sets of 1-3 points follow the reference's own float arithmetic, determinate there: on such frames BOTH float builds agree with
           # each other and with the contract, and the exact-f64 build -- double sums for those sets too -- is the odd one out)
           "contract_equals_both_float_builds_on_split": sum(1 for r in split if r["vs_f32"] == 0 and r["vs_pk4"] == 0),
           "contract_equals_some_reference_build_on_split": sum(1 for r in split if r["vs_exact"] == 0 or r["vs_f32"] == 0 or r["vs_pk4"] == 0),
           "largest_distance_to_the_nearest_build_on_split": max([min(r["vs_exact"], r["vs_f32"], r["vs_pk4"]) for r in split], default=0),
           "contract_no_further_from_exact_than_the_float_builds_on_split": sum(1 for r in split if r["vs_exact"] <= max(r["f32_vs_exact"], r["pk4_vs_exact"])),
           "split_frames": [{k: r[k] for k in r if k != "agree"} for r in split][:60]}
    if rows and "height_vs_exact" in rows[0]:
        out["max_sensor_height_difference_vs_exact_m"] = max(r["height_vs_exact"] for r in rows)
        out["max_sensor_height_difference_f32_build_vs_exact_m"] = max(r["height_f32_vs_exact"] for r in rows)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fresh", type=int, default=2000)
    ap.add_argument("--seqs", type=int, default=10)
    ap.add_argument("--seq-len", type=int, default=200)
    ap.add_argument("--dense", type=int, default=200)
    ap.add_argument("--workers", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_parity_statistics.json"))
    a = ap.parse_args()
    ol.build()
    assert all(ol.reference(x) is not None for _, x in FLAV), "oracle/_ref is not built (make -C oracle, needs /root/reference)"
    t0 = time.time()
    with mp.get_context("fork").Pool(a.workers) as pool:
        fresh = pool.map(fresh_job, [("fresh", i) for i in range(a.fresh)], chunksize=8)
        seqs = pool.map(seq_job, [(s, a.seq_len) for s in range(a.seqs)], chunksize=1)
        dense = pool.map(fresh_job, [("dense", i) for i in range(a.dense)], chunksize=2)
    rep = {"script": "tools/parity_statistics.py", "seconds": time.time() - t0,
           "fresh": summarise(fresh, "%d varied 64-beam frames (pwpp_synth.varied_frame(0..)), fresh state, default parameters" % a.fresh),
           "stateful": summarise([r for rows in seqs for r in rows], "%d sequences of %d varied frames, one long-lived object per build" % (a.seqs, a.seq_len)),
           "dense": summarise(dense, "%d dense 128-beam frames (pwpp_synth.make_dense_cloud(5000..)), 36-sector CZM, fresh state" % a.dense)}
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    for k in ("fresh", "stateful", "dense"):
        r = rep[k]
        print("%-8s %5d frames: reference unanimous on %5d, contract equal on %5d (rate %.4f, 95%% CI %.4f-%.4f), largest miss %d indices; split %d (contract = exact on %d)"
              % (k, r["frames"], r["reference_builds_unanimous"], r["contract_equals_unanimous_reference"], r["rate"], r["rate_ci95"][0], r["rate_ci95"][1],
                 r["largest_miss_indices"], r["reference_builds_split"], r["contract_equals_exact_on_split"]))


if __name__ == "__main__":
    main()
