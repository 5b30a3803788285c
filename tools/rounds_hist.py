"""How many patches stop at which R-GPF round (early termination: a round whose integer totals repeat the round before's ends the
patch; pwpp_patch_record.rounds).  KITTI samples (default CZM) and dense synthetic 128-beam clouds (36-sector CZM), as batches.
Run on the GPU box: python tools/rounds_hist.py > gpurun_out/rounds_hist.txt"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, conftest
import pwpp_hip, pwpp_synth


def table(name, handle, frames):
    handle.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
    rows = {}
    for f in range(len(frames)):
        rec = handle.patch_records(f)
        for big in (False, True):
            sel = (rec["n_points"] >= 1024) == big
            for r in range(1, 9):
                m = sel & (rec["rounds"] == r)
                k = ("big (>= 1024 points)" if big else "small (< 1024 points)", r)
                c = rows.setdefault(k, [0, 0])
                c[0] += int(m.sum())
                c[1] += int(rec["n_points"][m].sum())
    print("== %s: %d frames" % (name, len(frames)))
    for cls in ("small (< 1024 points)", "big (>= 1024 points)"):
        tot = sum(v[0] for k, v in rows.items() if k[0] == cls)
        pts = sum(v[1] for k, v in rows.items() if k[0] == cls)
        print("  %-22s %6d patches, %9d points" % (cls, tot, pts))
        for r in range(1, 9):
            v = rows.get((cls, r), [0, 0])
            if v[0]:
                print("      stopped after round %d: %6d patches (%5.1f %%), %9d points (%5.1f %%)" % (r, v[0], 100.0 * v[0] / max(tot, 1), v[1], 100.0 * v[1] / max(pts, 1)))


kitti = [conftest.load_kitti(k) for k in range(6)]
table("KITTI samples 0-5, default CZM, num_iter = 3", pwpp_hip.Handle(), kitti)
p = pwpp_hip.default_params()
for k in range(4):
    p.num_sectors_each_zone[k] = 36
table("dense synthetic 128-beam clouds (seeds 1000-1003), 36-sector CZM, num_iter = 3", pwpp_hip.Handle(p), [pwpp_synth.make_dense_cloud(1000 + k) for k in range(4)])
p5 = pwpp_hip.default_params()
p5.num_iter = 5
table("KITTI samples 0-5, num_iter = 5", pwpp_hip.Handle(p5), kitti)
