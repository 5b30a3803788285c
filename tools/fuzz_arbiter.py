#!/usr/bin/env python3
"""CPU only: the arithmetic contract of the product (restatement, fixed-point flavour -- what the HIP path reproduces bit
for bit) against the exact-f64 arbiter (the same control flow with the sums of patchworkpp.cpp:56-60 in exact arithmetic)
on the randomised inputs of tools/fuzz_parity.py, and the float flavour of the reference measured the same way.
Heights further than 32 m from a patch's lowest points are left out (the contract clamps them), and so
are parameter sets whose fits have one or two points (no arithmetic defines those planes).

usage: python tools/fuzz_arbiter.py [cases] [first_seed]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
os.environ["FUZZ_NO_ODD"] = "1"
import fuzz_parity as fz  # noqa: E402
from flavour_metrics import compare  # noqa: E402
from fuzz_parity import ol, to_oracle_params  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ol.build()
    lib = ol.restatement()
    tot = dict(frames=0, fxp_sym=0, fxp_frames=0, f32_sym=0, f32_frames=0, fxp_dc=0.0, f32_dc=0.0, fxp_dn_well=0.0, f32_dn_well=0.0,
               fxp_excess=0.0, only_fxp=0, only_f32=0, both=0)
    for seed in range(first, first + cases):
        rng = np.random.default_rng(seed)
        p = fz.random_params(rng)
        p.num_min_pts = max(p.num_min_pts, 5)
        p.num_lpr = max(p.num_lpr, 5)
        op = to_oracle_params(p)
        for _ in range(2):
            pts = fz.random_cloud(rng, p.sensor_height)
            ex = ol.Estimator(lib, op, arith=ol.ARITH_EXACT_F64).run(pts)
            fx = ol.Estimator(lib, op, arith=ol.ARITH_FXP).run(pts)
            f3 = ol.Estimator(lib, op, arith=ol.ARITH_EIGEN_F32).run(pts)
            a = compare(fx.ground_idx, fx.records, ex.ground_idx, ex.records, min_points=4)
            b = compare(f3.ground_idx, f3.records, ex.ground_idx, ex.records, min_points=4)
            tot["frames"] += 1
            tot["fxp_sym"] += a["symdiff"]
            tot["fxp_frames"] += a["symdiff"] > 0
            tot["f32_sym"] += b["symdiff"]
            tot["f32_frames"] += b["symdiff"] > 0
            tot["only_fxp"] += a["symdiff"] > 0 and b["symdiff"] == 0
            tot["only_f32"] += b["symdiff"] > 0 and a["symdiff"] == 0
            tot["both"] += a["symdiff"] > 0 and b["symdiff"] > 0
            for k, m in (("fxp", a), ("f32", b)):
                if not m["patches_differ"]:
                    tot[k + "_dc"] = max(tot[k + "_dc"], np.nan_to_num(m["dc"]))
                    tot[k + "_dn_well"] = max(tot[k + "_dn_well"], np.nan_to_num(m["dn_well"]))
            if not a["patches_differ"]:
                tot["fxp_excess"] = max(tot["fxp_excess"], np.nan_to_num(a["excess"]))
            if a["symdiff"]:
                print("seed %d: contract vs arbiter: %d indices differ of %d points (float flavour: %d)" % (seed, a["symdiff"], pts.shape[0], b["symdiff"]))
    print("%d frames | contract vs exact arbiter: %d indices in %d frames | float flavour vs arbiter: %d indices in %d frames | "
          "frames where only the contract differs %d, only the float flavour %d, both %d"
          % (tot["frames"], tot["fxp_sym"], tot["fxp_frames"], tot["f32_sym"], tot["f32_frames"], tot["only_fxp"], tot["only_f32"], tot["both"]))


if __name__ == "__main__":
    main()
