"""CPU probe (VERDICT r02 'What's weak' 1): the restatement's ARITH_FXP flavour (= the HIP path, bit for bit) against the three
builds of the reference (oracle/_ref) under parameter sets that produce fit sets of <= 3 points.
Prints the ground-index symmetric difference against the float reference build per (parameter set, frame)."""
import lzma
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

ROS = dict(sensor_height=1.88, num_iter=3, num_lpr=20, num_min_pts=0, th_seeds=0.3, th_dist=0.125, th_seeds_v=0.25,
           th_dist_v=0.9, max_range=80.0, min_range=1.0, uprightness_thr=0.101, enable_RNR=0)


def kitti(i):
    with lzma.open(os.path.join(ROOT, "tests", "golden", "kitti_%06d.bin.xz" % i)) as f:
        return np.frombuffer(f.read(), np.float32).reshape(-1, 4).copy()


def params(lib, variant):
    p = lib.default_params()
    for k, v in variant.items():
        setattr(p, k, v)
    return p


def main():
    lib = ol.restatement()
    refs = {a: ol.reference(a) for a in (ol.ARITH_EIGEN_F32, ol.ARITH_EXACT_F64, ol.ARITH_F32_PACKET4)}
    rows = [("defaults", {}, (0, 3), False), ("num_min_pts=0", dict(num_min_pts=0), (0, 3), False),
            ("num_min_pts=1", dict(num_min_pts=1), (0, 3), False), ("num_min_pts=3", dict(num_min_pts=3), (0,), False),
            ("ROS launch", ROS, (0, 3), True), ("num_lpr=1", dict(num_lpr=1), (0,), False)]
    total = 0
    for name, v, frames, n3 in rows:
        for fr in frames:
            pts = kitti(fr)
            if n3:
                pts = np.ascontiguousarray(pts[:, :3])
            base = ol.Estimator(refs[0], params(refs[0], v), arith=0).run(pts).ground_idx
            out = []
            for a in (ol.ARITH_EXACT_F64, ol.ARITH_F32_PACKET4):
                g = ol.Estimator(refs[a], params(refs[a], v), arith=a).run(pts).ground_idx
                out.append(len(np.setxor1d(g, base)))
            g = ol.Estimator(lib, params(lib, v), arith=ol.ARITH_FXP).run(pts).ground_idx
            d = len(np.setxor1d(g, base))
            if name != "num_lpr=1":
                total += d
            print("%-16s kitti %d  ref-exact %3d  ref-pk4 %3d  contract(=HIP) %3d" % (name, fr, out[0], out[1], d))
    # the ROS set as ONE object over 12 frames (N x 3)
    ests = {a: ol.Estimator(refs[a], params(refs[a], ROS), arith=a) for a in (0, 2)}
    fx = ol.Estimator(lib, params(lib, ROS), arith=ol.ARITH_FXP)
    for k in range(12):
        pts = np.ascontiguousarray(kitti(k % 6)[:, :3])
        base = ests[0].run(pts).ground_idx
        e = len(np.setxor1d(ests[2].run(pts).ground_idx, base))
        d = len(np.setxor1d(fx.run(pts).ground_idx, base))
        total += d if e == 0 else 0
        print("ROS sequence frame %2d  ref-exact %3d  contract %3d" % (k, e, d))
    print("TOTAL (rows where the reference's flavours agree):", total)
    return total


if __name__ == "__main__":
    sys.exit(0 if main() == 0 else 1)
