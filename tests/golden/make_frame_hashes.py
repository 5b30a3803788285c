"""Golden hashes for bench.py's spot checks: the ground MASK of a few frames of the two synthetic workloads,
as the CPU restatement of the contract gives it (which tests/test_oracle.py pins to the reference's own source compiled here, and
which equals all three builds of the reference on the varied frames -- tools/parity_statistics.py), reduced to a SHA-256 each.
bench.py compares the HIP path's masks of the same frames with these on the GPU box -- data only, nothing under oracle/ runs there.

    python tests/golden/make_frame_hashes.py        ->  tests/golden/frame_hashes.json"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "patchwork-plusplus_amd", "python"))
import numpy as np
import oracle_lib as ol
import pwpp_synth

VARIED = [0, 137, 301, 444, 512, 777, 900, 1023]   # of pwpp_synth.varied_frame(i): bench.py's `distinct` leg uses frames 0..1023
DENSE = [0, 7, 21, 40, 63]                          # of pwpp_synth.dense_frame(i): the `dense` leg uses 64 distinct clouds


def mask_hash(n, ground_idx):
    m = np.zeros(n, np.uint8)
    m[np.asarray(ground_idx)] = 1
    return hashlib.sha256(np.packbits(m).tobytes()).hexdigest()


def main():
    ol.build()
    lib = ol.restatement()
    out = {"what": "sha256 of np.packbits(ground mask) per frame, restatement of the contract (oracle/pwpp_oracle.cpp, ARITH_FXP), fresh state",
           "varied": {}, "dense": {}}
    for i in VARIED:
        pts = pwpp_synth.varied_frame(i)
        r = ol.Estimator(lib, arith=ol.ARITH_FXP).run(pts)
        out["varied"][str(i)] = {"points": int(pts.shape[0]), "ground": int(len(r.ground_idx)), "sha256": mask_hash(pts.shape[0], r.ground_idx)}
    p36 = lib.default_params()
    for k in range(4):
        p36.num_sectors_each_zone[k] = 36
    for i in DENSE:
        pts = pwpp_synth.dense_frame(i)
        r = ol.Estimator(lib, p36, arith=ol.ARITH_FXP).run(pts)
        out["dense"][str(i)] = {"points": int(pts.shape[0]), "ground": int(len(r.ground_idx)), "sha256": mask_hash(pts.shape[0], r.ground_idx)}
    with open(os.path.join(ROOT, "tests", "golden", "frame_hashes.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
