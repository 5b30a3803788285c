#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the build container only).

Inputs : /root/reference/data/00000{0..5}.bin   -- the only inputs the reference ships
         (KITTI Velodyne layout: float32 x,y,z,intensity).
Outputs: tests/golden/kitti_00000k.bin.xz        -- the frames, byte-identical after unxz
         tests/golden/kitti_golden.npz           -- what the REFERENCE's own patchworkpp.cpp
             (compiled unmodified against oracle/eigen_shim -> oracle/_ref/libpwpp_ref*.so)
             produces for them: ground masks, counts, centres, normals, adaptive state,
             sha256 of the index lists in the reference's own output order; for the three shim
             flavours (eigen-f32 / exact-f64 arbiter / f32 4-lane order), fresh-state per frame
             and as one 6-frame sequence.

The reference has no golden vectors of its own; these are outputs of the
reference itself run here, which is what pins oracle/pwpp_oracle.cpp.
"""
import hashlib
import lzma
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

REF_DATA = "/root/reference/data"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pack(res, n, out, key):
    mask = np.zeros(n, np.uint8)
    mask[res.ground_idx] = 1
    out[key + "ground_mask"] = np.packbits(mask)
    out[key + "counts"] = np.array([len(res.ground_idx), len(res.nonground_idx), len(res.centers)], np.int64)
    out[key + "centers"] = res.centers
    out[key + "normals"] = res.normals
    out[key + "state"] = np.concatenate([[res.sensor_height], res.elevation_thr, res.flatness_thr])
    out[key + "sha_ground_order"] = np.array(sha(res.ground_idx))
    out[key + "sha_nonground_order"] = np.array(sha(res.nonground_idx))
    out[key + "hist_len"] = np.array([[len(h) for h in res.hist_elev], [len(h) for h in res.hist_flat]], np.int64)


def main():
    ol.build()
    frames = []
    for k in range(6):
        raw = open(os.path.join(REF_DATA, "%06d.bin" % k), "rb").read()
        with open(os.path.join(HERE, "kitti_%06d.bin.xz" % k), "wb") as f:
            f.write(lzma.compress(raw, preset=9 | lzma.PRESET_EXTREME))
        frames.append(np.frombuffer(raw, np.float32).reshape(-1, 4))
    out = {"md5": np.array([hashlib.md5(f.tobytes()).hexdigest() for f in frames]),
           "n_points": np.array([f.shape[0] for f in frames], np.int64)}
    for arith, name in ((ol.ARITH_EIGEN_F32, "f32"), (ol.ARITH_EXACT_F64, "exact"), (ol.ARITH_F32_PACKET4, "pk4")):
        lib = ol.reference(arith)
        assert lib is not None, "build oracle/_ref first (make -C oracle ref)"
        for k, f in enumerate(frames):
            pack(ol.Estimator(lib, arith=arith).run(f), f.shape[0], out, "%s/fresh/%d/" % (name, k))
        est = ol.Estimator(lib, arith=arith)
        for k, f in enumerate(frames):
            pack(est.run(f), f.shape[0], out, "%s/seq/%d/" % (name, k))
    np.savez_compressed(os.path.join(HERE, "kitti_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
