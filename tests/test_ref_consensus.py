"""The KITTI-only pin, widened: 208 realistic 64-beam frames -- undulating terrain, slopes, boxes, reflected
noise, the sensor mounted between 1.55 and 1.90 m while the parameters keep the default 1.723 m, default CZM -- through the HIP
path and through ALL THREE builds of the reference's own patchworkpp.cpp under oracle/_ref (float sums in storage order, float
sums in a 4-lane order, double sums rounded once).

Where the three builds of the reference agree on a frame's ground set -- i.e. where the reference's result does not hang on the
summation order Eigen happens to be compiled with -- the product is held against that set; where they disagree among themselves
there is no single "reference result", those frames are counted and the product is held against the exact-arithmetic build (the
arbiter of DESIGN.md section 3.4).  MEASURED (round 4, all 208 frames on the CPU with the restatement of the contract, which the HIP
path equals bit for bit -- asserted below): the three builds are unanimous on 204 frames; on 203 of them the contract gives exactly
their set, on ONE (frame 6) it differs by one index of 128 075 -- the 2^-21 m grid of the z sums moves cov_xz by a few float ulps,
the normal by as many, and a point 1e-7 m from th_dist changes sides; on the 4 frames where the float builds part from exact
arithmetic (by 1, 1, 8 and 129 indices) the contract sides with exact arithmetic.  A grid four or eight times finer for z
(tried in the restatement) trades frame 6 for another knife edge (frame 137): no arithmetic that is not Eigen's own order can be
unanimous-exact on every frame, and Eigen's order is not knowable here.  So the assertions are: the HIP path
equals the contract on every frame; it equals a unanimous reference on at least 99 % of the unanimous frames and never differs
from it by more than 64 indices (round 5, 10 400 frames on the CPU -- tools/parity_statistics.py, profiles/r05_parity_statistics_10k.json:
99.83 % of 5 829 fresh and 99.79 % of 3 869 stateful unanimous frames exact, the eighteen misses 1-31 indices; dense 36-sector frames
334 of 334); on the other frames it equals exact arithmetic or is no further from it than the float builds are (true of these 208
frames; over the 10 400 it is further than both float builds on 4 of 368 split frames).
The report goes to gpurun_out/ref_consensus.json (tracked copy: profiles/r04_ref_consensus.json).

CPU part (-m "not gpu"): the same consensus logic on 12 frames with the CPU restatement of the contract standing in for the HIP
path, so that the harness itself is covered where there is no GPU."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import pwpp_synth

FLAVOURS = (("eigen_f32", ol.ARITH_EIGEN_F32), ("f32_packet4", ol.ARITH_F32_PACKET4), ("exact_f64", ol.ARITH_EXACT_F64))


def consensus_frame(i):
    """Frame i of the set: deterministic, realistic, varied."""
    return pwpp_synth.varied_frame(i)


def reference_sets(pts):
    """Ground index sets of the three reference builds (fresh object each, as the HIP batch's fresh state)."""
    out = {}
    for name, arith in FLAVOURS:
        lib = ol.reference(arith)
        assert lib is not None, "oracle/_ref is not built (make -C oracle)"
        out[name] = np.sort(ol.Estimator(lib, arith=arith).run(pts).ground_idx)
    return out


def judge(frames, product_sets):
    """The consensus bookkeeping: returns the report; raises on a frame where the product leaves the reference's consensus."""
    rep = {"frames": len(frames), "points": int(sum(f.shape[0] for f in frames)), "consensus_frames": 0, "split_frames": [],
           "product_equals_consensus": 0, "product_equals_exact_on_split_frames": 0, "consensus_misses": []}
    for i, (pts, mine) in enumerate(zip(frames, product_sets)):
        ref = reference_sets(pts)
        agree = np.array_equal(ref["eigen_f32"], ref["f32_packet4"]) and np.array_equal(ref["eigen_f32"], ref["exact_f64"])
        if agree:
            rep["consensus_frames"] += 1
            d = int(np.setxor1d(mine, ref["exact_f64"]).size)
            # (largest miss over 10 400 frames: 31 indices -- one small patch at the edge of a GLE decision; profiles/r05_parity_statistics_10k.json)
            assert d <= 64, "frame %d: the three reference builds agree, the product differs by %d indices" % (i, d)
            rep["product_equals_consensus"] += d == 0
            if d:
                rep["consensus_misses"].append({"frame": i, "indices": d, "points": int(pts.shape[0])})
        else:
            d = {"frame": i,
                 "f32_vs_exact": int(np.setxor1d(ref["eigen_f32"], ref["exact_f64"]).size),
                 "pk4_vs_exact": int(np.setxor1d(ref["f32_packet4"], ref["exact_f64"]).size),
                 "f32_vs_pk4": int(np.setxor1d(ref["eigen_f32"], ref["f32_packet4"]).size),
                 "product_vs_exact": int(np.setxor1d(mine, ref["exact_f64"]).size),
                 "product_vs_f32": int(np.setxor1d(mine, ref["eigen_f32"]).size)}
            rep["split_frames"].append(d)
            rep["product_equals_exact_on_split_frames"] += d["product_vs_exact"] == 0
    rep["split_rate"] = len(rep["split_frames"]) / max(len(frames), 1)
    return rep


def test_consensus_harness_with_the_restatement(oracle_built):
    """No GPU: the restatement of the product's contract (fixed-point sums) against the consensus of the three reference builds."""
    frames = [consensus_frame(i) for i in range(12)]
    lib = ol.restatement()
    mine = [np.sort(ol.Estimator(lib, arith=ol.ARITH_FXP).run(f).ground_idx) for f in frames]
    rep = judge(frames, mine)
    assert rep["consensus_frames"] + len(rep["split_frames"]) == 12
    assert all(d["product_vs_exact"] <= max(d["f32_vs_exact"], d["pk4_vs_exact"]) for d in rep["split_frames"]), rep
    # A RATE, not a list of frames: over 10 400 frames the contract misses a unanimous reference on 0.17-0.21 % of the
    # frames, by 1-31 indices (tools/parity_statistics.py, profiles/r05_parity_statistics_10k.json; frame 6 of this set is one of them).
    # Twelve frames may hold one such frame, not two, and a miss stays a handful of indices.
    assert len(rep["consensus_misses"]) <= 1 and all(m["indices"] <= 64 for m in rep["consensus_misses"]), rep["consensus_misses"]


@pytest.mark.gpu
def test_hip_path_equals_the_consensus_of_the_three_reference_builds():
    import pwpp_hip
    n = 208
    frames = [consensus_frame(i) for i in range(n)]
    h = pwpp_hip.Handle()
    mine, mine_ng = [], []
    for b0 in range(0, n, 104):  # two batches of 104 frames: the throughput plan and one-pass binning
        chunk = frames[b0:b0 + 104]
        h.estimate_ground_batch(chunk, mode=pwpp_hip.MODE_FRESH)
        mine += [np.sort(h.ground_indices(j)) for j in range(len(chunk))]
        mine_ng += [np.sort(h.nonground_indices(j)) for j in range(len(chunk))]
    one = pwpp_hip.Handle()  # and every eighth frame once more as a single frame (latency plan, two-pass binning)
    for i in range(0, n, 8):
        one.estimate_ground_batch([frames[i]], mode=pwpp_hip.MODE_FRESH)
        assert np.array_equal(np.sort(one.ground_indices(0)), mine[i]), "frame %d: single-frame and batch results differ" % i
    lib = ol.restatement()
    for i, f in enumerate(frames):  # the HIP path IS the contract: bit for bit on every frame
        ref = ol.Estimator(lib, arith=ol.ARITH_FXP).run(f)
        assert np.array_equal(mine[i], np.sort(ref.ground_idx)), "frame %d: HIP path and the restatement of its contract differ" % i
        # (round 5: the NON-ground lists too -- a two-part bin of nine blocks lost its last block's entries in the big-batch list kernel
        # while every ground set was right; tools/distinct_parity.py found it on 1024 varied frames)
        assert np.array_equal(mine_ng[i], np.sort(ref.nonground_idx)), "frame %d: non-ground list of the HIP path differs from the restatement's" % i
    rep = judge(frames, mine)
    rep["what"] = ("208 synthetic 64-beam frames (pwpp_synth.make_cloud: undulation 0-0.35 m, slopes, 10-70 boxes, sensor at 1.55-1.90 m, "
                   "default parameters and CZM), HIP path (batches of 104, fresh state) vs oracle/_ref's three builds of the reference")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "ref_consensus.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["product_equals_consensus"] >= 0.99 * rep["consensus_frames"], rep["consensus_misses"]
    # where the reference itself has no single answer the product is no further from exact arithmetic than the float builds
    assert all(d["product_vs_exact"] <= max(d["f32_vs_exact"], d["pk4_vs_exact"]) for d in rep["split_frames"]), rep["split_frames"]
