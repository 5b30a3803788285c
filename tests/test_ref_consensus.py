"""The KITTI-only pin, widened: 208 realistic 64-beam frames -- undulating terrain, slopes, boxes, reflected
noise, the sensor mounted between 1.55 and 1.90 m while the parameters keep the default 1.723 m, default CZM -- through the HIP
path and through ALL THREE builds of the reference's own patchworkpp.cpp under oracle/_ref (float sums in storage order, float
sums in a 4-lane order, double sums rounded once).

Where the three builds of the reference agree on a frame's ground set -- i.e. where the reference's result does not hang on the
summation order Eigen happens to be compiled with -- the product is held against that set; where they disagree among themselves
there is no single "reference result", those frames are counted and the product is held against the build nearest to it.

Contract v4 (round 6: exact integer moments on a 2^-30 m grid, on which every float of magnitude >= 2^-7 m lies): the sums of a
fit of 4+ points ARE the sums of exact arithmetic on the reference's floats, and the product equals the unanimous reference on
EVERY unanimous frame -- 204 of 204 here, 10 032 of 10 032 in tools/parity_statistics.py (profiles/r06_parity_statistics_10k.json;
rounds 3-5's 2^-21 m grid missed eighteen of those by 1-31 indices, frame 6 of this set among them).  So the pin is exact now:
ZERO misses on these 208 frames, bit-equality of the HIP path with the restatement on every frame, and on the four split frames
(19, 137, 144, 185: float builds 129 / 8 / 1 / 1 indices from exact arithmetic) the product equals the exact build.  Where the
builds split because a fit set of 1-3 points is summed in float by the reference (determinate there) and in double by the exact
build, the product follows the reference's floats: equal to BOTH float builds (see the 10 400-frame report).
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
            # (contract v4: never -- 0 misses over 10 032 unanimous frames, profiles/r06_parity_statistics_10k.json)
            assert d == 0, "frame %d: the three reference builds agree, the product differs by %d indices" % (i, d)
            rep["product_equals_consensus"] += d == 0
            if d:
                rep["consensus_misses"].append({"frame": i, "indices": d, "points": int(pts.shape[0])})
        else:
            d = {"frame": i,
                 "f32_vs_exact": int(np.setxor1d(ref["eigen_f32"], ref["exact_f64"]).size),
                 "pk4_vs_exact": int(np.setxor1d(ref["f32_packet4"], ref["exact_f64"]).size),
                 "f32_vs_pk4": int(np.setxor1d(ref["eigen_f32"], ref["f32_packet4"]).size),
                 "product_vs_exact": int(np.setxor1d(mine, ref["exact_f64"]).size),
                 "product_vs_f32": int(np.setxor1d(mine, ref["eigen_f32"]).size),
                 "product_vs_pk4": int(np.setxor1d(mine, ref["f32_packet4"]).size)}
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
    assert all(min(d["product_vs_exact"], max(d["product_vs_f32"], d["product_vs_pk4"])) == 0 for d in rep["split_frames"]), rep
    assert rep["consensus_misses"] == [] and rep["product_equals_consensus"] == rep["consensus_frames"], rep["consensus_misses"]  # (frame 6 was v3's miss)
    # the witness: rounds 3-5's 2^-21 m grid does miss frame 6, by one index
    v3 = np.sort(ol.Estimator(lib, arith=ol.ARITH_FXP21).run(frames[6]).ground_idx)
    assert np.setxor1d(v3, mine[6]).size == 1


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
    # the pinned outcome of this deterministic set: every unanimous frame exact, the four split frames are these and
    # the product equals the exact build on each of them
    assert rep["consensus_frames"] == 204 and rep["product_equals_consensus"] == 204 and rep["consensus_misses"] == [], rep["consensus_misses"]
    assert [d["frame"] for d in rep["split_frames"]] == [19, 137, 144, 185], rep["split_frames"]
    assert all(d["product_vs_exact"] == 0 for d in rep["split_frames"]), rep["split_frames"]


def _reference_job(i):
    pts = pwpp_synth.varied_frame(40000 + i)
    out = [pts]
    for _, arith in FLAVOURS:
        r = ol.Estimator(ol.reference(arith), arith=arith).run(pts)
        out.append((np.sort(r.ground_idx), np.sort(r.nonground_idx)))
    return out


@pytest.mark.gpu
def test_hip_path_directly_against_the_reference_builds_on_unseen_frames():
    """No restatement in between: 384 varied frames no other test uses, through the three builds of the reference's
    own source (forked workers) and through libpwpp_hip.so in one batch.  Unanimous builds -> the HIP path returns exactly their ground
    AND non-ground lists; split builds -> it equals one of them.  (tools/hip_vs_reference.py is the same on 16 384 frames:
    15 892 / 15 892 and 492 / 492, profiles/r06_hip_vs_reference_16384_frames.txt.)"""
    import multiprocessing as mp
    n = 384
    with mp.get_context("fork").Pool(min(32, max(1, (os.cpu_count() or 2) // 2))) as pool:
        ref = pool.map(_reference_job, range(n), chunksize=4)
    import pwpp_hip
    h = pwpp_hip.Handle()
    h.estimate_ground_batch([c[0] for c in ref], mode=pwpp_hip.MODE_FRESH)
    unanimous = split = 0
    for i, c in enumerate(ref):
        g, ng = np.sort(h.ground_indices(i)), np.sort(h.nonground_indices(i))
        same = [np.array_equal(g, c[k][0]) and np.array_equal(ng, c[k][1]) for k in (1, 2, 3)]
        if all(np.array_equal(c[1][0], c[k][0]) for k in (2, 3)):
            unanimous += 1
            assert all(same), "frame %d: the three builds of the reference agree, the HIP path differs by %d indices" % (40000 + i, np.setxor1d(g, c[1][0]).size)
        else:
            split += 1
            assert any(same), "frame %d: the builds split and the HIP path equals none of them" % (40000 + i)
    assert unanimous >= 0.9 * n and unanimous + split == n
