"""The two checks that found round 5's only real bug, as part of the -m gpu suite (until now builder-run scripts,
tools/distinct_parity.py and tools/varied_streams_parity.py):

* 1024 DISTINCT varied frames in one batch through a COLD handle (the frames it has to bin again after a segment overflow included)
  and once more through the then WARM handle: ground list, non-ground list and plane normals of every frame against the CPU
  restatement of the contract, bit for bit;
* 72 stateful streams in lock-step over 8 steps of distinct frames (the big-batch kernels with stream state; the first steps bin some
  frames again, per-stream redo with state restore): lists, normals, sensor height and thresholds of every stream at every step
  against 72 sequential runs of the restatement.

The restatement runs first, in a pool of forked workers, before this process touches the GPU."""
import multiprocessing as mp
import os

import numpy as np
import pytest

import oracle_lib as ol
import pwpp_synth

pytestmark = pytest.mark.gpu

N_DISTINCT = int(os.environ.get("PWPP_TEST_DISTINCT_FRAMES", "1024"))
STREAMS, STEPS = 72, 8


def _workers():
    return min(64, max(1, (os.cpu_count() or 2) // 2))


def _frame_job(i):
    pts = pwpp_synth.varied_frame(i)
    r = ol.Estimator(ol.restatement(), arith=ol.ARITH_FXP).run(pts)
    return pts, np.sort(r.ground_idx), np.sort(r.nonground_idx), r.normals.copy()


def _stream_job(s):
    est = ol.Estimator(ol.restatement(), arith=ol.ARITH_FXP)
    out = []
    for t in range(STEPS):
        pts = pwpp_synth.varied_frame(200000 + 1000 * s + t)
        r = est.run(pts)
        out.append((pts, np.sort(r.ground_idx), np.sort(r.nonground_idx), r.normals.copy(), r.sensor_height, list(r.elevation_thr), list(r.flatness_thr)))
    return out


def test_1024_distinct_frames_cold_then_warm(oracle_built):
    with mp.get_context("fork").Pool(_workers()) as pool:
        ref = pool.map(_frame_job, range(N_DISTINCT), chunksize=4)
    import pwpp_hip
    frames = [r[0] for r in ref]
    h = pwpp_hip.Handle()
    redone = []
    for attempt in ("cold", "warm"):
        h.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
        bad = []
        for i in range(N_DISTINCT):
            ok = (np.array_equal(np.sort(h.ground_indices(i)), ref[i][1]) and np.array_equal(np.sort(h.nonground_indices(i)), ref[i][2]) and
                  np.array_equal(h.normals(i), ref[i][3], equal_nan=True))
            if not ok:
                bad.append(i)
        assert not bad, "%s handle: %d of %d frames differ from the restatement (first: %s)" % (attempt, len(bad), N_DISTINCT, bad[:8])
        redone.append(h.redo_stats()[1])
    assert redone[1] == redone[0], "the warm handle binned frames again: %s" % (redone,)  # (the cold pass sized the segments)
    assert h.clamped_frames() == 0 and h.fixed_up_frames() == 0


def test_72_varied_streams_over_8_steps(oracle_built):
    with mp.get_context("fork").Pool(_workers()) as pool:
        ref = pool.map(_stream_job, range(STREAMS), chunksize=1)
    import pwpp_hip
    h = pwpp_hip.Handle()
    h.set_num_streams(STREAMS)
    for t in range(STEPS):
        h.estimate_ground_batch([ref[s][t][0] for s in range(STREAMS)], mode=pwpp_hip.MODE_STREAMS)
        for s in range(STREAMS):
            _, g, ng, nm, sh, et, ft = ref[s][t]
            st = h.state(s)
            assert np.array_equal(np.sort(h.ground_indices(s)), g) and np.array_equal(np.sort(h.nonground_indices(s)), ng), "stream %d step %d: lists differ" % (s, t)
            assert np.array_equal(h.normals(s), nm, equal_nan=True), "stream %d step %d: normals differ" % (s, t)
            assert st.sensor_height == sh and list(st.elevation_thr) == et and list(st.flatness_thr) == ft, "stream %d step %d: adaptive state differs" % (s, t)
