"""S stateful streams in lock-step over T steps of DISTINCT varied frames (stream s, step t sees pwpp_synth.varied_frame(200000 + 1000 s + t)),
cold handle, against S sequential runs of the restatement: ground / non-ground lists, normals, sensor height, thresholds of every stream at
every step.  S > 64: the big-batch kernels with stream state; the first steps bin some frames again (per-stream redo with state restore).
   run on the GPU box:  python tools/varied_streams_parity.py [S] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'patchwork-plusplus_amd/python'):
    sys.path.insert(0, os.path.join(ROOT, p))
import multiprocessing as mp
import numpy as np
import oracle_lib as ol
import pwpp_synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 72
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def stream_job(s):
    est = ol.Estimator(ol.restatement(), arith=ol.ARITH_FXP)
    out = []
    for t in range(T):
        pts = pwpp_synth.varied_frame(200000 + 1000 * s + t)
        r = est.run(pts)
        out.append((pts, np.sort(r.ground_idx), np.sort(r.nonground_idx), r.normals.copy(), r.sensor_height, list(r.elevation_thr), list(r.flatness_thr)))
    return out


if __name__ == "__main__":
    t0 = time.time()
    ol.build()
    with mp.get_context("fork").Pool(min(64, max(1, (os.cpu_count() or 2) // 2))) as pool:
        ref = pool.map(stream_job, range(S), chunksize=1)
    print("%d streams x %d steps through the restatement in %.1f s" % (S, T, time.time() - t0))
    import pwpp_hip
    h = pwpp_hip.Handle()
    h.set_num_streams(S)
    bad = 0
    for t in range(T):
        h.estimate_ground_batch([ref[s][t][0] for s in range(S)], mode=pwpp_hip.MODE_STREAMS)
        for s in range(S):
            _, g, ng, nm, sh, et, ft = ref[s][t]
            st = h.state(s)
            ok = (np.array_equal(np.sort(h.ground_indices(s)), g) and np.array_equal(np.sort(h.nonground_indices(s)), ng) and
                  np.array_equal(h.normals(s), nm, equal_nan=True) and st.sensor_height == sh and list(st.elevation_thr) == et and list(st.flatness_thr) == ft)
            if not ok:
                bad += 1
                if bad <= 5:
                    print("  stream %d step %d differs" % (s, t))
        print("step %d: %d mismatches so far; (frames, frames redone) %s" % (t, bad, h.redo_stats()))
    assert bad == 0
    print("%d streams x %d steps of distinct varied frames: identical to the restatement" % (S, T))
