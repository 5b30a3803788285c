"""The stateful form of tools/hip_vs_reference.py: S long-lived sensors (streams) over L frames each -- stream s sees
pwpp_synth.varied_frame(FIRST + 1000 s + t) at step t -- through ONE long-lived object per stream and build of the reference
(oracle/_ref: three builds, forked workers on the host cores) and through libpwpp_hip.so as S streams in lock-step
(PWPP_MODE_STREAMS).  At every step of every stream: where the three builds agree on the ground set the HIP path must return it, where
they split it must equal one of them; the sensor heights are compared at the end.
   run on the GPU box:  python tools/hip_vs_reference_streams.py [S] [L] [FIRST = 500000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'patchwork-plusplus_amd/python'):
    sys.path.insert(0, os.path.join(ROOT, p))
import multiprocessing as mp
import numpy as np
import oracle_lib as ol
import pwpp_synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
FIRST = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
FLAV = (("eigen_f32", ol.ARITH_EIGEN_F32), ("f32_packet4", ol.ARITH_F32_PACKET4), ("exact_f64", ol.ARITH_EXACT_F64))


def stream_job(s):
    est = [ol.Estimator(ol.reference(a), arith=a) for _, a in FLAV]
    rows = []
    for t in range(L):
        pts = pwpp_synth.varied_frame(FIRST + 1000 * s + t)
        rows.append([np.sort(e.run(pts).ground_idx) for e in est])
    return rows, [float(e._l.lib.pwo_get_height(e._h)) for e in est]


if __name__ == "__main__":
    assert all(ol.reference(a) is not None for _, a in FLAV), "oracle/_ref/*.so did not travel"
    t0 = time.time()
    with mp.get_context("fork").Pool(min(64, max(1, (os.cpu_count() or 2) // 2))) as pool:
        ref = pool.map(stream_job, range(S), chunksize=1)
    print("%d streams x %d frames through the three builds of the reference in %.1f s" % (S, L, time.time() - t0))
    import pwpp_hip
    h = pwpp_hip.Handle()
    h.set_num_streams(S)
    unanimous = equal_unanimous = split = equal_some = 0
    for t in range(L):
        h.estimate_ground_batch([pwpp_synth.varied_frame(FIRST + 1000 * s + t) for s in range(S)], mode=pwpp_hip.MODE_STREAMS)
        for s in range(S):
            g = np.sort(h.ground_indices(s))
            sets = ref[s][0][t]
            same = [np.array_equal(g, x) for x in sets]
            if np.array_equal(sets[0], sets[1]) and np.array_equal(sets[0], sets[2]):
                unanimous += 1
                equal_unanimous += all(same)
                if not all(same):
                    print("  stream %d step %d: the builds agree, the HIP path differs by %d indices" % (s, t, np.setxor1d(g, sets[0]).size))
            else:
                split += 1
                equal_some += any(same)
                if not any(same):
                    print("  stream %d step %d: the builds split (%d / %d indices between them), the HIP path is %d indices from the nearest"
                          % (s, t, np.setxor1d(sets[0], sets[2]).size, np.setxor1d(sets[1], sets[2]).size, min(np.setxor1d(g, x).size for x in sets)))
    dh = max(abs(h.state(s).sensor_height - ref[s][1][2]) for s in range(S))
    dh_f32 = max(abs(ref[s][1][0] - ref[s][1][2]) for s in range(S))
    print("%d streams x %d steps: the three builds are unanimous on %d stream-steps, the HIP path returns exactly their ground set on %d; they split on %d, "
          "the HIP path equals one of them on %d; final sensor heights within %.2e m of the exact build's (the float build: %.2e m); frames binned twice %d"
          % (S, L, unanimous, equal_unanimous, split, equal_some, dh, dh_f32, h.redo_stats()[1]))
