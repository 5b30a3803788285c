"""The HIP path held DIRECTLY against the reference's own source (the three builds under oracle/_ref: float sums in storage order, float
sums in a 4-lane order, double sums rounded once) -- not through the restatement -- on N varied 64-beam frames (pwpp_synth.varied_frame(first..),
fresh state, default parameters; default N = 2048, first = 20000: frames no other check of this repository has seen).  The reference
builds run first, in forked worker processes on the host cores; the frames then go through libpwpp_hip.so in batches of 1024.
Where the three builds agree on a frame's ground set the HIP result must be that set; where they split it must equal one of them.
   run on the GPU box:  python tools/hip_vs_reference.py [N] [first]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'patchwork-plusplus_amd/python'):
    sys.path.insert(0, os.path.join(ROOT, p))
import multiprocessing as mp
import numpy as np
import oracle_lib as ol
import pwpp_synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
FLAV = (("eigen_f32", ol.ARITH_EIGEN_F32), ("f32_packet4", ol.ARITH_F32_PACKET4), ("exact_f64", ol.ARITH_EXACT_F64))


def job(i):
    pts = pwpp_synth.varied_frame(FIRST + i)
    out = [pts]
    for _, a in FLAV:
        r = ol.Estimator(ol.reference(a), arith=a).run(pts)
        out.append((np.sort(r.ground_idx), np.sort(r.nonground_idx)))
    return out


if __name__ == "__main__":
    assert all(ol.reference(a) is not None for _, a in FLAV), "oracle/_ref/*.so did not travel"
    import pwpp_hip
    h = None
    unanimous = equal_unanimous = split = equal_some = worst = points = 0
    t0 = time.time()
    pool = mp.get_context("fork").Pool(min(64, max(1, (os.cpu_count() or 2) // 2)))  # (forked before this process touches the GPU)
    for c0 in range(0, N, 2048):  # chunks of 2048 frames: the host holds one chunk of clouds at a time
        ref = pool.map(job, range(c0, min(N, c0 + 2048)), chunksize=4)
        if h is None:
            h = pwpp_hip.Handle()
        points += sum(c[0].shape[0] for c in ref)
        for b0 in range(0, len(ref), 1024):
            chunk = ref[b0:b0 + 1024]
            h.estimate_ground_batch([c[0] for c in chunk], mode=pwpp_hip.MODE_FRESH)
            for i, c in enumerate(chunk):
                g, ng = np.sort(h.ground_indices(i)), np.sort(h.nonground_indices(i))
                same = [np.array_equal(g, c[k][0]) and np.array_equal(ng, c[k][1]) for k in (1, 2, 3)]
                if all(np.array_equal(c[1][0], c[k][0]) for k in (2, 3)):
                    unanimous += 1
                    equal_unanimous += all(same)
                    if not all(same):
                        worst = max(worst, int(np.setxor1d(g, c[1][0]).size))
                        print("  frame %d: the builds agree, the HIP path differs by %d indices" % (FIRST + c0 + b0 + i, np.setxor1d(g, c[1][0]).size))
                else:
                    split += 1
                    equal_some += any(same)
        print("  ... %d frames, %.0f s" % (min(N, c0 + 2048), time.time() - t0), flush=True)
    pool.close()
    print("%d frames (pwpp_synth.varied_frame(%d..%d), %d points): the three builds of the reference are unanimous on %d, the HIP path returns exactly "
          "their ground and non-ground lists on %d (largest miss: %d indices); they split on %d, the HIP path equals one of them on %d; "
          "frames binned twice %d, frames with parts moved through the arena %d"
          % (N, FIRST, FIRST + N - 1, points, unanimous, equal_unanimous, worst, split, equal_some, h.redo_stats()[1], h.arena_stats()[0]))
    assert equal_unanimous == unanimous and equal_some == split
