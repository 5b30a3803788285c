"""The HIP path against the CPU restatement of its contract on N DISTINCT varied frames in ONE cold batch (default 1024): ground and
non-ground index sets and plane normals of every frame, bit for bit -- including the frames the cold handle has to bin again after a
segment overflow (per-frame redo), and once more on the warm handle (no redo).  The oracle results are computed first, by a pool of
forked workers, before this process touches the GPU.
   run on the GPU box:  python tools/distinct_parity.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tests', 'patchwork-plusplus_amd/python'):
    sys.path.insert(0, os.path.join(ROOT, p))
import multiprocessing as mp
import numpy as np
import oracle_lib as ol
import pwpp_synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
DENSE = "--dense" in sys.argv      # dense 128-beam ~480 k-point frames, 36-sector CZM (BASELINE.json configs[4]) instead of varied 64-beam ones
ORDERED = "--ordered" in sys.argv  # PWPP_ORDER_REFERENCE: the lists position by position against the restatement's (heights; ties move)


def dense_params(lib):
    p = lib.default_params()
    for k in range(4):
        p.num_sectors_each_zone[k] = 36
    return p


def oracle_job(i):
    pts = pwpp_synth.make_dense_cloud(7000 + i) if DENSE else pwpp_synth.varied_frame(i)
    lib = ol.restatement()
    r = ol.Estimator(lib, dense_params(lib) if DENSE else None, arith=ol.ARITH_FXP).run(pts)
    return pts, np.sort(r.ground_idx), np.sort(r.nonground_idx), r.normals.copy(), np.asarray(r.ground_idx), np.asarray(r.nonground_idx)


if __name__ == "__main__":
    t0 = time.time()
    ol.build()
    with mp.get_context("fork").Pool(min(64, max(1, (os.cpu_count() or 2) // 2))) as pool:
        ref = pool.map(oracle_job, range(N), chunksize=4)
    print("%d frames generated and put through the restatement in %.1f s" % (N, time.time() - t0))
    import pwpp_hip
    frames = [r[0] for r in ref]
    prm = pwpp_hip.default_params()
    if DENSE:
        for k in range(4):
            prm.num_sectors_each_zone[k] = 36
    h = pwpp_hip.Handle(prm)
    h.set_output_order(ORDERED)
    for attempt in ("cold handle", "warm handle"):
        h.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
        bad = 0
        for i in range(N):
            g, ng = np.sort(h.ground_indices(i)), np.sort(h.nonground_indices(i))
            ok = np.array_equal(g, ref[i][1]) and np.array_equal(ng, ref[i][2]) and np.array_equal(h.normals(i), ref[i][3], equal_nan=True)
            if ok and ORDERED:
                z = frames[i][:, 2]
                ok = np.array_equal(z[h.ground_indices(i)], z[ref[i][4]]) and np.array_equal(z[h.nonground_indices(i)], z[ref[i][5]])
            bad += 0 if ok else 1
            if not ok and bad <= 5:
                nm = h.normals(i)
                print("  frame %d differs: ground %d vs %d (xor %d), non-ground %d vs %d (xor %d), normals %s vs %s, equal %s, max |d| %s" % (
                    i, len(g), len(ref[i][1]), np.setxor1d(g, ref[i][1]).size, len(ng), len(ref[i][2]), np.setxor1d(ng, ref[i][2]).size,
                    nm.shape, ref[i][3].shape, nm.shape == ref[i][3].shape and np.array_equal(nm, ref[i][3], equal_nan=True),
                    float(np.nanmax(np.abs(nm - ref[i][3]))) if nm.shape == ref[i][3].shape else None))
        assert bad == 0, "%d frames differ" % bad
        print("%s: %d frames, %d differ from the restatement; one-pass stats (batches, batches with a redo) %s, (frames, frames redone) %s"
              % (attempt, N, bad, h.one_pass_stats(), h.redo_stats()))
