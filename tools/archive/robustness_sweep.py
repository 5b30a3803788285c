"""Cloud shapes and batch sizes the plans and the one-pass capacities were not tuned on: 16- to 128-beam
synthetic sensors, 3 to 130 frames per batch.  Checks the partition property and replay agreement of every
frame and compares two frames per run with the CPU oracle (bit-exact).  run on the GPU box:
   python tools/robustness_sweep.py"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np
import oracle_lib as ol, pwpp_hip, pwpp_synth

oracle = ol.restatement()
def same(h, i, ref):
    return (np.array_equal(np.sort(h.ground_indices(i)), np.sort(ref.ground_idx)) and
            np.array_equal(h.normals(i), ref.normals, equal_nan=True))
cases = [(16, 1800), (32, 1024), (64, 900), (64, 2200), (128, 2400), (128, 3800)]
for beams, steps in cases:
    plain = len(sys.argv) > 1 and sys.argv[1] == "plain"
    src = [pwpp_synth.make_cloud(100 + k, beams=beams, azimuth_steps=steps) if plain else pwpp_synth.add_edge_cases(pwpp_synth.make_cloud(100 + k, beams=beams, azimuth_steps=steps), k) for k in range(3)]
    refs = [ol.Estimator(oracle, arith=ol.ARITH_FXP).run(p) for p in src] if oracle is not None else None
    for F in (1, 2, 3, 7, 40, 130):
        frames = [src[i % 3] for i in range(F)]
        h = pwpp_hip.Handle()
        t0 = time.perf_counter()
        h.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
        counts = h.all_counts()
        dt = time.perf_counter() - t0
        ok = all(tuple(counts[i, :3]) == tuple(counts[i % 3, :3]) and counts[i, 0] + counts[i, 1] + counts[i, 5] == frames[i].shape[0] for i in range(F))
        exact = refs is None or (same(h, 0, refs[0]) and same(h, F - 1, refs[(F - 1) % 3]))
        print("%3d beams x %4d steps (%6d pts) x %3d frames: properties %s, oracle %s, one-pass/redone %s, %.1f ms" %
              (beams, steps, src[0].shape[0], F, "ok" if ok else "FAIL", "bit-exact" if exact else "MISMATCH", h.one_pass_stats(), dt * 1e3))
        h.close()
