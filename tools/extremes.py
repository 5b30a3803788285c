import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, pwpp_hip, pwpp_synth
import oracle_lib as ol
oracle = ol.restatement()
rng = np.random.default_rng(3)
base = pwpp_synth.make_cloud(5, beams=64, azimuth_steps=2000)
# (a) one frame at the size limit: 4 194 304 points (a 64-beam cloud repeated with jitter)
reps = 4194304 // base.shape[0] + 1
big = np.concatenate([base + rng.normal(0, 0.01, base.shape).astype(np.float32) for _ in range(reps)])[:4194304]
h = pwpp_hip.Handle()
t0 = time.perf_counter(); h.estimate_ground_batch([big], mode=pwpp_hip.MODE_FRESH); c = h.all_counts()[0]; dt = time.perf_counter() - t0
print("4.19 M points in one frame: ground %d nonground %d patches %d dropped %d -> partition %s, %.1f ms" % (c[0], c[1], c[2], c[5], c[0] + c[1] + c[5] == big.shape[0], dt * 1e3))
ref = ol.Estimator(oracle, arith=ol.ARITH_FXP).run(big)
print("   oracle: sets equal %s, normals equal %s" % (np.array_equal(np.sort(h.ground_indices(0)), np.sort(ref.ground_idx)), np.array_equal(h.normals(0), ref.normals, equal_nan=True)))
# (b) one large frame among many tiny ones
tiny = [base[rng.choice(base.shape[0], 800, replace=False)] for _ in range(60)]
mix = tiny[:30] + [big[:2000000]] + tiny[30:]
h2 = pwpp_hip.Handle()
t0 = time.perf_counter(); h2.estimate_ground_batch(mix, mode=pwpp_hip.MODE_FRESH); cc = h2.all_counts(); dt = time.perf_counter() - t0
ok = all(cc[i, 0] + cc[i, 1] + cc[i, 5] == mix[i].shape[0] for i in range(len(mix)))
print("61 frames (60 x 800 points + 1 x 2 M points): partition %s, one-pass/redone %s, %.1f ms" % (ok, h2.one_pass_stats(), dt * 1e3))
r30 = ol.Estimator(oracle, arith=ol.ARITH_FXP).run(mix[30]); r5 = ol.Estimator(oracle, arith=ol.ARITH_FXP).run(mix[5])
print("   oracle: big frame %s, tiny frame %s" % (np.array_equal(np.sort(h2.ground_indices(30)), np.sort(r30.ground_idx)), np.array_equal(np.sort(h2.ground_indices(5)), np.sort(r5.ground_idx))))
