import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'patchwork-plusplus_amd/python')
import numpy as np, pwpp_hip, pwpp_synth
os.environ["PWPP_NO_ONE_PASS"] = "1"
for beams, steps in ((16, 1800), (64, 900)):
    src = [pwpp_synth.make_cloud(100 + k, beams=beams, azimuth_steps=steps) for k in range(3)]
    h = pwpp_hip.Handle()
    h.estimate_ground_batch(src * 3, mode=pwpp_hip.MODE_FRESH)
    mx = max(s.shape[0] for s in src); mn = mx + mx // 8
    for i in range(3):
        rec = h.patch_records(i); c = h.all_counts()[i]
        print(beams, steps, "frame", i, "n", src[i].shape[0], "rnr", c[3], "oor", c[4], "dropped", c[5], "pseudo cap", mn // 8 + 64,
              "max patch", rec["n_points"].max(), "bin", rec["bin"][rec["n_points"].argmax()], "zone-0 cap", int(4 * mn / 32 + 64))
