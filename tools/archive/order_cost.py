#!/usr/bin/env python3
"""What the reference-order mode (pwpp_set_output_order) costs: GPU time of a 256-frame KITTI batch in both modes."""
import lzma, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "patchwork-plusplus_amd", "python"))
import pwpp_hip

kitti = [np.frombuffer(lzma.open(os.path.join(HERE, "..", "tests", "golden", "kitti_%06d.bin.xz" % k)).read(), np.float32).reshape(-1, 4) for k in range(6)]
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = [kitti[k % 6] for k in range(F)]
for ref in (False, True):
    h = pwpp_hip.Handle()
    h.set_output_order(ref)
    t = []
    for rep in range(6):
        h.estimate_ground_batch(frames, mode=pwpp_hip.MODE_FRESH)
        t.append(h.time_us())
    print("%-9s order: %d frames  %.0f us  (%.0f frames/s on the device)" % ("reference" if ref else "scatter", F, min(t[1:]), F / (min(t[1:]) * 1e-6)))
