#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a PMC summary (tools/profile_r05.sh): HBM-side bytes per launch over the whole batch,
2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE counts half the bytes on gfx950: calibration in the same summary), per kernel.
usage: tools/make_hbm_traffic.py profiles/r04_pmc_summary.json > profiles/hbm_traffic.json"""
import json
import sys

src = sys.argv[1]
d = json.load(open(src))
out = {}
for k, e in d["kernels"].items():
    name = k.replace(", ", ",").replace(",true>", ">").replace(",false>", ">").replace("<true>", "").replace("<false>", "")  # (the contract width is a template argument of every fit kernel)
    name = "k_fit_w64<64,p>" if name.startswith("k_fit_w64<64,") else name
    if not name.startswith("k_fit_w64"):
        name = name.split("<")[0]  # (the other kernels under their plain names, as bench.py's kernel_ms has them)
    out[name] = int(round((2.0 * e.get("FETCH_SIZE_kb_raw", 0.0) + e.get("WRITE_SIZE_kb_raw", 0.0)) * 1024.0))
out["_total"] = sum(v for k, v in out.items() if not k.startswith("_") and k != "k_czm_bin" and k != "k_clear")
out["source"] = src + " (tools/profile_r06.sh r06: bench.py --in-flight 1 --no-overlap: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py's 1024-frame KITTI batch)"
out["_note"] = ("HBM-side bytes per launch over the whole 1024-frame batch, single-stream schedule: (2 x FETCH_SIZE + WRITE_SIZE) x 1024; FETCH_SIZE counts "
                "half the bytes on gfx950 for 4-, 8- and 16-byte loads alike (calibration in the source file).  k_fit_w64<64,p> = the big-bin class of the plan "
                "(p = 4).  _total = the kernels of one step (k_czm_bin and k_clear only run in the histogram probe / on a change of shape).  bench.py copies "
                "the dominant kernel's entry into roofline.traffic (a PMC pass cannot run inside the timed process).")
print(json.dumps(out, indent=1))
