"""Every dispatch of one pipeline run (library kernels AND the runtime's fill/copy kernels), in order,
from a rocprofv3 --kernel-trace csv: trace_all.py <kernel_trace.csv> [run index]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_clear" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
lo = starts[k - 1] if k > 0 else 0
# from the end of the previous run's last library kernel to this run's last library kernel
seq = rows[lo:starts[k + 1]] if k + 1 < len(starts) else rows[lo:]
prev_bin = seq[0]
t0 = None
for r in seq[1:] + []:
    pass
first = None
for i, r in enumerate(seq):
    if i == 0: continue
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if first is None and "k_emit" in seq[i - 1]["Kernel_Name"]: first = s
    if first is None: continue
    print("%8.2f us  +%6.2f us  %s" % ((s - first) / 1000.0, (e - s) / 1000.0, name[:50]))
