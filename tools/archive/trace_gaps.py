"""Per-kernel durations and the idle gaps between consecutive kernels of the single-frame pipeline,
from a rocprofv3 --kernel-trace csv (usage: trace_gaps.py <kernel_trace.csv> [launches to skip])."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a pipeline run starts with k_clear (k_czm_bin* in builds before it existed)
head = "k_clear" if any("k_clear" in r["Kernel_Name"] for r in rows) else "k_czm_bin"
runs, cur = [], []
for r in rows:
    if head in r["Kernel_Name"] and cur:
        runs.append(cur); cur = []
    cur.append(r)
runs.append(cur)
print("runs:", len(runs))
runs = runs[int(sys.argv[2]) if len(sys.argv) > 2 else 5:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for run in runs:
    for i, r in enumerate(run):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        dur[(i, name)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if i:
            gap[(i, name)].append(int(r["Start_Timestamp"]) - int(run[i - 1]["End_Timestamp"]))
tot_d = tot_g = 0
for k in sorted(dur):
    d = sorted(dur[k])[len(dur[k]) // 2] / 1000.0
    g = sorted(gap[k])[len(gap[k]) // 2] / 1000.0 if k in gap else 0.0
    tot_d += d; tot_g += g
    print("%2d %-34s n=%4d  kernel %7.2f us   gap before %6.2f us" % (k[0], k[1][:34], len(dur[k]), d, g))
print("sum kernels %.1f us, sum gaps %.1f us" % (tot_d, tot_g))
