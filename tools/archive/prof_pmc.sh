#!/bin/bash
# usage: tools/prof_pmc.sh <tag> <counters...>   (run on the GPU box via gpurun)
# one rocprofv3 --pmc pass over a short bench run; per-kernel averages printed and saved
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-events --skip-latency > $out.log 2>&1
python3 - "$out" <<'PY'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
if not f: print("no counter csv", glob.glob(sys.argv[1]+'/**/*',recursive=True)[:20]); sys.exit(0)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:60]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k in acc:
    n=len(cnt[k])
    print("%-62s n=%4d "%(k,n)+" ".join("%s=%.4g"%(c,v/n) for c,v in sorted(acc[k].items())))
PY
