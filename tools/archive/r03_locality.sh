#!/bin/bash
# round 3: does a smaller per-wave working set turn the fit kernels' re-reads into cache hits?  Per plan: kernel times of the
# single-stream schedule and FETCH_SIZE per launch (x2 = bytes on gfx950).   usage: tools/r03_locality.sh "<plan>" ...
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_locality
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for plan in "$@"; do
  i=$((i+1))
  export PWPP_FIT_PLAN="$plan"
  echo "=== plan $plan"
  python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --skip-latency --no-overlap --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fps %.0f ms %.3f'%(d['value'],d['ms_per_step'])); print({k:round(v,3) for k,v in d['kernel_ms'].items() if v>0.01})"
  rm -rf $OUT/f$i
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f$i -o f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --skip-latency --no-overlap --steps 2 --warmup 1 --no-profile-events > /dev/null 2> $OUT/f$i.log
  python3 - $OUT/f$i <<'PY'
import csv,sys,glob,collections,re
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(float); cnt=collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']).split('(')[0].replace('void ','')
    acc[k]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k in acc:
    if k.startswith('k_fit') or k.startswith('k_czm_bin_sc') or k.startswith('k_emit'): print('  %-24s n=%3d fetch GB/launch %.3f'%(k,len(cnt[k]),2*acc[k]/len(cnt[k])*1024/1e9))
PY
  rm -rf $OUT/f$i
done
