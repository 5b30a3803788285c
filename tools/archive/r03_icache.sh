#!/bin/bash
# instruction-cache counters of the pipeline's kernels (single-stream schedule): requests, hits, misses per launch
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -o q -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --skip-latency --skip-extras --steps 2 --warmup 1 --no-profile-events --no-overlap "$@" > /dev/null 2> $OUT/log.txt
python3 - "$OUT" <<'PY'
import csv,sys,glob,collections,re
out=sys.argv[1]
f=glob.glob(out+'/**/*counter_collection.csv',recursive=True)
if not f: print(open(out+'/log.txt').read()[-2000:]); sys.exit(0)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name']; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k in acc:
    s=re.sub(r'\(anonymous namespace\)::','',k).split('(')[0].replace('void ','')
    if not s.startswith('k_'): continue
    n=len(cnt[k]); v={c:x/n for c,x in acc[k].items()}
    req=v.get('SQC_ICACHE_REQ',0); miss=v.get('SQC_ICACHE_MISSES',0)
    print('%-24s n=%3d '%(s,n)+' '.join('%s=%.3e'%(c.replace('SQC_',''),x) for c,x in sorted(v.items()))+ '  miss/req=%.3f'%(miss/req if req else 0))
PY
