#!/bin/bash
# Round profile (run on the GPU box through gpurun): rocprofv3 kernel stats of the default bench command, the SQ
# counters of VERDICT r01 item 5 in one pass (8 SQ slots), FETCH_SIZE and WRITE_SIZE in their own passes, and the
# calibration of those two counters for 4 / 8 / 16-byte accesses (tools/ubench/fetch_calib).
# usage: tools/profile_r05.sh <tag> [bench args...]
R=${1:-r05}; shift
ARGS="$@"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --skip-latency --skip-extras --in-flight 1 --no-overlap $ARGS"
D="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --skip-latency --skip-extras $ARGS"   # the default command: two batches in flight
timeout 110 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $B --steps 5 --warmup 2 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log < /dev/null
timeout 110 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o s -- $D --steps 10 --warmup 2 > $OUT/bench_default_under_rocprof.json 2> $OUT/stats_default.log < /dev/null
timeout 110 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -o q -- $B --steps 2 --warmup 1 --no-profile-events > /dev/null 2> $OUT/sq.log < /dev/null
timeout 110 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- $B --steps 2 --warmup 1 --no-profile-events > /dev/null 2> $OUT/fetch.log < /dev/null
timeout 110 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- $B --steps 2 --warmup 1 --no-profile-events > /dev/null 2> $OUT/write.log < /dev/null
timeout 110 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calf -o c -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib > $OUT/calib.txt 2> $OUT/calf.log < /dev/null
timeout 110 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calw -o c -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib > /dev/null 2> $OUT/calw.log < /dev/null
python3 - "$OUT" <<'PY'
import csv,sys,glob,collections,json,re
out=sys.argv[1]
def pmc(d):
    f=glob.glob(out+'/'+d+'/**/*counter_collection.csv',recursive=True)
    if not f: return {}
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name']; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    return {k:{c:v/len(cnt[k]) for c,v in acc[k].items()} for k in acc}
def short(k): return re.sub(r'\(anonymous namespace\)::','',k).split('(')[0].replace('void ','')
cal_f=pmc('calf'); cal_w=pmc('calw')
calib={}
GiB=float(1<<30)
for k,v in cal_f.items():
    s=short(k)
    if 'k_read' in s: calib[s+' FETCH_SIZE_kb_per_GiB_read']=v.get('FETCH_SIZE',0)/( (GiB//12*12)/GiB if '4_8' in s else 1.0)
for k,v in cal_w.items():
    s=short(k)
    if 'k_write' in s: calib[s+' WRITE_SIZE_kb_per_256MiB_written']=v.get('WRITE_SIZE',0)
sq=pmc('sq'); fe=pmc('fetch'); wr=pmc('write')
res={}
for k in sorted(set(sq)|set(fe)|set(wr)):
    s=short(k)
    if not s.startswith('k_'): continue
    e=dict(sq.get(k,{}))
    e['FETCH_SIZE_kb_raw']=fe.get(k,{}).get('FETCH_SIZE',0.0)
    e['WRITE_SIZE_kb_raw']=wr.get(k,{}).get('WRITE_SIZE',0.0)
    res[s]=e
json.dump({'calibration':calib,'kernels':res},open(out+'/pmc_summary.json','w'),indent=1)
print(json.dumps(calib,indent=1))
for s,e in res.items():
    print('%-26s'%s,' '.join('%s=%.4g'%(c,v) for c,v in sorted(e.items())))
st=glob.glob(out+'/stats/**/*kernel_stats.csv',recursive=True)
print(open(st[0]).read() if st else "no stats csv")
st=glob.glob(out+'/stats_default/**/*kernel_stats.csv',recursive=True)
print(open(st[0]).read() if st else "no default stats csv")
PY
