#!/bin/bash
# SQ counters of the fit kernels under a given plan: tools/sq_counters.sh <tag> [bench args]   (PWPP_FIT_PLAN from the environment)
R=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$R
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -o q -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --skip-latency --steps 2 --warmup 1 --no-profile-events "$@" > /dev/null 2> $OUT/log.txt
python3 - "$OUT" <<'PY'
import csv,sys,glob,collections,re
out=sys.argv[1]
f=glob.glob(out+'/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
kt=glob.glob(out+'/**/*kernel_trace.csv',recursive=True)[0]
dur=collections.defaultdict(list)
for r in csv.DictReader(open(kt)): dur[r['Kernel_Name']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k in acc:
    s=re.sub(r'\(anonymous namespace\)::','',k).split('(')[0].replace('void ','')
    if 'fit' not in s and 'czm_bin' not in s and 'emit' not in s: continue
    n=len(cnt[k]); v={c:x/n for c,x in acc[k].items()}
    print('%-22s ms(under pmc)=%.3f INSTS_VALU=%.3e ACTIVE_VALU=%.3e (%.0f%% of 1024 SIMDs x kernel) WAVE_CYC=%.3e ACTIVE_ANY=%.0f%% WAIT_ANY=%.0f%% WAIT_INST=%.0f%% waves=%d'%(
      s,sum(dur[k])/len(dur[k]),v['SQ_INSTS_VALU'],v['SQ_ACTIVE_INST_VALU'],100*v['SQ_ACTIVE_INST_VALU']*4/(1024*sum(dur[k])/len(dur[k])*1e-3*2.4e9),v['SQ_WAVE_CYCLES'],
      100*v['SQ_ACTIVE_INST_ANY']/v['SQ_WAVE_CYCLES'],100*v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES'],100*v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES'],v['SQ_WAVES']))
PY
