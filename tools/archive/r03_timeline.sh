#!/bin/bash
# timeline of ONE step of the default schedule (rocprofv3 kernel trace): start / duration of every launch relative to the step's k_clear
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_tl
rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile-events --skip-latency --skip-extras "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv,os,glob
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03_tl/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
clears=[i for i,r in enumerate(rows) if 'k_clear' in r['Kernel_Name']]
a=clears[-3]; b=clears[-2]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')
    print("%-26s q=%-3s start=%8.1f end=%8.1f dur=%7.1f"%(n[:26],r.get('Queue_Id','?'),(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
print("step span us:", (max(int(r['End_Timestamp']) for r in rows[a:b])-t0)/1e3, " next step starts at", (int(rows[b]['Start_Timestamp'])-t0)/1e3)
PY
