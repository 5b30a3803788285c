cd /tmp && export TMPDIR=/tmp
for ord in 012345 432105 210345 342105; do
PWPP_FIT_ORDER=$ord rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ord$ord -o o -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-events --skip-latency > /dev/null 2>&1
ORD=$ord python3 - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ord'+os.environ['ORD']+'/o_kernel_trace.csv')))
rows=[r for r in rows if 'k_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=rows[-11:]
tot=(int(last[-1]['End_Timestamp'])-int(last[0]['Start_Timestamp']))/1e3
print(os.environ['ORD'],"pipeline_us=%.0f"%tot," ".join("%s=%.0f"%(r['Kernel_Name'].split('::')[-1][:14],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in last))
PY
done
