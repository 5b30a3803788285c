cd /tmp && export TMPDIR=/tmp
PWPP_FIT_PLAN=$1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-events --skip-latency > /dev/null 2>&1
python3 - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/tr/t_kernel_trace.csv')))
rows=[r for r in rows if 'k_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last pipeline pass: from last k_czm_bin
idx=max(i for i,r in enumerate(rows) if 'k_czm_bin' in r['Kernel_Name'])
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    print("%-38s start=%8.1f dur=%8.1f grid=%s,%s"%(r['Kernel_Name'].split('::')[-1][:38],(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r['Grid_Size_X'],r['Grid_Size_Y']))
PY
