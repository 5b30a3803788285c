cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ordp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ordp -o s -- python $GRAFT_REPO_ROOT/tools/order_cost.py 1024 > /tmp/ord.log 2>&1
tail -2 /tmp/ord.log
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/ordp/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us', r['Percentage'])
PY
