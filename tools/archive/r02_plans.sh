#!/bin/bash
# fit-plan candidates on the 1024-frame batch (PWPP_FIT_PLAN is read once, at pwpp_create)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O; : > $O/plans.txt
for plan in "$@"; do
  echo "== $plan" >> $O/plans.txt
  PWPP_FIT_PLAN=$plan python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-latency 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]), round(d[\"ms_per_step\"],3), {k:round(v,3) for k,v in d[\"kernel_ms\"].items() if v>0.01})" >> $O/plans.txt 2>&1
done
cat $O/plans.txt
