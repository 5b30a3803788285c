#!/bin/bash
# fit-plan sweep on the bench workload (1024 KITTI frames): frames/s and the fit kernels' times per plan
# usage (GPU box): tools/plan_sweep_r04.sh "plan1" "plan2" ...   ("" = the default plan)
for plan in "$@"; do
  PWPP_FIT_PLAN="$plan" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --skip-extras --skip-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-34s %8.0f f/s  %.3f ms/step   fits %s' % ('$plan' or '(default)', d['value'], d['ms_per_step'], ' '.join('%s=%.3f'%(n,v) for n,v in k.items() if 'fit' in n and v>0.02)))"
done
