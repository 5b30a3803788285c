#!/bin/bash
# the class boundary of the big-batch plan again, on the round's last build (both fit kernels have changed since plan_sweep_r06.sh)
for r in 1 2 3; do
for plan in "" "W16:767,W64.8:65535" "W16:1279,W64.8:65535" "W16:1535,W64.8:65535" "W16:511,W64.8:65535"; do
  PWPP_FIT_PLAN="$plan" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --skip-latency --skip-extras --profile-steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-28s %7.0f f/s %6.3f ms  '%('$plan' or 'default (W16:1023,W64.8)',d['value'],d['ms_per_step'])+' '.join('%s=%.3f'%(n.replace('k_',''),v) for n,v in k.items() if v>0.01))"
done; done
