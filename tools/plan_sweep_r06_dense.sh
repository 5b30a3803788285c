#!/bin/bash
# big-bin class of the plan on dense 128-beam frames (BASELINE.json configs[4]) under contract v4
for r in 1 2; do
for plan in "W16:1023,W64.2:65535" "W16:1023,W64.4:65535" "W16:1023,W64.8:65535"; do
  PWPP_FIT_PLAN="$plan" python bench.py --workload dense --frames 512 --steps 12 --warmup 3 --no-cpu-baseline --skip-latency --skip-extras --profile-steps 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-28s %7.0f f/s %6.3f ms  sync %.3f  '%('$plan' or 'default',d['value'],d['ms_per_step'],d['synchronous']['ms_per_step'])+' '.join('%s=%.3f'%(n.replace('k_',''),v) for n,v in k.items() if v>0.01))"
done; done
