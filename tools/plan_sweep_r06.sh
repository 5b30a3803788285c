#!/bin/bash
# fit plans under contract v4 (wide sums: k_fit_w64<16,*> runs three waves per SIMD, so 4 x 1024 waves of 64 patches are 1.33 generations)
for r in 1 2; do
for plan in "" "W16.32:1023,W64.4:65535" "W16.16:1023,W64.4:65535" "W16.32:1023,W64.2:65535" "W16.32:511,W64.4:65535" "W16.32:2047,W64.4:65535"; do
  PWPP_FIT_PLAN="$plan" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --skip-latency --skip-extras --profile-steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-28s %7.0f f/s %6.3f ms  sync %.3f  '%('$plan' or 'default',d['value'],d['ms_per_step'],d['synchronous']['ms_per_step'])+' '.join('%s=%.3f'%(n.replace('k_',''),v) for n,v in k.items() if v>0.01))"
done; done

# (more of the same, run later in the round: eight and sixteen big bins per wave)
for r in 1 2; do
for plan in "" "W16:1023,W64.8:65535" "W16:2047,W64.8:65535" "W16:511,W64.8:65535"; do
  PWPP_FIT_PLAN="$plan" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --skip-latency --skip-extras --profile-steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-28s %7.0f f/s %6.3f ms  sync %.3f  '%('$plan' or 'default',d['value'],d['ms_per_step'],d['synchronous']['ms_per_step'])+' '.join('%s=%.3f'%(n.replace('k_',''),v) for n,v in k.items() if v>0.01))"
done; done
for r in 1 2; do
for plan in "" "W16:1023,W64.16:65535" "W16:1023,W64.4:65535"; do
  PWPP_FIT_PLAN="$plan" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --skip-latency --skip-extras --profile-steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-28s %7.0f f/s %6.3f ms  sync %.3f  '%('$plan' or 'default (W64.8)',d['value'],d['ms_per_step'],d['synchronous']['ms_per_step'])+' '.join('%s=%.3f'%(n.replace('k_',''),v) for n,v in k.items() if v>0.01))"
done; done
