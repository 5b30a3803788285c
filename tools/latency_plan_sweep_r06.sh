#!/bin/bash
# single-frame latency (configs[1]) by fit plan under contract v4: min / median / max over the six KITTI sources, GPU us
for plan in "" "H64:511" "H64:2047" "H64:255" "B64:65535" "S64:65535"; do
  PWPP_FIT_PLAN="$plan" python bench.py --steps 10 --warmup 2 --frames 128 --no-cpu-baseline --skip-extras --no-profile-events 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); l=d['latency']
print('%-14s min %.1f median %.1f max %.1f   '%('$plan' or 'default', l['gpu_us_min'], l['gpu_us_median'], l['gpu_us_max']) + ' '.join('%.0f'%r['gpu_us'] for r in l['by_source']))"
done
