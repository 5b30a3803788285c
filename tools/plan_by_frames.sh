#!/bin/bash
# frames/s of fit plans vs batch size: tools/plan_by_frames.sh "<frames...>" plan1 plan2 ...
frames=$1; shift
for fr in $frames; do
  for plan in "$@"; do
    PWPP_FIT_PLAN=$plan python bench.py --frames $fr --steps 20 --warmup 3 --no-cpu-baseline --skip-latency --no-profile-events 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frames $fr plan $plan fps', round(d['value']), 'ms', round(d['ms_per_step'],3))"
  done
done
