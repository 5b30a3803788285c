#!/bin/bash
# batches in flight x fit classes side by side, under contract v4 (longer fit kernels than rounds 1-5)
for r in 1 2; do
for cfg in "2 0" "3 0" "2 1" "3 1" "4 0"; do
  set -- $cfg
  if [ "$2" = "1" ]; then export PWPP_FIT_CONCURRENT=1; else unset PWPP_FIT_CONCURRENT; fi
  python bench.py --steps 90 --warmup 6 --in-flight $1 --no-cpu-baseline --skip-latency --skip-extras --no-profile-events 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('in flight $1  fit_concurrent $2   %7.0f f/s %6.3f ms'%(d['value'],d['ms_per_step']))"
done; done
