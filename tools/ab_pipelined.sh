#!/bin/bash
# A/B of builds of libpwpp_hip.so under bench.py's timed region (two batches in flight), interleaved: tools/ab_pipelined.sh libA.so libB.so ... 
for r in 1 2; do
  for L in "$@"; do
    PWPP_LIB_PATH=$L python bench.py --steps 30 --warmup 3 --no-cpu-baseline --skip-latency --skip-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['ms_per_step'],3), 'sync', round(d['synchronous']['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items() if v>0.02})"
  done
done
