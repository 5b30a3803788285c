#!/bin/bash
# A/B of two builds of libpwpp_hip.so on the same box, interleaved: tools/ab_bench.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do
  for L in $A $B; do
    PWPP_LIB_PATH=$L python bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-latency 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items() if v>0.02})"
  done
done
