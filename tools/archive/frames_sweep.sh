#!/bin/bash
# usage: tools/frames_sweep.sh "<plan>" frames...   -- per-frame kernel times of a fit plan vs batch size
plan=$1; shift
for fr in "$@"; do
  PWPP_FIT_PLAN=$plan python bench.py --frames $fr --steps 10 --warmup 2 --no-cpu-baseline --skip-latency 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); F=$fr; print('$plan', F, 'fps', round(d['value']), 'us/frame:', {k:round(1000*v/F,3) for k,v in d['kernel_ms'].items() if v>0.02})"
done
