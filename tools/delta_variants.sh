#!/bin/bash
# round 5: the delta-round variants of the fit kernels side by side on one box (ab/d*.so built by tools/ab_build.sh)
for L in d0 d1 d2 d3 d3a1 d3a2 d3a4 d0; do
  PWPP_BENCH_NO_SELFCHECK=1 PWPP_LIB_PATH=ab/$L.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-latency --skip-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items() if v>0.02})"
done
