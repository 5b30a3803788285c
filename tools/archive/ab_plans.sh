#!/bin/bash
# fit-kernel times of several builds x plans on one box (single-stream schedule): tools/ab_plans.sh "<lib1> <lib2> ..." "<plan1>" "<plan2>" ...
LIBS=$1; shift
for plan in "$@"; do
  for L in $LIBS; do
    PWPP_FIT_PLAN="$plan" PWPP_LIB_PATH=$GRAFT_REPO_ROOT/ab/$L.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-latency --no-overlap 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-28s'%('$L','$plan'), round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items() if v>0.02 and k.startswith('k_fit')})"
  done
done
