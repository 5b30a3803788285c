#!/bin/bash
# end-to-end (default schedule) frames/s of builds x plans, interleaved rounds: tools/ab_e2e.sh <rounds> "<lib1> <lib2>" "<plan1>" ...
R=$1; shift; LIBS=$1; shift
for r in $(seq $R); do
for plan in "$@"; do
  for L in $LIBS; do
    PWPP_FIT_PLAN="$plan" PWPP_LIB_PATH=$GRAFT_REPO_ROOT/ab/$L.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --skip-latency --profile-steps 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-28s'%('$L','$plan'), round(d['value']), round(d['ms_per_step'],3))"
  done
done
done
