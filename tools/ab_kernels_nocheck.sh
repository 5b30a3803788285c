#!/bin/bash
# tools/ab_kernels.sh for builds whose results are deliberately wrong (ablations): no self-check, per-kernel times only
R=$1; shift; LIBS=$1; shift
for r in $(seq $R); do
  for L in $LIBS; do
    PWPP_BENCH_NO_SELFCHECK=1 PWPP_LIB_PATH=$GRAFT_REPO_ROOT/ab/$L.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --skip-latency --skip-extras --profile-steps 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-8s %7.0f f/s %6.3f ms  '%('$L',d['value'],d['ms_per_step'])+' '.join('%s=%.3f'%(n.replace('k_',''),v) for n,v in k.items() if v>0.01))"
  done
done
