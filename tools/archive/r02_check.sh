#!/bin/bash
# quick check after a kernel change: single-stream and default bench (per-kernel times), then the GPU parity suite
mkdir -p gpurun_out
run() {  # name, bench args, env...
  local name=$1; local args=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --skip-latency $args > gpurun_out/check_$name.json 2> gpurun_out/check_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/check_%s.json'%n).read().strip().splitlines()[-1])
    k=d['kernel_ms']
    print('%-14s %8.0f f/s  %.3f ms | K1 %.3f  W16 %.3f  W64 %.3f  K5 %.3f  K6 %.3f | K1/W16 %.3f redo %s'%(n,d['value'],d['ms_per_step'],k.get('k_czm_bin_scatter',0),k.get('k_fit_w64<16,64>',0),k.get('k_fit_w64<64,2>',0),k.get('k_gle_tgr',0),k.get('k_emit',0),k.get('k_czm_bin_scatter',0)/k.get('k_fit_w64<16,64>',1),d['binning']['redone_two_pass']))
except Exception as e:
    print(n,'FAILED',e); print(open('gpurun_out/check_%s.err'%n).read()[-800:])
PY
}
run single --no-overlap A=1
run default "" A=1
for extra in "$@"; do run "x_$extra" "" $extra; done
[ -n "$SKIP_TESTS" ] || timeout 600 python -m pytest tests -m gpu -q --timeout 100 -o timeout_method=thread 2>&1 | tail -15 > gpurun_out/check_tests.txt
cat gpurun_out/check_tests.txt
