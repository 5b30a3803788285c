#!/bin/bash
# round 2, two parts per bin: which bins to split, where to split them (single-stream per-kernel times), then the parity suite
mkdir -p gpurun_out
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 3 --no-overlap --no-cpu-baseline --skip-latency > gpurun_out/parts2_$name.json 2> gpurun_out/parts2_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/parts2_%s.json'%n).read().strip().splitlines()[-1])
    k=d['kernel_ms']
    print('%-12s %8.0f f/s  %.3f ms | K1 %.3f  W16 %.3f  W64 %.3f  K5 %.3f  K6 %.3f | ws %.1f GB redo %s'%(n,d['value'],d['ms_per_step'],k.get('k_czm_bin_scatter',0),k.get('k_fit_w64<16,64>',0),k.get('k_fit_w64<64,2>',0),k.get('k_gle_tgr',0),k.get('k_emit',0),d['binning']['workspace_gb'],d['binning']['redone_two_pass']))
except Exception as e:
    print(n,'FAILED',e); print(open('gpurun_out/parts2_%s.err'%n).read()[-800:])
PY
}
run zone0 A=1
run allzones PWPP_HI_SPLIT_ZONES=4
run nosplit PWPP_HI_SPLIT_ZONES=0
run z0_h04 PWPP_HI_SPLIT=0.4
run z0_h08 PWPP_HI_SPLIT=0.8
run z01 PWPP_HI_SPLIT_ZONES=2
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --skip-latency > gpurun_out/parts2_overlap.json 2> gpurun_out/parts2_overlap.err; python -c "
import json;d=json.loads(open('gpurun_out/parts2_overlap.json').read().strip().splitlines()[-1]);print('overlap default',d['value'],d['ms_per_step'])"
timeout 1200 python -m pytest tests -m gpu -q --timeout 100 -o timeout_method=thread 2>&1 | tail -25 > gpurun_out/parts2_tests.txt
cat gpurun_out/parts2_tests.txt
