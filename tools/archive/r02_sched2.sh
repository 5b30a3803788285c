#!/bin/bash
# round 2: memory stream + several fit streams
mkdir -p gpurun_out
run() {  # name, bench args, env...
  local name=$1; local args=$2; shift; shift
  env "$@" timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --skip-latency $args > gpurun_out/sched_$name.json 2> gpurun_out/sched_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/sched_%s.json'%n).read().strip().splitlines()[-1])
    k=d['kernel_ms']
    print('%-12s %8.0f f/s  %.3f ms | K1 %.3f  W16 %.3f  W64 %.3f  K5 %.3f  K6 %.3f | ws %.1f GB redo %s'%(n,d['value'],d['ms_per_step'],k.get('k_czm_bin_scatter',0),k.get('k_fit_w64<16,64>',0),k.get('k_fit_w64<64,2>',0),k.get('k_gle_tgr',0),k.get('k_emit',0),d['binning']['workspace_gb'],d['binning']['redone_two_pass']))
except Exception as e:
    print(n,'FAILED',e); print(open('gpurun_out/sched_%s.err'%n).read()[-800:])
PY
}
run single --no-overlap A=1
run side2 "" PWPP_OVERLAP_MODE=0 PWPP_OVERLAP_RANGES=2
run p4s2 "" PWPP_OVERLAP_RANGES=4 PWPP_FIT_STREAMS=2
run p4s3 "" PWPP_OVERLAP_RANGES=4 PWPP_FIT_STREAMS=3
run p8s2 "" PWPP_OVERLAP_RANGES=8 PWPP_FIT_STREAMS=2
run p8s3 "" PWPP_OVERLAP_RANGES=8 PWPP_FIT_STREAMS=3
run p8s4 "" PWPP_OVERLAP_RANGES=8 PWPP_FIT_STREAMS=4
run p16s4 "" PWPP_OVERLAP_RANGES=16 PWPP_FIT_STREAMS=4
run p2s2 "" PWPP_OVERLAP_RANGES=2 PWPP_FIT_STREAMS=2
run single_fc --no-overlap PWPP_FIT_CONCURRENT=1
