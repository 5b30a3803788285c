#!/bin/bash
# A/B of two environments of the default bench, interleaved on the same box (later runs of a call are slower: the box warms up)
# usage: tools/ab_env.sh "ENV_A=…" "ENV_B=…" [rounds]
A=$1; B=$2; R=${3:-3}
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for v in "$A" "$B"; do
    env $v timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --skip-latency > gpurun_out/ab.json 2> gpurun_out/ab.err
    python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); k=d['kernel_ms']
print('%-40s %8.0f f/s %.3f ms | %s'%(sys.argv[1],d['value'],d['ms_per_step'],' '.join('%s %.3f'%(n.replace('k_','').replace('czm_','').replace('fit_',''),v) for n,v in k.items() if v>0.02)))
PY
  done
done
