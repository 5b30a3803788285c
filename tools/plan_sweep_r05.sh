#!/bin/bash
# fit plans and options under the headline's schedule (two batches in flight): tools/plan_sweep_r05.sh
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 4 --no-cpu-baseline --skip-extras --skip-latency 2>/dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-44s %8.0f f/s  %.3f ms/step  sync %.3f  fits %s' % ('$name', d['value'], d['ms_per_step'], d['synchronous']['ms_per_step'], ' '.join('%s=%.3f'%(n,v) for n,v in k.items() if 'fit' in n and v>0.02)))"
}
run "default" PWPP_X=0
run "W16:1023,W64.2:65535" PWPP_FIT_PLAN="W16:1023,W64.2:65535"
run "W16:1023,W64.8:65535" PWPP_FIT_PLAN="W16:1023,W64.8:65535"
run "W16:511,W64.4:65535" PWPP_FIT_PLAN="W16:511,W64.4:65535"
run "W16:2047,W64.4:65535" PWPP_FIT_PLAN="W16:2047,W64.4:65535"
run "W32:1023,W64.4:65535" PWPP_FIT_PLAN="W32:1023,W64.4:65535"
run "fit_concurrent" PWPP_FIT_CONCURRENT=1
run "bin_block 512" PWPP_BIN_BLOCK=512
run "default again" PWPP_X=0
