#!/bin/bash
# round-2 first contact: baseline of the round-1 build on today's box + the plan variants VERDICT asks to retry
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
for plan in "W16:1023,W64.2:65535" "W16:1023,W64.4:65535" "W16:1023,W64.8:65535" "W16:2047,W64.4:65535" "W16.32:1023,W64.4:65535"; do
  echo "== $plan" >> $O/plans.txt
  PWPP_FIT_PLAN=$plan python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-latency 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]), round(d[\"ms_per_step\"],3), {k:round(v,3) for k,v in d[\"kernel_ms\"].items() if \"fit\" in k})" >> $O/plans.txt 2>&1
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --overlap > $O/bench_overlap.json 2> $O/bench_overlap.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload dense > $O/bench_dense.json 2> $O/bench_dense.err
cat $O/plans.txt
for f in default overlap dense; do python -c "import json;d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]);print('$f',round(d['value']),d['ms_per_step'],d.get('kernel_ms'))"; done
