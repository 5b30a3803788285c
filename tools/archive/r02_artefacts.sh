#!/bin/bash
# the measured artefacts of round 2 (run on the GPU box): bench lines of the default, single-stream, dense and
# streams configurations -> gpurun_out/r02/ (copied to profiles/ by hand)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-overlap --no-cpu-baseline > $O/bench_single_stream.json 2> $O/bench_single_stream.err
python bench.py --workload dense --no-cpu-baseline --steps 20 > $O/bench_dense_1024.json 2> $O/bench_dense_1024.err
python bench.py --workload dense --frames 128 --no-cpu-baseline --steps 40 > $O/bench_dense_128.json 2> $O/bench_dense_128.err
python tools/streams_rate.py > $O/streams.txt 2>&1
python tools/small_batches.py > $O/small_batches.txt 2>&1
for f in default single_stream dense_1024 dense_128; do python -c "
import json;d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
print('$f',round(d['value']),round(d['ms_per_step'],3),d.get('binning'),d['latency'],{k:round(v,3) for k,v in d['kernel_ms'].items() if v>0.01}, d.get('cpu_baseline',{}).get('value'))"; done
tail -5 $O/streams.txt; tail -8 $O/small_batches.txt
python tools/order_cost.py 256 > $O/order_cost.txt 2>&1; cat $O/order_cost.txt
