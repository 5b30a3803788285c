#!/bin/bash
# round 2, two parts per bin: parity suite, then the default bench and the single-stream one
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -25 > gpurun_out/parts_tests.txt
cat gpurun_out/parts_tests.txt
timeout 300 python bench.py --steps 40 --warmup 5 > gpurun_out/parts_bench.json 2> gpurun_out/parts_bench.err
tail -3 gpurun_out/parts_bench.err; cat gpurun_out/parts_bench.json
timeout 300 python bench.py --steps 40 --warmup 5 --no-overlap > gpurun_out/parts_bench_single.json 2>> gpurun_out/parts_bench.err
cat gpurun_out/parts_bench_single.json
