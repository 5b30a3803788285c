#!/bin/bash
# the overlap schedule's knobs on the bench workload, after round 4's kernels (frames/s, default schedule end to end)
run() { echo -n "$1: "; env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --skip-extras --skip-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.0f f/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
run "X=0"
run "PWPP_OVERLAP_RANGES=3"
run "PWPP_OVERLAP_RANGES=4"
run "PWPP_OVERLAP_MODE=0"
run "PWPP_FIT_STREAMS=1"
run "PWPP_FIT_STREAMS=3"
run "PWPP_HI_SPLIT=0.4"
run "PWPP_HI_SPLIT=0.8"
run "PWPP_HI_SPLIT_ZONES=2"
run "PWPP_BIN_BLOCK=512"
run "X=0"
