#!/bin/bash
# per-kernel medians of the single-frame pipeline (fresh frame, then one stateful stream in steady state)
# from rocprofv3 kernel traces; run on the GPU box from the repo root:  bash tools/trace_single.sh
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr1 /tmp/tr2
(cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -o t -- python tools/single_frame_trace.py) > /tmp/tr1.log 2>&1
(cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr2 -o t -- python tools/stream_trace.py) > /tmp/tr2.log 2>&1
cd $R
echo "== fresh single frame"; python tools/trace_gaps.py $(find /tmp/tr1 -name "*kernel_trace.csv" | head -1) 50
echo "== one stateful stream, steady state"; python tools/trace_gaps.py $(find /tmp/tr2 -name "*kernel_trace.csv" | head -1) 150
