#!/bin/bash
# kernel timeline of the lane-graph step: gpurun_out/<tag>_step_timeline.txt
tag=${1:-rXXl}
export TMPDIR=/tmp
repo=$(pwd)
( cd /tmp && rm -rf /tmp/trace_$tag && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$tag -o b -- \
    python $repo/bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph ${2:-lanes} > /tmp/trace_$tag.log 2>&1 )
python tools/trace_step.py $(find /tmp/trace_$tag -name "*kernel_trace.csv" | head -1) 2 > gpurun_out/${tag}_step_timeline.txt
tail -1 gpurun_out/${tag}_step_timeline.txt
