#!/bin/bash
# Kernel timeline + per-kernel statistics of the default training step (eager launches): gpurun_out/<tag>_step_timeline.txt,
# gpurun_out/<tag>_kernel_summary.txt.   usage (through gpurun): tools/trace_round.sh r03a
tag=${1:-rXX}
export TMPDIR=/tmp
repo=$(pwd)
( cd /tmp && rm -rf /tmp/trace_$tag && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$tag -o b -- \
    python $repo/bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off > /tmp/trace_$tag.log 2>&1 )
cp $(find /tmp/trace_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_b8_bf16_kernel_stats.csv
python tools/kstats.py gpurun_out/${tag}_bench_b8_bf16_kernel_stats.csv 18 70 > gpurun_out/${tag}_kernel_summary.txt
python tools/trace_step.py $(find /tmp/trace_$tag -name "*kernel_trace.csv" | head -1) 2 > gpurun_out/${tag}_step_timeline.txt
tail -1 gpurun_out/${tag}_step_timeline.txt
