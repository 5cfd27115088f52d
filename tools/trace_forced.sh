#!/bin/bash
# kernel trace of the data-parallel code path on ONE forced RCCL rank (eager launches): gpurun_out/<tag>_kernel_summary.txt
tag=${1:-rXXf}
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 MASTER_PORT=29556 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 OTAL_FORCE_DIST=1
repo=$(pwd)
( cd /tmp && rm -rf /tmp/trace_$tag && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$tag -o b -- \
    python $repo/bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off > /tmp/trace_$tag.log 2>&1 )
cp $(find /tmp/trace_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
python tools/kstats.py gpurun_out/${tag}_kernel_stats.csv 18 70 > gpurun_out/${tag}_kernel_summary.txt
python tools/trace_step.py $(find /tmp/trace_$tag -name "*kernel_trace.csv" | head -1) 2 > gpurun_out/${tag}_step_timeline.txt
tail -1 gpurun_out/${tag}_step_timeline.txt
