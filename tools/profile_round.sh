#!/bin/bash
# The round's committed evidence (run on the GPU box through gpurun): kernel-trace statistics of the default bench command,
# the FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, no trace domains besides --kernel-trace) and the bench line.
# usage: tools/profile_round.sh r02     -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
tag=${1:-rXX}
export TMPDIR=/tmp
repo=$(pwd)
# Per-kernel durations are taken with ONE stream (OTAL_WGRAD_STREAM=0 OTAL_BRANCH_LANE=0): on the step's two / three streams
# concurrent kernels stretch each other, and the sum of their durations exceeds the step (tools/trace_round.sh records that
# timeline).  bench.py's roofline leg times its per-launch HIP events the same way (ops.CONV_PROFILE turns the streams off).
( cd /tmp && rm -rf /tmp/prof_$tag && OTAL_WGRAD_STREAM=0 OTAL_BRANCH_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- \
    python $repo/bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off > /tmp/prof_$tag.log 2>&1 )
cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_b8_bf16_kernel_stats.csv
python tools/kstats.py gpurun_out/${tag}_bench_b8_bf16_kernel_stats.csv 18 60 > gpurun_out/${tag}_kernel_summary.txt
bash tools/pmc_step.sh gpurun_out/${tag}_pmc_step_traffic > /dev/null 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default_b8_bf16.json 2> gpurun_out/${tag}_bench_default.err
