tag=$1
export TMPDIR=/tmp
repo=$(pwd)
( cd /tmp && rm -rf /tmp/prof_$tag && OTAL_WGRAD_STREAM=0 OTAL_BRANCH_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- \
    python $repo/bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off > /tmp/prof_$tag.log 2>&1 )
cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_b8_bf16_kernel_stats.csv
python tools/kstats.py gpurun_out/${tag}_bench_b8_bf16_kernel_stats.csv 18 90 > gpurun_out/${tag}_kernel_summary.txt
