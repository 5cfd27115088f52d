python -m pytest tests/test_ops_gpu.py tests/test_loss_gpu.py tests/test_model_gpu.py tests/test_determinism_gpu.py tests/test_train_gpu.py tests/test_conv1x1_stream_gpu.py tests/test_half_storage_gpu.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r05d_tests.txt
export TMPDIR=/tmp; repo=$(pwd)
( cd /tmp && rm -rf /tmp/prof_c && OTAL_WGRAD_STREAM=0 OTAL_BRANCH_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o b -- python $repo/bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off > /tmp/prof_c.log 2>&1 )
cp $(find /tmp/prof_c -name "*kernel_stats.csv" | head -1) gpurun_out/r05d_kernel_stats.csv
python tools/kstats.py gpurun_out/r05d_kernel_stats.csv 18 90 > gpurun_out/r05d_kernel_summary.txt
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras > gpurun_out/r05d_bench.json 2> gpurun_out/r05d_bench.err
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras > gpurun_out/r05d_bench2.json 2>> gpurun_out/r05d_bench.err
