python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r05g_tests.txt
bash tools/kernel_time.sh 90 -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off > gpurun_out/r05g_k.txt 2>&1
bash tools/ab_lanes.sh "OTAL_LIB_PATH=ab/c774.so" "-" > gpurun_out/r05g_ab.txt 2>&1
