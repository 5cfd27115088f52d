python -m pytest tests/test_loss_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_determinism_gpu.py tests/test_train_gpu.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r05e_tests.txt
bash tools/ab_lanes.sh "OTAL_LIB_PATH=ab/c774.so" "OTAL_LOSS_NOSTAGE=1" "-" "OTAL_LOSS_NOSTAGE=1" "-" > gpurun_out/r05e_ab.txt 2>&1
bash tools/kernel_time.sh 40 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-kernels --no-roofline --no-extras --graph off 2>&1 | grep -E "loss|boundary|bmp|heads|head_convs|proj" > gpurun_out/r05e_k.txt
