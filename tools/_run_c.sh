python tools/probe_grad_copies.py 2>&1 | grep -v amdgpu > gpurun_out/r05i_gradcopies.txt
python -m pytest tests/test_model_gpu.py tests/test_determinism_gpu.py tests/test_train_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r05i_tests.txt
bash tools/ab_lanes.sh "OTAL_NO_GRAD_SLOTS_HEADS=1" "-" "OTAL_NO_GRAD_SLOTS_HEADS=1" "-" > gpurun_out/r05i_ab.txt 2>&1
