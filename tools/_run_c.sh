bash tools/ab_lanes.sh "OTAL_PREP_LANE=0" "-" "OTAL_PREP_LANE=0" "-" > gpurun_out/r05f_ab.txt 2>&1
python -m pytest tests/test_train_gpu.py tests/test_determinism_gpu.py tests/test_drivers_gpu.py tests/test_dp_two_ranks_gpu.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r05f_tests.txt
