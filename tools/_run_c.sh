OTAL_WDIRECT_BM=32 python -m pytest tests/test_ops_gpu.py tests/test_half_chain_gpu.py tests/test_bf16_layer_pin_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r05i_wd32_tests.txt
bash tools/ab_lanes.sh "-" "OTAL_WDIRECT_BM=32" "OTAL_WDIRECT_BM=32 OTAL_WDIRECT_BLOCKS=512" "-" "OTAL_WDIRECT_BM=32" > gpurun_out/r05i_wd32.txt 2>&1
