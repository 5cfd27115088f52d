python -m pytest tests/test_half_chain_gpu.py tests/test_ops_gpu.py -q -m gpu -k "pool" 2>&1 | tail -3 > gpurun_out/r05h_pooltests.txt
python tools/micro_pool_half.py 2>&1 | grep -v amdgpu > gpurun_out/r05h_pool_new.txt
OTAL_POOL_NOW12=1 python tools/micro_pool_half.py 2>&1 | grep -v amdgpu > gpurun_out/r05h_pool_old.txt
