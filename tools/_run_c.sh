python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r05g_fulltests.txt
