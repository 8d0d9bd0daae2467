mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "generic or usage_patterns or different_state" 2>&1 | tail -2
timeout 600 python tools/time_config.py 400000 1024 64 block 2>&1 | tail -2
PPCA_GENERIC_SOLVE=bc timeout 600 python tools/time_config.py 400000 1024 64 block 2>&1 | tail -1
