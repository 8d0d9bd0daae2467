timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mixture or config5 or golden or edge or shape_sweep or stats_raw or trainer" 2>&1 | tail -3
timeout 600 python tools/time_mix.py 5000000 256 10 8 14 2>&1 | tail -4
