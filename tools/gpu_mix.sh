mkdir -p gpurun_out/r2m
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mixture or config5 or golden_mix or sharded_mixture or different_state" 2>&1 | tail -6
timeout 600 python tools/time_mix.py 5000000 256 10 8 2>&1 | tail -5
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench', j['value'], j['roofline']['kernel_avg_ms'])"
