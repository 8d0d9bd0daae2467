#!/bin/bash
# parity tests served by the batched blocked solver, verbose and with a per-test timeout (diagnostic)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s4t; mkdir -p $O
cd $R
PYTHONUNBUFFERED=1 timeout 1200 python -u -m pytest tests/test_gpu_parity.py -v -m gpu --timeout 240 -k "generic_pipeline or config4 or different_state_sizes or config5_shape" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for k in 20 32 48 64; do tools/s4bench/s4bench $k 1000000 | tail -2; tools/s4bench/s4bench_t $k 1000000 | tail -10; done > $O/s4bench.txt 2>&1
