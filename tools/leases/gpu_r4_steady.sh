# steady-state parity tests + a baseline bench line (round 4, first GPU call)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4a}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_steady_state.py -m gpu -q --durations=20 > $OUT/steady.log 2>&1; tail -40 $OUT/steady.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
