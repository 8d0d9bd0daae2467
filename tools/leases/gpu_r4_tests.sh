OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4tests}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $OUT/gpu_tests.log 2>&1; tail -30 $OUT/gpu_tests.log
