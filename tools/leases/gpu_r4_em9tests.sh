OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4e9}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PPCA_EM9=1
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $OUT/gpu_tests_em9.log 2>&1; tail -12 $OUT/gpu_tests_em9.log
