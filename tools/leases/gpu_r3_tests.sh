OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3g}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/gpu_tests.log 2>&1; tail -22 $OUT/gpu_tests.log
bash tools/leases/gpu_r3_cliffprof.sh ${1:-r3g}
