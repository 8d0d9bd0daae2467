OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4c}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_steady_state.py -m gpu -q -x -k "eight_wave or rescales or full_size_statistics or config3" > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
bash tools/leases/gpu_r4_ab.sh $1 p0 p2 p3
bash tools/leases/gpu_r4_timing.sh $1 t2
