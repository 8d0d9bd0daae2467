OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4j}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_steady_state.py -m gpu -q -x --durations=5 > $OUT/tests.log 2>&1; tail -12 $OUT/tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_kernel or reference_largest or generic_pipeline" > $OUT/tests2.log 2>&1; tail -5 $OUT/tests2.log
timeout 300 python bench.py --n 2000000 --d 200 --k 16 --steps 10 --warmup 2 --no-cpu > $OUT/bench_d200k16.json 2> $OUT/bench.err
python -c "
import json; j=json.load(open('$OUT/bench_d200k16.json')); print('d200 k16 N=2M', round(j['ms_per_step'],3), 'ms/step')"
