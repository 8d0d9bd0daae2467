OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4h}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_steady_state.py -m gpu -q -x --durations=8 -k "outlier or eight_wave or rescales or config3 or output_passes" > $OUT/tests.log 2>&1; tail -16 $OUT/tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mixture or golden or edge or guard or config5 or full_size_properties" > $OUT/tests2.log 2>&1; tail -5 $OUT/tests2.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; j=json.load(open('$OUT/bench.json')); print('bench', round(j['value'],2), 'it/s', round(j['ms_per_step'],3), 'ms/step; kernel', round(j['roofline']['kernel_avg_ms'],3), 'ms')"
