# the cliff around the fused kernel: shapes one step outside it, after the generic pipeline's small-shape fixes
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3c}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generic or config4 or reference_usage" > $OUT/gpu_tests_generic.log 2>&1; tail -3 $OUT/gpu_tests_generic.log
for s in "200 16" "256 11" "256 16" "300 10" "512 10" "256 10" "200 10"; do
  set -- $s
  timeout 600 python bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu > $OUT/cliff_d$1_k$2.json 2> $OUT/cliff_d$1_k$2.err
done
timeout 600 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/' + os.environ.get('COUT','r3c') + '/*.json')):
    try:
        j = json.load(open(f)); r = j['roofline']
        print(os.path.basename(f), round(j['value'], 2), 'it/s', round(j['ms_per_step'], 3), 'ms frac', round(r['frac'], 3), r['kernel'][:40])
    except Exception as e:
        print(f, 'ERR', e)
PY
