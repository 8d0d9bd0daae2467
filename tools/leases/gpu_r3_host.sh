# Round-3 host-side items on the GPU: test suite, transfer rates, config 5 bench line, fp64-Gram engine line, 2-rank self-launch error path
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3h}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -4 $OUT/gpu_tests.log
timeout 300 python tools/time_transfer.py > $OUT/transfer.log 2>&1; cat $OUT/transfer.log
timeout 900 python bench.py --config 5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; tail -3 $OUT/bench_cfg5.err
timeout 600 python bench.py --gram fp64 --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m_fp64gram.json 2> $OUT/bench_fp64gram.err
timeout 300 python bench.py --gpus 2 --n 200000 --steps 2 --warmup 1 --no-cpu > $OUT/two_ranks.json 2> $OUT/two_ranks.err; echo "two-rank self-launch rc=$?"; tail -5 $OUT/two_ranks.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/' + os.environ.get('HOUT','r3h') + '/*.json')):
    try:
        j = json.load(open(f)); r = j['roofline']
        print(os.path.basename(f), round(j['value'], 2), j['unit'], round(j['ms_per_step'], 3), 'ms frac', round(r['frac'], 3), j.get('gram_engine','')[:12], j.get('regimes',{}).get('first_iterations',{}).get('ms_per_step'))
    except Exception as e:
        print(f, 'ERR', e)
PY
