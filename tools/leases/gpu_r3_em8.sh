# Round-3: eight-wave role-split EM kernel -- parity (fuzz, golden statistics), headline, A/B against the four-wave kernel.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3b}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/dbg_stats.py > $OUT/dbg_stats.log 2>&1; tail -12 $OUT/dbg_stats.log
timeout 600 python tools/fuzz_gpu.py 1 40 > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m.json 2> $OUT/bench_n10m.err; tail -2 $OUT/bench_n10m.err
PPCA_EM8=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_n10m_em4.json 2> $OUT/bench_n10m_em4.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/' + os.environ.get('E8OUT','r3b') + '/*.json')):
    try:
        j = json.load(open(f)); r = j['roofline']
        print(os.path.basename(f), round(j['value'], 2), 'it/s', round(j['ms_per_step'], 3), 'ms frac', round(r['frac'], 3), 'llk', j['llk_per_sample_last_input_model'])
    except Exception as e:
        print(f, 'ERR', e)
PY
if [ "$2" = "tests" ]; then timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -5 $OUT/gpu_tests.log; fi
