# Round 5: the GPU suite, then the headline / outlier / small-shard bench lines.   bash tools/leases/r5_tests.sh <out-dir> [pytest -k expr]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5tests}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -x --durations=10 -k "$2" > $OUT/gpu_tests.log 2>&1
else
  timeout 3000 python -m pytest tests -m gpu -q --durations=15 > $OUT/gpu_tests.log 2>&1
fi
tail -25 $OUT/gpu_tests.log
python bench.py --steps 20 --warmup 3 --no-cpu > $OUT/bench_n10m.json 2> $OUT/bench_n10m.err
python bench.py --steps 20 --warmup 3 --no-cpu --outliers 1 > $OUT/bench_n10m_outliers1.json 2> $OUT/bench_n10m_outliers1.err
python bench.py --steps 20 --warmup 3 --no-cpu --outliers 10 > $OUT/bench_n10m_outliers10.json 2> $OUT/bench_n10m_outliers10.err
python bench.py --n 1250000 --steps 200 --warmup 10 --no-cpu > $OUT/bench_n1250k.json 2> $OUT/bench_n1250k.err
PPCA_QPREP_CACHE=0 python bench.py --n 1250000 --steps 200 --warmup 10 --no-cpu > $OUT/bench_n1250k_nocache.json 2> $OUT/bench_n1250k_nocache.err
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/" + (os.sys.argv[1] if len(os.sys.argv) > 1 else "")
PY
for f in bench_n10m bench_n10m_outliers1 bench_n10m_outliers10 bench_n1250k bench_n1250k_nocache; do
  python -c "
import json; j=json.load(open('$OUT/$f.json')); r=j['roofline']
print('$f', round(j['value'],2), 'it/s', round(j['ms_per_step'],4), 'ms/step, kernel', round(r['kernel_avg_ms'],4), 'ms, step - kernel', round(1e3*(j['ms_per_step']-r['kernel_avg_ms']),1), 'us, fallback', r.get('fallback'), j.get('guards_last_pass'))" 2>&1 | tail -2
done | tee $OUT/summary.log
