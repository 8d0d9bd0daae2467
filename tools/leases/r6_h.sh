set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6h
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests/test_gpu_steady_state.py -m gpu -x -q > $OUT/tests_steady.log 2>&1
tail -5 $OUT/tests_steady.log
python bench.py --steps 20 --warmup 3 --no-cpu --no-secondary > $OUT/bench_n10m.json 2> $OUT/bench_n10m.err
python bench.py --steps 20 --warmup 3 --no-cpu --outliers 10 > $OUT/bench_n10m_outliers10.json 2> $OUT/bench_n10m_outliers10.err
python - <<PY
import json
for f in ("bench_n10m", "bench_n10m_outliers10"):
    j = json.load(open("$OUT/%s.json" % f)); print(f, round(j["value"], 3), j["roofline"]["fallback"]["mode_last_pass"], j.get("guards_last_pass"))
PY
timeout 900 python tools/fuzz_gpu2.py > $OUT/fuzz2.log 2>&1; tail -2 $OUT/fuzz2.log
timeout 900 python tools/fuzz_gpu.py > $OUT/fuzz1.log 2>&1; tail -2 $OUT/fuzz1.log
