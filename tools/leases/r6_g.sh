# Round 6: rows above the scale go round the fixed-point form (em9 back role) -- tests, fuzz, outlier bench lines, headline, config 5
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6g
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests/test_gpu_steady_state.py -m gpu -x -q > $OUT/tests_steady.log 2>&1
tail -5 $OUT/tests_steady.log
python bench.py --steps 20 --warmup 3 --no-cpu --no-secondary > $OUT/bench_n10m.json 2> $OUT/bench_n10m.err
python bench.py --steps 20 --warmup 3 --no-cpu --outliers 1 > $OUT/bench_n10m_outliers1.json 2> $OUT/bench_n10m_outliers1.err
python bench.py --steps 20 --warmup 3 --no-cpu --outliers 10 > $OUT/bench_n10m_outliers10.json 2> $OUT/bench_n10m_outliers10.err
python bench.py --steps 20 --warmup 3 --no-cpu --outliers 1000 > $OUT/bench_n10m_outliers1000.json 2> $OUT/bench_n10m_outliers1000.err
python bench.py --config 5 --no-cpu > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python - <<PY
import json
for f in ("bench_n10m", "bench_n10m_outliers1", "bench_n10m_outliers10", "bench_n10m_outliers1000", "bench_cfg5"):
    try:
        j = json.load(open("$OUT/%s.json" % f))
        print(f, round(j["value"], 3), round(j["ms_per_step"], 3), j["roofline"].get("fallback"), j.get("guards_last_pass"), j.get("regimes", {}).get("first_iterations", {}).get("ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
timeout 900 python tools/fuzz_gpu2.py > $OUT/fuzz2.log 2>&1; tail -4 $OUT/fuzz2.log
