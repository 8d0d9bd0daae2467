# Round 6, first lease: the multi-component mixture step -- targeted tests, config 5 A/B (multi vs component by component) on one box,
# kernel statistics of the new step, and the default bench line with its secondary legs.
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6a
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "mix or multi or config5 or golden" > $OUT/tests_mix.log 2>&1
tail -5 $OUT/tests_mix.log
python bench.py --config 5 --no-cpu > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
PPCA_MIX_MULTI=0 python bench.py --config 5 --no-cpu > $OUT/bench_cfg5_single.json 2> $OUT/bench_cfg5_single.err
python bench.py --config 5 --no-cpu > $OUT/bench_cfg5_b.json 2> $OUT/bench_cfg5_b.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cfg5 -- python3 $R/bench.py --config 5 --no-cpu > $OUT/kt_cfg5_bench.json 2> $OUT/kt_cfg5.err
cd $R
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for f in bench_cfg5 bench_cfg5_single bench_cfg5_b; do python - <<PY
import json
try:
    j = json.load(open("$OUT/$f.json"))
    print("$f", j["ms_per_step"], j["regimes"]["first_iterations"]["ms_per_step"], j["roofline"]["rows_gathered_last_step"])
except Exception as e:
    print("$f failed", e)
PY
done
tail -3 $OUT/bench_cfg5.err
python - <<PY
import json
j = json.load(open("$OUT/bench_default.json"))
print("headline", j["value"], j["ms_per_step"], j["roofline"]["frac"])
for k, v in j.get("secondary", {}).items():
    print(k, {a: v.get(a) for a in ("value", "ms_per_step", "error", "leg_wall_s")}, (v.get("roofline") or {}).get("frac"))
PY
f=$(ls -t $OUT/kt_cfg5/*/*kernel_stats.csv | head -1); head -30 $f
