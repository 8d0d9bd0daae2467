set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6i
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -k "llk or output or eight_wave or golden or mixture or multi or config5 or shape_sweep or edge or fuzz or guard" > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
python tools/time_passes.py 4000000 256 10 2>&1 | grep -v amdgpu.ids | head -4 | tee $OUT/passes.log
python bench.py --config 5 --no-cpu > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python - <<PY
import json
j = json.load(open("$OUT/bench_cfg5.json")); print("cfg5", j["ms_per_step"], j["regimes"]["first_iterations"]["ms_per_step"], j["llk_per_sample_trace"][-1])
PY
