# em8 changes: parity of the fused path + steady-state tests, then bench lines (default, weighted)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4b}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_steady_state.py "tests/test_gpu_parity.py::test_fused_path_shape_sweep" "tests/test_gpu_parity.py::test_em_iterations_match_oracle" "tests/test_gpu_parity.py::test_golden" "tests/test_gpu_parity.py::test_mixture_against_oracle" "tests/test_gpu_parity.py::test_edge_cases" -m gpu -q -x > $OUT/tests.log 2>&1; tail -8 $OUT/tests.log
timeout 600 python bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err || timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - ${1:-r4b} <<'PY'
import json,sys,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out",sys.argv[1] if len(sys.argv)>1 else "r4b","bench.json")
try:
    j=json.loads(open(p).read().strip().splitlines()[-1]); print("BENCH", j["value"], j["ms_per_step"], j["roofline"]["kernel_avg_ms"], j["roofline"]["frac"])
except Exception as e: print("bench parse failed", e)
PY
