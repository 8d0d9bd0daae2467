OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4l}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_steady_state.py tests/test_gpu_parity.py -m gpu -q -x -k "output_passes or fused_path_shape_sweep or mixture_against or golden or edge_cases or guard" > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
for v in 1 0; do echo "PPCA_LLK8=$v"; PPCA_LLK8=$v timeout 300 python tools/time_passes.py 5000000 256 10 2>&1 | grep -v amdgpu.ids | grep -i "llk" ; done | tee $OUT/llk_times.log
timeout 900 python bench.py --config 5 --no-cpu > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python -c "
import json; j=json.load(open('$OUT/bench_cfg5.json')); r=j['roofline']; print('cfg5', round(j['ms_per_step'],2), 'ms/iter; first', [round(x,1) for x in j['regimes']['first_iterations']['ms_per_step']])"
