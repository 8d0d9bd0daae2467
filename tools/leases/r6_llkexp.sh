R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6llk
mkdir -p $OUT
cd $R
(
timeout 120 python tools/time_passes.py 4000000 256 10 2>&1 | grep -E "llk|smooth|extrapolate" | head -3 | sed "s/^/base /"
export PPCA_LLK8_TIMING=1
PPCA_HIP_LIB=$R/ppca_rs_amd/libppca_hip_exp_TIMING.so timeout 90 python tools/time_passes.py 4000000 256 10 2>&1 | grep -E "llk" | head -2 | sed "s/^/TIMING /"
unset PPCA_LLK8_TIMING
timeout 300 python bench.py --config 5 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg5', j['ms_per_step'], j['llk_per_sample_trace'][-1])"
timeout 600 python -m pytest tests -m gpu -x -q -k "llk or eight_wave or golden or multi_component or output_passes" 2>&1 | tail -2
) | tee $OUT/llk_exp.log
