# Round 5: config 4 with the XCD-aware tile order of the int8 GEMM on / off x the pipeline variants.  bash tools/leases/r5_cfg4_xcd.sh <out-dir> <lib-suffix>...
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5cfg4x}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
shift
LIBS="$@"
for L in $LIBS; do
  PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "generic_pipeline_matches_oracle or outlier_row_on_the_split" 2>&1 | tail -1 | sed "s/^/[$L xcd=1] /"
done 2>&1 | tee $OUT/parity.log
for rep in 1 2; do for L in $LIBS; do for X in 1 0; do
  PPCA_I8GEMM_XCD=$X PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 600 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L xcd=$X config 4', round(j['ms_per_step'],2), 'ms per EM iteration')"
  for s in "256 20" "512 10"; do set -- $s
    PPCA_I8GEMM_XCD=$X PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 300 python bench.py --n 2000000 --d $1 --k $2 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L xcd=$X d=$1 k=$2', round(j['ms_per_step'],2), 'ms per EM iteration')"
  done
done; done; done 2>&1 | tee $OUT/ab.log
