# Round 5: the int8 GEMM's barrier-free ring K loop (I8_RING) against the two-buffer loop: parity of the split pipeline, then config 4.
#   bash tools/leases/r5_ring.sh <out-dir> <lib-suffix>...      ("main" = the tree's libppca_hip.so)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5ring}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
shift
lib() { if [ "$1" = main ]; then echo $PWD/ppca_rs_amd/libppca_hip.so; else echo $PWD/ppca_rs_amd/libppca_hip_$1.so; fi; }
for L in "$@"; do
  PPCA_HIP_LIB=$(lib $L) timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "generic_pipeline_matches_oracle or outlier_row_on_the_split or full_size_properties_config4" 2>&1 | tail -2 | sed "s/^/[$L] /"
done 2>&1 | tee $OUT/parity.log
LIBS="$@"
for rep in 1 2; do for L in $LIBS; do
  PPCA_HIP_LIB=$(lib $L) timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L config 4', round(j['ms_per_step'],2), 'ms per EM iteration')"
done; done 2>&1 | tee $OUT/ab.log
for s in "256 20" "512 10"; do set -- $s; for L in $LIBS; do
    PPCA_HIP_LIB=$(lib $L) timeout 300 python bench.py --n 2000000 --d $1 --k $2 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L d=$1 k=$2', round(j['ms_per_step'],2), 'ms per EM iteration')"
done; done 2>&1 | tee -a $OUT/ab.log
