# Round 5: config 4 A/B of development builds of the int8 GEMM's ring loop (no parity here: tools/leases/r5_ring.sh has it).
#   bash tools/leases/r5_ring_ab.sh <out-dir> <lib-suffix>...      ("main" = the tree's libppca_hip.so)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5ringab}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
shift
lib() { if [ "$1" = main ]; then echo $PWD/ppca_rs_amd/libppca_hip.so; else echo $PWD/ppca_rs_amd/libppca_hip_$1.so; fi; }
LIBS="$@"
for rep in 1 2; do for L in $LIBS; do
  PPCA_HIP_LIB=$(lib $L) timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L config 4', round(j['ms_per_step'],2), 'ms per EM iteration')"
done; done 2>&1 | tee $OUT/ab.log
