# Round 5: interleaved A/B of development libraries at the headline size, no parity check (timing experiments included).
#   bash tools/leases/r5_quick.sh <out-dir> <lib-suffix>...
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5q}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
shift
for r in 1 2; do for L in "$@"; do
  PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L', round(j['value'],2), 'it/s', round(j['roofline']['kernel_avg_ms'],3), 'ms per launch')"
done; done 2>&1 | tee $OUT/quick.log
