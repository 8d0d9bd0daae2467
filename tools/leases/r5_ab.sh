# Round 5: parity check of the development build(s), then A/B at the headline size, same box, interleaved.
#   bash tools/leases/r5_ab.sh <out-dir> <lib-suffix>...     (libraries ppca_rs_amd/libppca_hip_<suffix>.so from tools/devbuild.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5ab}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
shift
for L in "$@"; do
  PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 600 python tools/em9_check.py 2>&1 | grep -v amdgpu.ids | tail -3 | sed "s/^/[$L] /"
done 2>&1 | tee $OUT/check.log
for rep in 1 2; do
for L in "$@"; do
  PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/ab_$L.json 2> $OUT/ab_$L.err
  python -c "
import json; j=json.load(open('$OUT/ab_$L.json')); print('$L', round(j['value'],2), 'it/s', round(j['roofline']['kernel_avg_ms'],3), 'ms')"
done
done 2>&1 | tee $OUT/ab.log
