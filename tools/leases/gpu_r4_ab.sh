# A/B of kernel-tuning builds at the headline size, same box, interleaved: args: out-dir lib-suffixes...
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4ab}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
shift
for rep in 1 2; do
for L in "$@"; do
  PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/ab_$L.json 2> $OUT/ab_$L.err
  python -c "
import json; j=json.load(open('$OUT/ab_$L.json')); print('$L', round(j['value'],2), 'it/s', round(j['roofline']['kernel_avg_ms'],3), 'ms')"
done
done
