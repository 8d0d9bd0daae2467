# per-wave phase tables of em8 (diagnostic builds: tools/devbuild.py --timing --name=X); args: out-dir lib-suffixes...
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4t}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
shift
for L in "$@"; do
  PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 300 python bench.py --n 2000000 --steps 3 --warmup 1 --no-cpu > $OUT/timing_$L.json 2> $OUT/timing_$L.err
  echo "== $L"; grep "em8 wave\|em8 cycles" $OUT/timing_$L.err | tail -9
  python -c "
import json; j=json.load(open('$OUT/timing_$L.json')); print(j['value'], j['roofline']['kernel_avg_ms'])"
done
