# phase table of the eight-wave kernel (diagnostic build: tools/devbuild.py --timing)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3t}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_dev.so timeout 300 python bench.py --n 2000000 --steps 3 --warmup 1 --no-cpu > $OUT/timing.json 2> $OUT/timing.err
grep "em8 cycles" $OUT/timing.err | tail -2
python -c "
import json; j=json.load(open('$OUT/timing.json')); print(j['value'], j['roofline']['kernel_avg_ms'])"
