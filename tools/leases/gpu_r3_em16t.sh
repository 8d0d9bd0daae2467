# phase table of estep16_kernel (diagnostic build: tools/devbuild16.py --timing) + plain timing of the dev build
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3et}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
K=${K:-16}; D=${D:-200}
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_dev16.so timeout 300 python bench.py --n 2000000 --d $D --k $K --steps 3 --warmup 1 --no-cpu > $OUT/timing.json 2> $OUT/timing.err
grep "em16 estep" $OUT/timing.err | tail -1
python -c "
import json; j=json.load(open('$OUT/timing.json')); print(j['value'], 'it/s', j['ms_per_step'], 'ms')"
