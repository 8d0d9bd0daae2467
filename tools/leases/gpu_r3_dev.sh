# dev loop of the eight-wave kernel: k = 10 builds (tools/devbuild.py [--name=X] [-D...]; tools/devbuild.py --timing --name=devt)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3d}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_dev.so
PPCA_FUZZ_K=10 timeout 600 python tools/fuzz_gpu.py 1 24 2>&1 | tail -2
for v in dev ${EXTRA_LIBS}; do
  export PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$v.so
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > $OUT/$v.json 2> $OUT/$v.err
  python -c "
import json
j=json.load(open('$OUT/$v.json')); print('$v', 'em', round(j['value'],2), 'it/s', round(j['roofline']['kernel_avg_ms'],3),'ms', 'llk', j['llk_per_sample_last_input_model'])
"
done
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_devt.so timeout 300 python bench.py --n 2000000 --steps 3 --warmup 1 --no-cpu > $OUT/timing.json 2> $OUT/timing.err
grep "em8 cycles" $OUT/timing.err | tail -1
