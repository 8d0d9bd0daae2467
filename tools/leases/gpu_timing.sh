mkdir -p gpurun_out/r2t
PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_dev.so python bench.py --n 2000000 --steps 4 --warmup 1 --no-cpu > gpurun_out/r2t/timing.json 2> gpurun_out/r2t/timing.err
tail -4 gpurun_out/r2t/timing.err
python -c "
import json; j=json.load(open('gpurun_out/r2t/timing.json')); print(j['value'], j['roofline']['kernel_avg_ms'])"
