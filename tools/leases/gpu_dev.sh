mkdir -p gpurun_out/r2t
export PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_dev.so
PPCA_FUZZ_K=10 timeout 600 python tools/fuzz_gpu.py 1 30 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/r2t/dev10m.json 2> gpurun_out/r2t/dev10m.err
python -c "
import json
j=json.load(open('gpurun_out/r2t/dev10m.json')); print('em', round(j['value'],2), 'it/s', round(j['roofline']['kernel_avg_ms'],3),'ms', 'llk', j['llk_per_sample_last_input_model'])
"
tail -2 gpurun_out/r2t/dev10m.err
timeout 300 python tools/time_passes.py 4000000 256 10 2>&1 | grep -v "^\[" | tail -6
