mkdir -p gpurun_out/r2g
timeout 900 python bench.py --config 4 --steps 5 --warmup 1 > gpurun_out/r2g/bench_cfg4.json 2> gpurun_out/r2g/bench_cfg4.err
python -c "
import json; j=json.load(open('gpurun_out/r2g/bench_cfg4.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_avg_ms'], j['cpu_baseline']['value'] if j['cpu_baseline'] else None)"
tail -2 gpurun_out/r2g/bench_cfg4.err
PPCA_GENERIC_FP64=1 PPCA_GENERIC_REG_SOLVE=1 timeout 900 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('round-1 form:', j['value'], j['ms_per_step'])"
