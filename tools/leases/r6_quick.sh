R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6dress
mkdir -p $OUT
cd $R
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
echo "wall ${SECONDS}s"
python3 -c "
import json
j=json.loads(open('$OUT/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline']['value'], {k:(round(v['value'],2), round(v['ms_per_step'],3)) for k,v in j['secondary'].items()})
"
