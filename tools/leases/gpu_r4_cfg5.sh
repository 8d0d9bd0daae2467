OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4k}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --config 5 --no-cpu > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python -c "
import json; j=json.load(open('$OUT/bench_cfg5.json')); r=j['roofline']; print('cfg5', round(j['ms_per_step'],2), 'ms/iter; first', [round(x,1) for x in j['regimes']['first_iterations']['ms_per_step']], 'frac', round(r['frac'],4), 'executed TF', round(r['fp64_achieved_tflops'],1), 'rows frac', round(r['rows_gathered_fraction_of_K_x_N'],4), j['guards_last_pass'])"
timeout 600 python bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; j=json.load(open('$OUT/bench.json')); print('headline', round(j['value'],2), round(j['ms_per_step'],3), j['guards_last_pass'])"
