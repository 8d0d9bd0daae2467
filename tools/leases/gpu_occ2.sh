cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generic_pipeline or config4" 2>&1 | tail -2
for o in 1 0; do
PPCA_SOLVE_OCC2=$o timeout 600 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu 2>/dev/null > /tmp/o.json
python -c "
import json; j=json.load(open('/tmp/o.json')); print('config 4 occ2=$o', round(j['ms_per_step'],2), 'ms')"
done
for s in "256 20" "256 40"; do set -- $s
for o in 1 0; do
PPCA_SOLVE_OCC2=$o timeout 600 python bench.py --n 1000000 --d $1 --k $2 --steps 3 --warmup 1 --no-cpu 2>/dev/null > /tmp/o.json
python -c "
import json; j=json.load(open('/tmp/o.json')); print('d=$1 k=$2 occ2=$o', round(j['ms_per_step'],2), 'ms')"
done; done
