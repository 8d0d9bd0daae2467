# the fused pre-solve pass of the split pipeline: parity (generic tests) and timing against the three separate passes
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3pp}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generic or config4 or reference_usage or different_state or two_kernel" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for s in "512 10" "300 10" "256 20"; do
  set -- $s
  for prep in 1 0; do
    PPCA_GENERIC_PREP=$prep timeout 600 python bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu 2>/dev/null > /tmp/o.json
    python -c "
import json; j=json.load(open('/tmp/o.json')); print('d=$1 k=$2 prep=$prep', round(j['ms_per_step'],3), 'ms')"
  done
done
for prep in 1 0; do
PPCA_GENERIC_PREP=$prep timeout 600 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu 2>/dev/null > /tmp/o.json
python -c "
import json; j=json.load(open('/tmp/o.json')); print('config 4 prep=$prep', round(j['ms_per_step'],2), 'ms')"
done
