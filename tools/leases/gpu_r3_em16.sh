# two-kernel EM pass for 11 <= k <= 16: parity, chunking, timing against the split pipeline
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3e}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/em16_check.py > $OUT/check.log 2>&1; echo "check rc=$?"; tail -15 $OUT/check.log
PPCA_GEN_CHUNK=256 timeout 600 python tools/em16_check.py > $OUT/check_chunks.log 2>&1; echo "chunked check rc=$?"; tail -3 $OUT/check_chunks.log
for s in "200 16" "256 16" "256 11" "256 13"; do
  set -- $s
  timeout 600 python bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu > $OUT/e16_d$1_k$2.json 2> $OUT/e16_d$1_k$2.err
  python - $OUT/e16_d$1_k$2.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); r = j['roofline']
    print(sys.argv[1].split('/')[-1], round(j['value'], 2), 'it/s', round(j['ms_per_step'], 3), 'ms frac', round(r['frac'], 3))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
PPCA_EM16=0 timeout 600 python bench.py --n 2000000 --d 200 --k 16 --steps 4 --warmup 1 --no-cpu > $OUT/old_d200_k16.json 2> $OUT/old.err; python -c "
import json; j=json.load(open('$OUT/old_d200_k16.json')); print('split pipeline d200 k16', j['ms_per_step'])"
