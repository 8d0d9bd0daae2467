# Round 6: state sizes 65..128 on the blocked MFMA solver (NB = 5..8) and the int8-sliced contractions
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6f
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "generic_pipeline_matches_oracle" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
for k in 65 80 100 128; do
  timeout 600 python bench.py --n 2000000 --d 256 --k $k --steps 2 --warmup 1 --no-cpu > $OUT/cliff_d256_k$k.json 2> $OUT/cliff_d256_k$k.err
  python - <<PY
import json
try:
    j = json.load(open("$OUT/cliff_d256_k$k.json")); print("k=$k", j["ms_per_step"], j["roofline"]["frac"])
except Exception as e:
    print("k=$k failed", e)
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_k65 -- python3 $R/bench.py --n 2000000 --d 256 --k 65 --steps 2 --warmup 1 --no-cpu > $OUT/kt_k65.json 2> $OUT/kt_k65.err
f=$(ls -t $OUT/kt_k65/*/*kernel_stats.csv | head -1); head -14 $f | cut -c1-180
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_k128 -- python3 $R/bench.py --n 2000000 --d 256 --k 128 --steps 2 --warmup 1 --no-cpu > $OUT/kt_k128.json 2> $OUT/kt_k128.err
f=$(ls -t $OUT/kt_k128/*/*kernel_stats.csv | head -1); head -14 $f | cut -c1-180
