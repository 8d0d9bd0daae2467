# Round 5: the fixed per-step cost at an eighth of the headline's rows (what an 8-GPU shard sees): per-kernel durations by
# rocprofv3, then the step time against the em9 kernel time.   bash tools/leases/r5_fixed.sh <out-dir>
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5fixed}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/bench.py --n 1250000 --steps 200 --warmup 10 --no-cpu > $OUT/kt_bench.json 2> $OUT/kt.err
f=$(ls $OUT/kt/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/n1250k_kernel_stats.csv && python3 - "$f" <<'PY' | tee $OUT/kernel_stats.log
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %6s avg %9.2f us  total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
cd $GRAFT_REPO_ROOT
for v in "" "PPCA_QPREP_CACHE=0"; do
  env $v python bench.py --n 1250000 --steps 200 --warmup 10 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('N=1.25M $v', round(j['value'],1), 'it/s', round(1e3*j['ms_per_step'],1), 'us per step, kernel', round(1e3*r['kernel_avg_ms'],1), 'us, step - kernel', round(1e3*(j['ms_per_step']-r['kernel_avg_ms']),1), 'us')"
done | tee $OUT/fixed_cost.log
