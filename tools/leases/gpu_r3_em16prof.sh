# per-kernel times of the two-kernel EM pass (rocprofv3 kernel trace) at N = 2 M
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3ep}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in "200 16" "256 11"; do
  set -- $s
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_d$1_k$2 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu > $OUT/bench_d$1_k$2.json 2> $OUT/bench_d$1_k$2.err
  f=$(find $OUT/prof_d$1_k$2 -name "*kernel_stats.csv" | head -1)
  python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us', r['Percentage'])
PY
done
