OUT=$GRAFT_REPO_ROOT/gpurun_out/r3cfg4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu > $OUT/b.json 2> $OUT/b.err
python3 - $(find $OUT/p -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us', r['Percentage'])
PY
