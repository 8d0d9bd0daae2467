OUT=$GRAFT_REPO_ROOT/gpurun_out/r3pass
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o p -- python3 $GRAFT_REPO_ROOT/tools/time_passes.py 2000000 200 16 > $OUT/b.log 2> $OUT/b.err
python3 - $(find $OUT/p -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us', r['Percentage'])
PY
