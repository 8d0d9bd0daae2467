R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6wpred
mkdir -p $OUT
cd $R
(
timeout 900 python -m pytest tests -m gpu -x -q -k "wp_digits or generic or config4 or split or outlier or shape" 2>&1 | tail -5
echo "== fuzz"; timeout 600 python tools/fuzz_generic.py 17 32 2>&1 | tail -2
for r in 1 2; do echo "== cfg4 new"; timeout 300 python bench.py --config 4 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg4', j['ms_per_step'])"; done
for s in "256 20" "256 32" "256 48" "512 10" "300 10"; do set -- $s; echo "== d$1 k$2"; timeout 300 python bench.py --n 2000000 --d $1 --k $2 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o cfg4 --output-format csv -- python3 $R/bench.py --config 4 --no-cpu --steps 2 --warmup 1 > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs python3 -c "import csv,sys; [print(r[0][:70], r[1], r[3]) for r in list(csv.reader(open(sys.argv[1])))[1:8]]"
rm -rf $OUT/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s --output-format csv -- python3 $R/bench.py --n 2000000 --d 512 --k 10 --no-cpu --steps 3 --warmup 1 > $OUT/prof2.log 2>&1
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs python3 -c "import csv,sys; [print(r[0][:70], r[1], r[3]) for r in list(csv.reader(open(sys.argv[1])))[1:8]]"
rm -rf $OUT/prof
) 2>&1 | tee $OUT/wpred.log
