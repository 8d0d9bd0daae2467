R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6shapes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sh in "512 10" "300 10" "200 16" "256 13"; do
  set -- $sh
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s --output-format csv -- python3 $R/bench.py --n 2000000 --d $1 --k $2 --no-cpu --steps 3 --warmup 1 > $OUT/prof_$1_$2.log 2>&1
  echo "== d $1 k $2"; tail -1 $OUT/prof_$1_$2.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"
  find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs python3 -c "import csv,sys; [print(r[0][:90], r[1], r[3]) for r in list(csv.reader(open(sys.argv[1])))[1:14] if 'synth' not in r[0]]"
  rm -rf $OUT/prof
done 2>&1 | tee $OUT/shapes.log
