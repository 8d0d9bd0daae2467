# Round 5: the skinny statistics products on v_mfma_f64_4x4x4 (PPCA_SKINNY_444 = 1 / 0): parity of the split pipeline, config 4, neighbours, per-kernel times.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5skinny}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "generic or split or config4 or k16 or cliff" 2>&1 | tail -3 | tee $OUT/parity.log
run() { timeout 300 python bench.py "$@" --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],2), 'ms per EM iteration')"; }
for rep in 1 2; do
  echo -n "4x4x4 config 4: "; run --config 4
  echo -n "PPCA_SKINNY_444=0 config 4: "; PPCA_SKINNY_444=0 run --config 4
done 2>&1 | tee $OUT/ab.log
for s in "256 20" "256 32" "256 48" "512 10" "300 10"; do set -- $s
  echo -n "4x4x4 d=$1 k=$2: "; run --n 2000000 --d $1 --k $2
  echo -n "PPCA_SKINNY_444=0 d=$1 k=$2: "; PPCA_SKINNY_444=0 run --n 2000000 --d $1 --k $2
done 2>&1 | tee -a $OUT/ab.log
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  PPCA_SKINNY_444=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$v -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu > $OUT/kt_$v.json 2> $OUT/kt_$v.err
  f=$(ls -t $OUT/kt_$v/*/*kernel_stats.csv | head -1)
  grep -i "skinny_xt" $f | head -3 | cut -c1-160 | sed "s/^/[PPCA_SKINNY_444=$v] /"
done 2>&1 | tee -a $OUT/ab.log
