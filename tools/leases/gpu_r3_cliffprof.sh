# kernel-level profile of the generic pipeline one step outside the fused kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3c}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in ${SHAPES:-"256 11" "200 16"}; do
  set -- $s
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_d$1_k$2 -- python3 $GRAFT_REPO_ROOT/bench.py --n 2000000 --d $1 --k $2 --steps 3 --warmup 1 --no-cpu > $OUT/kt_d$1_k$2.json 2> $OUT/kt_d$1_k$2.err
  f=$(ls -t $(find $OUT/kt_d$1_k$2 -name "*kernel_stats.csv") | head -1)
  echo "== d=$1 k=$2"; head -18 $f | cut -d, -f1-4 | cut -c1-120
done
