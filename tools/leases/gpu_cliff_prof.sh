#!/bin/bash
# kernel breakdown of the split pipeline at the shapes just outside the fused kernels (diagnostic)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4cliff; mkdir -p $O
cd $R
for s in "256 20" "256 32"; do
  set -- $s
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_d$1_k$2 -- python3 bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu > $O/bench_d$1_k$2.json 2> $O/err_d$1_k$2.log
  f=$(ls $O/p_d$1_k$2/*/*kernel_stats.csv | head -1); cp $f $O/stats_d$1_k$2.csv
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_cfg4 -- python3 bench.py --config 4 --steps 4 --warmup 1 --no-cpu > $O/bench_cfg4.json 2> $O/err_cfg4.log
f=$(ls $O/p_cfg4/*/*kernel_stats.csv | head -1); cp $f $O/stats_cfg4.csv
