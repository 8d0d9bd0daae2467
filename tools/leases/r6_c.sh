# Round 6: where the time goes one step outside the fused kernels (d = 512 / 300, k = 10 on the split pipeline; d = 200, k = 16 on K3b)
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in "512 10" "300 10" "200 16"; do
  set -- $s
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_d$1_k$2 -- python3 $R/bench.py --n 2000000 --d $1 --k $2 --steps 4 --warmup 1 --no-cpu > $OUT/cliff_d$1_k$2.json 2> $OUT/cliff_d$1_k$2.err
  f=$(ls -t $OUT/kt_d$1_k$2/*/*kernel_stats.csv | head -1); head -14 $f | cut -c1-200
done
cd $R
python tools/time_passes.py 4000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/passes.log; cat $OUT/passes.log
