mkdir -p gpurun_out/r2g
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2g/prof -- python3 $GRAFT_REPO_ROOT/tools/time_config.py 400000 1024 64 block > $GRAFT_REPO_ROOT/gpurun_out/r2g/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r2g/prof -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -24
