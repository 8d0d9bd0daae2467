# the whole GPU suite (as the driver runs it), then the smoke entry
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6tests
mkdir -p $OUT
cd $R
timeout 3000 python -m pytest tests/ -x -q -m gpu > $OUT/tests_gpu.log 2>&1
tail -8 $OUT/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
