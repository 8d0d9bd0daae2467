set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6e
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "eight_wave_sweep or output_passes or test_golden or extrapol" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
python tools/time_passes.py 4000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/passes.log; cat $OUT/passes.log
python tools/time_passes.py 4000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/passes_b.log; cat $OUT/passes_b.log
