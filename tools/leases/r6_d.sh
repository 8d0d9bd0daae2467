# Round 6: smooth / extrapolate on the eight-wave sweep -- tests, timings A/B against the four-wave pass_kernel
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6d
mkdir -p $OUT
cd $R
python -m ppca_rs_amd.build > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q -k "eight_wave_sweep or output_passes or shape_sweep or test_golden or edge or zero or extrapol" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
python tools/time_passes.py 4000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/passes.log; cat $OUT/passes.log
PPCA_RECON8=0 python tools/time_passes.py 4000000 256 10 2>&1 | grep -v amdgpu.ids > $OUT/passes_recon8_off.log; cat $OUT/passes_recon8_off.log
python tools/time_passes.py 4000000 200 10 2>&1 | grep -v amdgpu.ids > $OUT/passes_d200.log; cat $OUT/passes_d200.log
