set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6llk
mkdir -p $OUT
cd $R
for v in base NOSTAGE0 NOSOLVE base; do
  if [ $v = base ]; then unset PPCA_HIP_LIB; else export PPCA_HIP_LIB=$R/ppca_rs_amd/libppca_hip_exp_$v.so; fi
  python tools/time_passes.py 4000000 256 10 2>&1 | grep -E "llk|smooth|extrapolate" | head -3 | sed "s/^/$v /"
done | tee $OUT/llk_exp.log
