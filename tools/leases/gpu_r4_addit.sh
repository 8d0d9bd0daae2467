OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4g}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
shift
for L in "$@"; do PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 300 python tools/check_additivity.py 10000000 2>&1 | tail -12 | tee -a $OUT/addit.log; done
