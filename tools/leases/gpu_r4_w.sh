OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4w}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
shift
for L in "$@"; do echo "== $L"; PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 300 python tools/time_weighted.py 10000000 256 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/weighted.log; done
