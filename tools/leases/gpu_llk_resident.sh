# the llk sweep with its rows L2-resident (diagnostic build) against the normal kernel: how much of a sweep is HBM
cd $GRAFT_REPO_ROOT
for v in devllk devllkres; do echo "== $v"; PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$v.so python tools/time_passes.py 5000000 256 10 2>&1 | grep "llk (total)"; done
