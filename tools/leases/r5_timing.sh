# Round 5: per-wave phase tables of timing builds (tools/devbuild.py --timing --name=<suffix>), N = 2 M.
#   bash tools/leases/r5_timing.sh <out-dir> <lib-suffix>...
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5t}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
shift
for L in "$@"; do
  PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_$L.so timeout 300 python bench.py --n 2000000 --steps 3 --warmup 1 --no-cpu > $OUT/timing_$L.json 2> $OUT/timing_$L.err
  grep "cycles/tile" $OUT/timing_$L.err | grep "wave [0-3]" | tail -4 | sed "s/^/[$L] /"
done 2>&1 | tee $OUT/phase_tables.log
