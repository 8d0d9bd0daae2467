# Round 5: ablations of the int8 GEMM's K loop (timing experiments, results wrong): per-kernel time of i8gemm at config 4's shape.
#   bash tools/leases/r5_i8abl.sh <out-dir> <lib-suffix>...
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5i8abl}
mkdir -p $OUT
shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  PPCA_HIP_LIB=$GRAFT_REPO_ROOT/ppca_rs_amd/libppca_hip_$L.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$L -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --n 700000 --steps 2 --warmup 1 --no-cpu > $OUT/kt_$L.json 2> $OUT/kt_$L.err
  f=$(ls $OUT/kt_$L/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$L" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "i8gemm_kernel" in r["Name"]:
        print("%-12s %-50s calls %4s avg %8.1f us" % (sys.argv[2], r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done 2>&1 | tee $OUT/ablations.log
