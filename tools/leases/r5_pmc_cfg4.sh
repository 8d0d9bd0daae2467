# Round 5: FETCH_SIZE / WRITE_SIZE of the split pipeline's kernels at config 4's shape, XCD-aware tile order on / off.
#   bash tools/leases/r5_pmc_cfg4.sh <out-dir> [lib-suffix]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5pmc4}
mkdir -p $OUT
LIB=$GRAFT_REPO_ROOT/ppca_rs_amd/libppca_hip.so
[ -n "$2" ] && LIB=$GRAFT_REPO_ROOT/ppca_rs_amd/libppca_hip_$2.so
export PPCA_HIP_LIB=$LIB
cd /tmp && export TMPDIR=/tmp
export PMC_N=175000 PMC_STEPS=1
for X in 1 0; do
  export PPCA_I8GEMM_XCD=$X
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_x$X -- python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py > $OUT/fetch_x$X.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_x$X -- python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py > $OUT/write_x$X.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/make_traffic_all.py $OUT/fetch_x$X $OUT/write_x$X 175000 1024 1 $OUT/traffic_cfg4_x$X.json "$(cat $GRAFT_REPO_ROOT/tools/commit_stamp.txt 2>/dev/null)" "config 4 shape, PPCA_I8GEMM_XCD=$X" | tail -22
done
cd /tmp
for X in 1 0; do
  PPCA_I8GEMM_XCD=$X rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_x$X -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu > $OUT/kt_x$X.json 2> $OUT/kt_x$X.err
  f=$(ls $OUT/kt_x$X/*/*kernel_stats.csv | head -1); cp $f $OUT/cfg4_kernel_stats_x$X.csv
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-70s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
