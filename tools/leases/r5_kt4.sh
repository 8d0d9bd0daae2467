# Round 5: per-kernel statistics of config 4 for development builds.   bash tools/leases/r5_kt4.sh <out-dir> <lib-suffix>...
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5kt4}
mkdir -p $OUT
shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  LIB=$GRAFT_REPO_ROOT/ppca_rs_amd/libppca_hip.so
  [ "$L" != main ] && LIB=$GRAFT_REPO_ROOT/ppca_rs_amd/libppca_hip_$L.so
  export PPCA_HIP_LIB=$LIB
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$L -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu > $OUT/kt_$L.json 2> $OUT/kt_$L.err
  f=$(ls $OUT/kt_$L/*/*kernel_stats.csv | head -1); cp $f $OUT/cfg4_kernel_stats_$L.csv
  echo "== $L"
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print("%-70s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  python3 - $OUT/kt_$L <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "i8gemm" in r["Kernel_Name"]:
        per[r["Grid_Size_X"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g, v in sorted(per.items()):
    print("i8gemm grid", g, "n", len(v), "avg us %.1f" % (sum(v) / len(v)))
PY
done 2>&1 | tee $OUT/summary.log
