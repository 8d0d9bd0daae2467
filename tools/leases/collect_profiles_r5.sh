# gpurun_out/r5f (tools/leases/r5_final.sh) -> profiles/r05, then the numbers file and README table generated FROM those files
# (tools/make_profiles_r05.py): nothing in profiles/r05/numbers.md is typed by hand.
set -e
cd "$(dirname "$0")/../.."
S=gpurun_out/r5f; D=profiles/r05
mkdir -p $D/cliff
for f in mfma_peak.txt mfma_peak.json traffic.json traffic_cfg4.json traffic_cfg5.json bench_n10m.json bench_n10m_outliers1.json bench_n10m_outliers10.json \
         bench_n10m_fp64gram.json bench_n10m_em8.json bench_n1250k.json bench_n1250k_nocache.json weighted_n10m.log bench_cfg5.json bench_cfg4.json \
         bench_cfg4_plain_order.json passes.log passes_d200_k16.log additivity_n10m.log outlier_probe.log soak_em.log; do
  [ -f $S/$f ] && cp $S/$f $D/$f
done
cp $S/cliff_d*_k*.json $D/cliff/ 2>/dev/null || true
pick() { ls -t $S/$1/*/*$2 2>/dev/null | head -1; }  # (the newest: gpurun_out/ keeps the files of earlier leases)
for p in "kt bench_n10m" "kt_n1250k n1250k" "kt_out1 bench_n10m_outliers1" "kt_cfg5 bench_cfg5" "kt_cfg4 bench_cfg4"; do set -- $p; f=$(pick $1 kernel_stats.csv); [ -n "$f" ] && cp $f $D/$2_kernel_stats.csv; done
for p in "pmc_fetch pmc_n10m_FETCH_SIZE" "pmc_write pmc_n10m_WRITE_SIZE" "pmc_mfma pmc_n1m_mfma" "pmc_inst pmc_n1m_inst"; do set -- $p; f=$(pick $1 counter_collection.csv); [ -n "$f" ] && cp $f $D/$2_counter_collection.csv; done
python3 tools/make_profiles_r05.py
