# gpurun_out/r6f2 (tools/leases/r6_final.sh) -> profiles/r06, then the numbers file generated FROM those files
# (tools/make_profiles_r06.py): nothing in profiles/r06/numbers.md is typed by hand.
set -e
cd "$(dirname "$0")/../.."
S=gpurun_out/r6f2; D=profiles/r06
mkdir -p $D/cliff
for f in mfma_peak.txt mfma_peak.json traffic.json traffic_cfg4.json traffic_cfg5.json bench_n10m.json bench_n10m_outliers1.json bench_n10m_outliers10.json \
         bench_n10m_outliers1000.json bench_n10m_outliers10_heavy0.json bench_n10m_fp64gram.json bench_n1250k.json weighted_n10m.log bench_cfg5.json \
         bench_cfg5_component_by_component.json bench_cfg4.json bench_cfg2.json bench_cfg1.json passes.log passes_recon8_off.log passes_d200_k10.log \
         passes_d200_k16.log additivity_n10m.log outlier_probe.log soak_em.log fuzz1.log fuzz2.log; do
  [ -f $S/$f ] && cp $S/$f $D/$f
done
cp $S/cliff_d*_k*.json $D/cliff/ 2>/dev/null || true
pick() { ls -t $S/$1/*/*$2 2>/dev/null | head -1; }  # (the newest: gpurun_out/ keeps the files of earlier leases)
for p in "kt bench_n10m" "kt_n1250k n1250k" "kt_out10 bench_n10m_outliers10" "kt_cfg5 bench_cfg5" "kt_cfg4 bench_cfg4" "kt_passes passes"; do set -- $p; f=$(pick $1 kernel_stats.csv); [ -n "$f" ] && cp $f $D/$2_kernel_stats.csv; done
for p in "pmc_fetch pmc_n10m_FETCH_SIZE" "pmc_write pmc_n10m_WRITE_SIZE" "pmc_mfma pmc_n1m_mfma" "pmc_inst pmc_n1m_inst"; do set -- $p; f=$(pick $1 counter_collection.csv); [ -n "$f" ] && cp $f $D/$2_counter_collection.csv; done
python3 tools/make_profiles_r06.py
