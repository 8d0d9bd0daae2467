#!/bin/bash
# Copies the summaries of the final measurement run (tools/leases/gpu_final_r4.sh -> gpurun_out/r4f) into profiles/r04.
# profiles/r04/traffic.json was written ON THE BOX by tools/make_traffic.py before the bench line that cites it ran.
set -e
R=/root/repo; S=$R/gpurun_out/r4f; D=$R/profiles/r04
mkdir -p $D $D/cliff
first() { ls -t $(find "$1" -name "$2") | head -1; }
for f in bench_n10m bench_n10m_fp64gram bench_n10m_em4 bench_n10m_em8 bench_cfg5 bench_cfg5_llk2 bench_cfg4; do cp $S/$f.json $D/$f.json; done
cp $S/traffic.json $D/traffic.json
cp $S/mfma_peak.json $D/mfma_peak.json; cp $S/mfma_peak.txt $D/mfma_peak.txt; cp $S/solve4_ab.log $S/s4bench.log $D/
cp $S/kt_bench.json $D/bench_n10m_under_rocprof.json
cp $(first $S/kt "*kernel_stats.csv") $D/bench_n10m_kernel_stats.csv
cp $(first $S/kt_cfg5 "*kernel_stats.csv") $D/bench_cfg5_kernel_stats.csv
cp $(first $S/kt_cfg4 "*kernel_stats.csv") $D/bench_cfg4_kernel_stats.csv
cp $(first $S/kt_d200_k16 "*kernel_stats.csv") $D/cliff_d200_k16_kernel_stats.csv
cp $(first $S/pmc_fetch "*counter_collection.csv") $D/pmc_n10m_FETCH_SIZE_counter_collection.csv
cp $(first $S/pmc_write "*counter_collection.csv") $D/pmc_n10m_WRITE_SIZE_counter_collection.csv
cp $(first $S/pmc_mfma "*counter_collection.csv") $D/pmc_n1m_mfma_counter_collection.csv
cp $(first $S/pmc_inst "*counter_collection.csv") $D/pmc_n1m_inst_counter_collection.csv
cp $S/weighted_n10m.log $S/passes.log $S/passes_llk2.log $S/passes_d200_k16.log $S/additivity_n10m.log $S/outlier_probe.log $D/
grep -E "^(em9|em9front|em9back|base|front|back|shared|decoupled|accf64|noearly|noerrb|accint64) " $S/variants.log > $D/variants.log
grep "em8 wave\|em8 cycles" $S/timing.err | tail -9 > $D/em8_phase_table.log
grep "em8 wave" $S/timing_em9.err | tail -8 | sed "s/em8 wave/em9 wave/" > $D/em9_phase_table.log
cp $S/cliff_d*.json $D/cliff/
python3 - <<PY
import json, glob
D = "$D"
rows = []
for p in sorted(glob.glob(D + "/cliff/cliff_d*.json")):
    j = json.load(open(p)); c = j["config"]; r = j["roofline"]
    rows.append((c["d"], c["state_size"], j["value"], j["ms_per_step"], r["frac"], r["kernel"]))
rows.sort(key=lambda t: (t[1], t[0]))
with open(D + "/cliff.md", "w") as fh:
    fh.write("# The shapes around the fused kernel (N = 2 M, 30 % masked, one MI355X; \`python bench.py --n 2000000 --d D --k K --steps 4 --warmup 1 --no-cpu\`)\n\n")
    fh.write("| d | k | EM it/s | ms / iteration | fraction of the algorithmic fp64 roof | path |\n|---|---|---|---|---|---|\n")
    for d_, k_, v, ms, fr, kn in rows:
        fh.write(f"| {d_} | {k_} | {v:.1f} | {ms:.2f} | {fr:.3f} | {kn} |\n")
print(open(D + "/cliff.md").read())
PY
ls -la $D
