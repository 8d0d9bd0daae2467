#!/bin/bash
# Copies the summaries of the final measurement run (tools/leases/gpu_final_r3.sh -> gpurun_out/r3f) into profiles/r03 and
# writes profiles/r03/traffic.json (HBM bytes per sample of the dominant kernel by the PMC passes at N = 10 M),
# traffic_cfg4.json (the generic pipeline's HBM bytes per sample at d = 1024, k = 64) and cliff.md.
set -e
R=/root/repo; S=$R/gpurun_out/r3f; D=$R/profiles/r03
mkdir -p $D
first() { ls -t $(find "$1" -name "$2") | head -1; }
for f in bench_n10m bench_n10m_fp64gram bench_n10m_em4 bench_cfg5 bench_cfg4; do cp $S/$f.json $D/$f.json; done
cp $S/kt_bench.json $D/bench_n10m_under_rocprof.json
cp $(first $S/kt "*kernel_stats.csv") $D/bench_n10m_kernel_stats.csv
cp $(first $S/kt_fp64gram "*kernel_stats.csv") $D/bench_n10m_fp64gram_kernel_stats.csv
cp $(first $S/pmc_fetch "*counter_collection.csv") $D/pmc_n10m_FETCH_SIZE_counter_collection.csv
cp $(first $S/pmc_write "*counter_collection.csv") $D/pmc_n10m_WRITE_SIZE_counter_collection.csv
cp $(first $S/pmc_mfma "*counter_collection.csv") $D/pmc_n1m_mfma_counter_collection.csv
cp $(first $S/pmc_inst "*counter_collection.csv") $D/pmc_n1m_inst_counter_collection.csv
cp $(first $S/kt_cfg4 "*kernel_stats.csv") $D/bench_cfg4_kernel_stats.csv
cp $(first $S/pmc_cfg4 "*counter_collection.csv") $D/pmc_cfg4_counter_collection.csv
cp $(first $S/pmc_cfg4_fetch "*counter_collection.csv") $D/pmc_cfg4_FETCH_SIZE_counter_collection.csv
cp $(first $S/pmc_cfg4_write "*counter_collection.csv") $D/pmc_cfg4_WRITE_SIZE_counter_collection.csv
cp $(first $S/kt_cfg5 "*kernel_stats.csv") $D/bench_cfg5_kernel_stats.csv
cp $(first $S/kt_d256_k11 "*kernel_stats.csv") $D/cliff_d256_k11_kernel_stats.csv
cp $(first $S/kt_d200_k16 "*kernel_stats.csv") $D/cliff_d200_k16_kernel_stats.csv
cp $(first $S/kt_d512_k10 "*kernel_stats.csv") $D/cliff_d512_k10_kernel_stats.csv
cp $S/passes.log $D/passes_n4m.log
cp $S/transfer.log $D/transfer.log
grep "em8 cycles" $S/timing.err | tail -1 > $D/em8_phase_table.log || true
grep "llk2 cycles" $S/timing_passes.log | tail -1 > $D/llk2_phase_table.log || true
grep "em16 estep" $S/timing16.err | tail -1 > $D/em16_phase_table.log || true
grep -v amdgpu $S/llk2_resident.log > $D/llk2_resident.log || true
grep -v "amdgpu\|cycles" $S/em16_check.log > $D/em16_check.log || true
mkdir -p $D/cliff
cp $S/cliff_d*.json $D/cliff/
python3 - <<PY
import csv, json, subprocess, glob, os
D = "$D"
def mean(path, counter, needle):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    return (sum(v) / len(v), len(v)) if v else (0.0, 0)
commit = subprocess.check_output(["git", "-C", "$R", "rev-parse", "--short", "HEAD"]).decode().strip()
K = "em8_kernel<10, false, false>"
f, nf = mean(D + "/pmc_n10m_FETCH_SIZE_counter_collection.csv", "FETCH_SIZE", K)
w, nw = mean(D + "/pmc_n10m_WRITE_SIZE_counter_collection.csv", "WRITE_SIZE", K)
cal, _ = mean(D + "/pmc_n10m_FETCH_SIZE_counter_collection.csv", "FETCH_SIZE", "column_presence_kernel")
n = 10_000_000
known = n * 256 * 8
ratio = known / (cal * 1024)
total = ratio * f * 1024 + w * 1024
out = {
    "commit": commit, "kernel": "ppca::" + K, "n_samples": n,
    "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches": nf,
    "calibration": {"kernel": "column_presence_kernel", "known_bytes": known, "FETCH_SIZE_KB": cal, "correction": ratio},
    "hbm_bytes_per_launch": total, "hbm_bytes_per_sample": total / n,
    "algorithmic_bytes_per_sample": 8 * 256 + 256 / 8 + 8,
    "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python3 tools/pmc_run.py with PMC_N=10000000; "
           "FETCH_SIZE corrected by the ratio measured on column_presence_kernel (every element of X read once) in the same run",
}
json.dump(out, open(D + "/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
# config 4: all kernels of the generic pipeline over tools/pmc_generic.py's two EM steps on PMC_N = 175000 rows
def total_counter(path, counter, skip=("synth_", "column_presence", "fill")):
    tot = 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and not any(s in r["Kernel_Name"] for s in skip):
            tot += float(r["Counter_Value"])
    return tot
n4, steps4 = 175000, 2
ft = total_counter(D + "/pmc_cfg4_FETCH_SIZE_counter_collection.csv", "FETCH_SIZE")
wt = total_counter(D + "/pmc_cfg4_WRITE_SIZE_counter_collection.csv", "WRITE_SIZE")
b4 = (ratio * ft * 1024 + wt * 1024) / (n4 * steps4)
json.dump({"commit": commit, "n_samples": n4, "em_steps": steps4, "FETCH_SIZE_KB_total": ft, "WRITE_SIZE_KB_total": wt,
           "fetch_correction": ratio, "hbm_bytes_per_sample": b4, "algorithmic_bytes_per_sample": 8 * 1024 + 1024 / 8 + 8,
           "how": "sum over every kernel of the generic pipeline (tools/pmc_generic.py: two EM steps, d = 1024, k = 64, 50 % masked) of 2 x FETCH_SIZE (gfx950 "
                  "correction as calibrated above) + WRITE_SIZE, per sample and EM step"}, open(D + "/traffic_cfg4.json", "w"), indent=1)
print("config 4 HBM bytes per sample and step:", b4)
# cliff table
rows = []
for p in sorted(glob.glob(D + "/cliff/cliff_d*.json")):
    j = json.load(open(p)); c = j["config"]; r = j["roofline"]
    rows.append((c["d"], c["state_size"], j["value"], j["ms_per_step"], r["frac"], r["kernel"]))
rows.sort(key=lambda t: (t[1], t[0]))
with open(D + "/cliff.md", "w") as fh:
    fh.write("# The shapes around the fused kernel (N = 2 M, 30 % masked, one MI355X; `python bench.py --n 2000000 --d D --k K --steps 4 --warmup 1 --no-cpu`)\n\n")
    fh.write("| d | k | EM it/s | ms / iteration | fraction of the algorithmic fp64 roof | path |\n|---|---|---|---|---|---|\n")
    for d_, k_, v, ms, fr, kn in rows:
        fh.write(f"| {d_} | {k_} | {v:.1f} | {ms:.2f} | {fr:.3f} | {kn} |\n")
print(open(D + "/cliff.md").read())
PY
ls -la $D
