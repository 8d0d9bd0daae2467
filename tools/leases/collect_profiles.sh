#!/bin/bash
# Copies the summaries of the final measurement run (tools/leases/gpu_final.sh -> gpurun_out/r2f) into profiles/r02 and
# writes profiles/r02/traffic.json (HBM bytes per sample of the dominant kernel by the PMC passes at N = 10 M).
set -e
R=/root/repo; S=$R/gpurun_out/r2f; D=$R/profiles/r02
mkdir -p $D
cp $S/bench_n10m.json $D/bench_n10m.json
cp $S/kt_bench.json $D/bench_n10m_under_rocprof.json
cp $(ls -t $(find $S/kt -name "*kernel_stats.csv") | head -1) $D/bench_n10m_kernel_stats.csv
cp $(ls -t $(find $S/pmc_fetch -name "*counter_collection.csv") | head -1) $D/pmc_n10m_FETCH_SIZE_counter_collection.csv
cp $(ls -t $(find $S/pmc_write -name "*counter_collection.csv") | head -1) $D/pmc_n10m_WRITE_SIZE_counter_collection.csv
cp $(ls -t $(find $S/pmc_mfma -name "*counter_collection.csv") | head -1) $D/pmc_n1m_mfma_counter_collection.csv
cp $S/bench_cfg4.json $D/bench_cfg4.json
cp $(ls -t $(find $S/kt_cfg4 -name "*kernel_stats.csv") | head -1) $D/bench_cfg4_kernel_stats.csv
cp $(ls -t $(find $S/kt_mix -name "*kernel_stats.csv") | head -1) $D/mix_cfg5_kernel_stats.csv
cp $S/mix_cfg5.log $D/mix_cfg5_time.log
[ -f $S/passes.log ] && cp $S/passes.log $D/passes_n4m.log
python3 - <<PY
import csv, json, subprocess, collections
D = "$D"
def mean(path, counter, needle):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)
K = "pass_kernel<10, true, 4, true, false>"
f, nf = mean(D + "/pmc_n10m_FETCH_SIZE_counter_collection.csv", "FETCH_SIZE", K)
w, nw = mean(D + "/pmc_n10m_WRITE_SIZE_counter_collection.csv", "WRITE_SIZE", K)
cal, _ = mean(D + "/pmc_n10m_FETCH_SIZE_counter_collection.csv", "FETCH_SIZE", "column_presence_kernel")
n = 10_000_000
known = n * 256 * 8
ratio = known / (cal * 1024)
total = ratio * f * 1024 + w * 1024
out = {
    "commit": subprocess.check_output(["git", "-C", "$R", "rev-parse", "--short", "HEAD"]).decode().strip(),
    "kernel": "ppca::" + K,
    "n_samples": n,
    "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches": nf,
    "calibration": {"kernel": "column_presence_kernel", "known_bytes": known, "FETCH_SIZE_KB": cal, "correction": ratio},
    "hbm_bytes_per_launch": total,
    "hbm_bytes_per_sample": total / n,
    "algorithmic_bytes_per_sample": 8 * 256 + 256 / 8 + 8,
    "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python3 tools/pmc_run.py with PMC_N=10000000; "
           "FETCH_SIZE corrected by the ratio measured on column_presence_kernel (every element of X read once) in the same run",
}
json.dump(out, open(D + "/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
ls -la $D
