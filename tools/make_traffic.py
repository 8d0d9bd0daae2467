"""HBM bytes per sample of the dominant EM kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_run.py),
corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE is calibrated in the same run on a kernel of
known traffic (column_presence_kernel reads every element of X exactly once).

    python tools/make_traffic.py <fetch_dir> <write_dir> <n_samples> <out.json> [commit]
"""
import csv, glob, json, os, sys


def first_csv(d):
    return sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]


def mean(path, counter, needle):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]
    return (sum(v) / len(v), len(v)) if v else (0.0, 0)


def main():
    fdir, wdir, n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    commit = sys.argv[5] if len(sys.argv) > 5 else None
    fcsv, wcsv = first_csv(fdir), first_csv(wdir)
    K = "em9_kernel<10, false, false>"  # the default EM kernel since the end of round 4 (PPCA_EM9=0: em8_kernel)
    if mean(fcsv, "FETCH_SIZE", K)[1] == 0:
        K = "em8_kernel<10, false, false>"
    f, nf = mean(fcsv, "FETCH_SIZE", K)
    w, _ = mean(wcsv, "WRITE_SIZE", K)
    cal, _ = mean(fcsv, "FETCH_SIZE", "column_presence_kernel")
    known = n * 256 * 8
    ratio = known / (cal * 1024)
    total = ratio * f * 1024 + w * 1024
    j = {"commit": commit, "kernel": "ppca::" + K, "n_samples": n, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches": nf,
         "calibration": {"kernel": "column_presence_kernel", "known_bytes": known, "FETCH_SIZE_KB": cal, "correction": ratio},
         "hbm_bytes_per_launch": total, "hbm_bytes_per_sample": total / n, "algorithmic_bytes_per_sample": 8 * 256 + 256 / 8 + 8,
         "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python3 tools/pmc_run.py with PMC_N=%d; "
                "FETCH_SIZE corrected by the ratio measured on column_presence_kernel (every element of X read once) in the same run" % n}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j, indent=1))


if __name__ == "__main__":
    main()
