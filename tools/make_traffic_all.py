"""HBM bytes per sample and step summed over EVERY kernel of a multi-kernel path (the split pipeline of config 4, a mixture
iteration of config 5) from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE is calibrated in the same run on a kernel of known traffic
(column_presence_kernel reads every element of X exactly once: Dataset.empty_dimensions()).

    python tools/make_traffic_all.py <fetch_dir> <write_dir> <n_samples> <d> <steps> <out.json> <commit> <what>
"""
import csv, glob, json, os, sys

SKIP = ("column_presence_kernel", "synth_", "FillFunctor", "copyBuffer", "fillBuffer", "elementwise_kernel")


def first_csv(d):
    return sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]


def rows(path, counter):
    """(kernel, value) of the dispatches BETWEEN the last two column_presence_kernel launches (the drivers bracket the steady-state
    steps with Dataset.empty_dimensions()), plus those two launches themselves (the calibration)."""
    r = [(int(x["Dispatch_Id"]), x["Kernel_Name"], float(x["Counter_Value"])) for x in csv.DictReader(open(path)) if x["Counter_Name"] == counter]
    r.sort()
    marks = [i for i, (_, k, _) in enumerate(r) if "column_presence_kernel" in k]
    if len(marks) >= 2:
        r = r[marks[-2]:marks[-1] + 1]
    return [(k, v) for _, k, v in r]


def main():
    fdir, wdir, n, d, steps, out, commit, what = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], sys.argv[7], sys.argv[8]
    f, w = rows(first_csv(fdir), "FETCH_SIZE"), rows(first_csv(wdir), "WRITE_SIZE")
    cal = [v for k, v in f if "column_presence_kernel" in k]
    known = n * d * 8
    ratio = known / (sum(cal) / len(cal) * 1024)
    use = lambda k: not any(s in k for s in SKIP)
    ft, wt = sum(v for k, v in f if use(k)), sum(v for k, v in w if use(k))
    per_kernel = {}
    short = lambda k: k.replace("(anonymous namespace)::", "").split("(")[0][:80]
    for k, v in f:
        if use(k):
            per_kernel[short(k)] = per_kernel.get(short(k), 0.0) + ratio * v * 1024
    for k, v in w:
        if use(k):
            per_kernel[short(k)] = per_kernel.get(short(k), 0.0) + v * 1024
    total = ratio * ft * 1024 + wt * 1024
    top = sorted(per_kernel.items(), key=lambda kv: -kv[1])[:8]
    j = {"commit": commit, "what": what, "n_samples": n, "steps": steps, "FETCH_SIZE_KB_total": ft, "WRITE_SIZE_KB_total": wt,
         "fetch_correction": ratio, "calibration": {"kernel": "column_presence_kernel", "known_bytes": known, "launches": len(cal)},
         "hbm_bytes_per_sample": total / n / steps,
         "largest_kernels_bytes_per_sample": {k: v / n / steps for k, v in top},
         "how": "sum over every kernel of the path of corrected FETCH_SIZE + WRITE_SIZE (separate rocprofv3 --pmc passes with --kernel-trace), "
                "per sample and step; fabric bytes (Infinity-Cache hits included)"}
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j, indent=1))


if __name__ == "__main__":
    main()
