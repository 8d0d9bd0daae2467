"""Per-kernel averages of a rocprofv3 --pmc pass: python tools/pmc_agg.py <dir>... [--filter=i8gemm]
Sums every counter over its dimensions (XCDs, SEs, instances) per dispatch, then averages over the dispatches of a (kernel, grid) pair."""
import csv, glob, sys, collections
flt = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--filter=")]
flt = flt[0] if flt else ""
for d in [a for a in sys.argv[1:] if not a.startswith("--")]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if flt not in r["Kernel_Name"]:
                continue
            per[(r["Kernel_Name"][:48], r["Grid_Size"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        agg = collections.defaultdict(list)
        for (k, g, _, c), v in per.items():
            agg[(k, g, c)].append(v)
        for (k, g, c), v in sorted(agg.items()):
            print("%-48s grid %-9s %-36s n=%-3d avg %.4g" % (k, g, c, len(v), sum(v) / len(v)))
