"""Summarise rocprofv3 --pmc CSVs: python tools/pmc_summary.py <kernel-substring> <dir>..."""
import collections, csv, glob, sys
sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(f"{k:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        ts = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
        if ts:
            print(f"  durations us: {[round(t, 1) for t in ts]}")
