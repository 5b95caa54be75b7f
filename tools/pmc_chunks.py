#!/usr/bin/env python3
"""Per-run means of a rocprofv3 --pmc counter for one kernel: dispatches in order, cut into chunks of N."""
import csv, glob, sys, collections
d, pat, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
for k in range(0, len(ids), n):
    chunk = [rows[i] for i in ids[k:k + n]]
    names = sorted(chunk[0])
    print(k // n, {c: round(sum(x[c] for x in chunk) / len(chunk), 1) for c in names})
