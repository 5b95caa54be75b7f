#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv values per kernel name (per dispatch)."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if pat and pat not in name:
        continue
    acc[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in cs.items():
        print(f"    {c:24s} n={len(v):4d} mean={sum(v)/len(v):16.1f} min={min(v):16.1f} max={max(v):16.1f}")
