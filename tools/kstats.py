#!/usr/bin/env python3
"""Prints a rocprofv3 --stats kernel_stats.csv as a short table (top N kernels)."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:n]:
    print(f"{r['Name'][:86]:86s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} "
          f"min_us={float(r['MinNs'])/1e3:8.2f} max_us={float(r['MaxNs'])/1e3:8.2f} pct={float(r['Percentage']):6.2f}")
