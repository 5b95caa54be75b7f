"""Reads a rocprofv3 kernel trace of tools/shard_probe.py and prints, for a few forwards, when the stand-in copy of
the exchange ran relative to the lds_rows_kernel launch it is meant to overlap.  Usage: overlap_check.py trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
shown = 0
for k, r in enumerate(rows):
    if "lds_rows_kernel" not in r["Kernel_Name"]:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    near = [q for q in rows[max(0, k - 6):k + 8] if "copy" in q["Kernel_Name"].lower() or "elementwise" in q["Kernel_Name"].lower()]
    for q in near:
        qs, qe = int(q["Start_Timestamp"]), int(q["End_Timestamp"])
        print(f"lds_rows [{0:7.1f},{(e - s) / 1e3:7.1f}] us   copy [{(qs - s) / 1e3:7.1f},{(qe - s) / 1e3:7.1f}] us  queue {q.get('Queue_Id')}/{r.get('Queue_Id')}")
    shown += 1
    if shown >= 12:
        break
