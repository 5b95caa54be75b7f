"""Per-kernel timeline of one training step from a rocprofv3 --kernel-trace results database (rocpd sqlite):
python tools/step_timeline.py <results.db> [which] [marker-kernel-substring]
Prints the kernels between launches which-1 and which (default: the last two; negative counts from the end) of the
marker kernel (default: the embedding-table Adam update, which
ends a step): start offset, duration, gap to the previous kernel's end, name."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
marker = sys.argv[3] if len(sys.argv) > 3 else "adam_table_kernel"
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = idx[which - 1] + 1, idx[which] + 1
t0, prev_end, busy = rows[a][1], rows[a][1], 0
for name, s, e in rows[a:b]:
    short = name.replace("(anonymous namespace)::", "").split("(")[0][:70]
    print(f"{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {short}")
    prev_end, busy = max(prev_end, e), busy + (e - s)
print(f"step: {b - a} kernels, span {(rows[b - 1][2] - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
