"""Per-kernel timeline of one training step from a rocprofv3 --kernel-trace results database (rocpd sqlite):
python tools/step_timeline.py <results.db> [which] [marker-kernel-substring]
Prints the kernels between launches which-1 and which (default: the last two; negative counts from the end) of the
marker kernel (default: the kernel that ends a step, see below): start and end offset, duration, hardware
queue, name -- kernels of a step that run side by side show up on different queues with overlapping times."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
# the kernel that ends a step: the cut rows' launch of the backward pass's last step when it carries the table's Adam
# update, else the embedding-table Adam kernel
markers = [sys.argv[3]] if len(sys.argv) > 3 else ["set_hub_kernel<5>", "adam_table_kernel"]
idx = next(ix for ix in ([i for i, r in enumerate(rows) if m in r[0]] for m in markers) if ix)
a, b = idx[which - 1] + 1, idx[which] + 1
t0, busy = rows[a][1], 0
for name, s, e, q in rows[a:b]:
    short = name.replace("(anonymous namespace)::", "").split("(")[0][:70]
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  q{q}  {short}")
    busy += e - s
print(f"step: {b - a} kernels, span {(max(r[2] for r in rows[a:b]) - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
