#!/usr/bin/env python3
"""Builds profiles/<tag>_traffic.json from rocprofv3 --pmc passes (FETCH_SIZE in one run, WRITE_SIZE in
another) of bench.py: mean bytes per dispatch of the propagation kernels, FETCH_SIZE doubled (gfx950 tallies
the 128-B requests of 16-B/lane reads at 64 B), summed per fused layer launch.

    make_traffic_json.py OUT.json  tmall FETCH_DIR WRITE_DIR  [industry FETCH_DIR WRITE_DIR]
"""
import csv, glob, json, sys, collections

# the kernels of one layer launch, per forward layer launch of the timed step (the figure is the mean over the launches)
KERNELS = {"tmall": [["set_layer_kernel<0>", "set_hub_kernel<0>"], ["set_layer_kernel<1>", "set_hub_kernel<1>"]],
           "industry": [["lds_rows_kernel", "window_rows_kernel", "hub_reduce_kernel"]]}


def means(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


out = {"note": "HBM-side (fabric) bytes per propagation-layer launch from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE "
               "passes (separate runs of `bench.py --steps 4 --warmup 2 --no-cpu-baseline [--workload industry]`), means over "
               "all dispatches of the kernel; counter values are KiB; FETCH_SIZE doubled (gfx950 counts 128-B requests as "
               "64 B for 16-B/lane reads), WRITE_SIZE as reported. Infinity-Cache hits are included in these counters."}
args = sys.argv[2:]
for k in range(0, len(args), 3):
    wl, fd, wd = args[k:k + 3]
    fm, wm = means(fd, "FETCH_SIZE"), means(wd, "WRITE_SIZE")
    entry, totals = {}, []
    for launch in KERNELS[wl]:
        total = 0.0
        for kern in launch:
            fb = sum(v for n, v in fm.items() if kern in n) * 1024 * 2
            wb = sum(v for n, v in wm.items() if kern in n) * 1024
            entry[kern] = {"fetch_bytes": fb, "write_bytes": wb}
            total += fb + wb
        totals.append(total)
    entry["traffic_bytes_per_layer_launch"] = sum(totals) / len(totals)
    entry["per_launch"] = totals
    out[wl] = entry
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out["source_digest"] = bench.source_digest()      # bench.py refuses the file once the kernel sources change
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
