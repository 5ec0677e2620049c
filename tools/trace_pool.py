#!/usr/bin/env python
"""Per-dispatch view of k_tokenize_pool in a rocprofv3 --kernel-trace CSV: durations grouped by grid size; with a second argument N also the LAST N
full-batch dispatches of the product kernel on their own -- the timed region of `bench.py --steps K` is its last 24 K full batches (what comes before is the
prewarm and the warmup, which run without the HIP events the timed region's launches are bracketed by): the figure bench.py's avg_kernel_ms is to be held against.
usage: python tools/trace_pool.py <kernel_trace.csv> [N]"""
import csv, sys
from collections import defaultdict
g = defaultdict(list)
full = []   # (start, duration) of the product kernel over full 4096-sentence batches (grid 1024 x 256 threads)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_tokenize_pool" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("(")[0][-22:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g[key].append(dur)
    if "pool<false" in r["Kernel_Name"] and key[1] == "262144":
        full.append((int(r["Start_Timestamp"]), dur))
def line(v):
    v = sorted(v)
    return f"n={len(v)} avg {sum(v)/len(v):.1f} us p10 {v[len(v)//10]:.1f} p50 {v[len(v)//2]:.1f} p90 {v[len(v)*9//10]:.1f} max {v[-1]:.1f}"
for k, v in sorted(g.items(), key=lambda kv: -len(kv[1])):
    print(f"{k}: {line(v)}")
if len(sys.argv) > 2 and full:
    n = int(sys.argv[2])
    full.sort()
    print(f"the last {min(n, len(full))} full-batch dispatches of the product kernel (the timed region): {line([d for _, d in full[-n:]])}")
