#!/usr/bin/env python
"""Per-dispatch view of k_tokenize_pool in a rocprofv3 --kernel-trace CSV: durations grouped by grid size.
usage: python tools/trace_pool.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
g = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_tokenize_pool" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("(")[0][-22:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(g.items(), key=lambda kv: -len(kv[1])):
    v.sort()
    print(f"{k}: n={len(v)} avg {sum(v)/len(v):.1f} us p10 {v[len(v)//10]:.1f} p50 {v[len(v)//2]:.1f} p90 {v[len(v)*9//10]:.1f} max {v[-1]:.1f}")
