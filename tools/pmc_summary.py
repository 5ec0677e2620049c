#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection CSVs (one dir per --pmc pass) per kernel.
usage: python tools/pmc_summary.py <outdir> [--json out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f, newline="") as fh:
        seen = defaultdict(set)
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            c = row["Counter_Name"]
            a = agg[k][c]
            a[0] += float(row["Counter_Value"])
            seen[(k, c)].add(row["Dispatch_Id"])
        for (k, c), ids in seen.items():
            agg[k][c][1] += len(ids)
out = {}
for k, cs in agg.items():
    out[k] = {c: {"sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / max(v[1], 1)} for c, v in cs.items()}
    print(k)
    for c, v in sorted(out[k].items()):
        print(f"   {c:32s} per-dispatch {v['per_dispatch']:16.1f}   (n={v['dispatches']})")
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
