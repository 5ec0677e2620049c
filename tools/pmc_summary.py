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
            keys = [k]
            # the dominant kernel once more, restricted to launches over a full 4096-sentence batch (1024 workgroups x 256 work-items):
            # the run also holds ragged last batches and single-launch small calls, and "per dispatch" of the mix is not "per batch"
            if "k_tokenize_pool" in k and row.get("Grid_Size") == "262144":
                keys.append(k + " [full 4096-sentence launches]")
            for kk in keys:
                a = agg[kk][c]
                a[0] += float(row["Counter_Value"])
                seen[(kk, c)].add(row["Dispatch_Id"])
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
