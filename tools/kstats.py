#!/usr/bin/env python
"""Print a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py <dir or csv>"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
for r in csv.DictReader(open(p)):
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}  {r['Percentage']}%")
