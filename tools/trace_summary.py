#!/usr/bin/env python
"""Timeline summary of a rocprofv3 --kernel-trace CSV: per-kernel durations, concurrency, idle gaps.
usage: python tools/trace_summary.py <kernel_trace.csv> [last_n_kernels]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-28:], r['Queue_Id']) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(ks) // 3
ks = ks[-n:]
t0, t1 = ks[0][0], max(k[1] for k in ks)
ev = []
for s, e, _, _ in ks:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = defaultdict(int); cur = 0; last = t0
for t, d in ev:
    busy[cur] += t - last; last = t; cur += d
tot = t1 - t0
print(f"window {tot/1e3:.0f} us, {len(ks)} kernels; time with k kernels in flight:", {k: f"{v/tot*100:.0f}%" for k, v in sorted(busy.items())})
dur = defaultdict(list)
for s, e, nm, q in ks:
    dur[nm].append((e - s) / 1e3)
for nm, v in dur.items():
    v.sort()
    print(f"  {nm:30s} n={len(v):4d} avg {sum(v)/len(v):7.1f} us  p50 {v[len(v)//2]:7.1f}  max {v[-1]:7.1f}  sum {sum(v)/1e3:7.2f} ms")
print("queues:", sorted(set(k[3] for k in ks)))
