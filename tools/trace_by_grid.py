#!/usr/bin/env python3
"""Per-kernel, per-launch-shape durations from a rocprofv3 kernel trace CSV (which launch sizes run below the
microbenchmark rate).  usage: trace_by_grid.py kernel_trace.csv [name-substring ...]"""
import collections
import csv
import re
import sys


def short(n):
    m = re.search(r'(\w+)(<[^>]*>)?\(', n.replace('(anonymous namespace)', ''))
    return (m.group(1) + (m.group(2) or '')) if m else n[:30]


rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
# TRACE_WINDOW_S=<seconds>: only the launches of the last so many seconds of the trace (the timed passes of a bench run)
import os
if os.environ.get("TRACE_WINDOW_S"):
    tend = max(int(r['End_Timestamp']) for r in rows)
    rows = [r for r in rows if int(r['Start_Timestamp']) > tend - float(os.environ["TRACE_WINDOW_S"]) * 1e9]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = short(r['Kernel_Name'])
    if want and not any(w in k for w in want):
        continue
    wg = [int(r[f'Workgroup_Size_{a}']) for a in 'XYZ']
    gr = [int(r[f'Grid_Size_{a}']) // max(1, w) for a, w in zip('XYZ', wg)]
    key = (k, tuple(gr), wg[0])
    agg[key][0] += 1
    agg[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = collections.defaultdict(float)
for (k, g, w), (c, us) in agg.items():
    tot[k] += us
for k in sorted(tot, key=lambda x: -tot[x]):
    print(f"{k}  total {tot[k]/1e3:.2f} ms")
    for (kk, g, w), (c, us) in sorted(agg.items(), key=lambda x: -x[1][1]):
        if kk == k:
            print(f"    grid {str(g):20s} wg {w:4d} launches {c:6d} avg {us/c:8.1f} us  total {us/1e3:8.2f} ms")

# size classes over the selected kernels: where the launches sit and what a workgroup costs there
cls = collections.defaultdict(lambda: [0, 0, 0.0])
for (k, g, w), (c, us) in agg.items():
    n = g[0] * g[1] * g[2]
    b = 0 if n < 512 else 1 if n < 1024 else 2 if n < 2048 else 3 if n < 4096 else 4 if n < 8192 else 5
    cls[b][0] += c
    cls[b][1] += c * n
    cls[b][2] += us
names = ["<512", "512-1023", "1024-2047", "2048-4095", "4096-8191", ">=8192"]
print("workgroups per launch: launches, total ms, ns per workgroup")
for b in sorted(cls):
    print(f"    {names[b]:10s} {cls[b][0]:7d} {cls[b][2]/1e3:9.2f} ms {cls[b][2]*1e3/max(1, cls[b][1]):7.1f} ns/wg")
