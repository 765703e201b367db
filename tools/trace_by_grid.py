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
