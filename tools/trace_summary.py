#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV: GPU busy/idle over the last WINDOW seconds, per-kernel totals, gap histogram.
usage: trace_summary.py kernel_trace.csv [window_s]"""
import collections
import csv
import re
import sys


def short(n):
    m = re.search(r'(\w+)(<[^>]*>)?\(', n.replace('(anonymous namespace)', ''))
    return (m.group(1) + (m.group(2) or '')) if m else n[:30]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    win_s = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    tend = int(rows[-1]['End_Timestamp'])
    win = [r for r in rows if int(r['Start_Timestamp']) > tend - win_s * 1e9]
    # union of busy intervals (kernels of different streams may overlap)
    busy, cur_s, cur_e = 0, None, None
    for r in win:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = int(win[-1]['End_Timestamp']) - int(win[0]['Start_Timestamp'])
    print(f"window {span/1e6:.1f} ms, launches {len(win)}, GPU busy (union) {busy/1e6:.1f} ms, idle {100*(1-busy/span):.1f}%")
    tot = collections.defaultdict(float)
    cnt = collections.Counter()
    for r in win:
        k = short(r['Kernel_Name'])
        tot[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        cnt[k] += 1
    for k, v in sorted(tot.items(), key=lambda x: -x[1])[:16]:
        print(f"  {k:40s} {cnt[k]:7d} {v:8.1f} ms  avg {v/cnt[k]*1e3:7.1f} us")
    gaps = collections.Counter()
    gsum = collections.defaultdict(float)
    prev_e = None
    for r in win:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if prev_e is not None and s > prev_e:
            g = (s - prev_e) / 1e3
            b = '<10us' if g < 10 else '<50us' if g < 50 else '<500us' if g < 500 else '<5ms' if g < 5000 else '>=5ms'
            gaps[b] += 1
            gsum[b] += g / 1e3
        prev_e = e if prev_e is None else max(prev_e, e)
    print("  gaps:", {b: (gaps[b], round(gsum[b], 1)) for b in ['<10us', '<50us', '<500us', '<5ms', '>=5ms']})
    # the largest gaps with the kernels on either side (what the host was doing while the GPU had nothing queued)
    big = []
    prev_e, prev_k = None, None
    for r in win:
        s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if prev_e is not None and s_ > prev_e:
            big.append(((s_ - prev_e) / 1e3, prev_k, short(r['Kernel_Name'])))
        if prev_e is None or e_ > prev_e:
            prev_e, prev_k = e_, short(r['Kernel_Name'])
    pair = collections.defaultdict(lambda: [0, 0.0])
    for g, a, b in big:
        if g >= 30:
            pair[(a, b)][0] += 1
            pair[(a, b)][1] += g
    print("  gaps >= 30 us by (kernel before -> kernel after), total us:")
    for (a, b), (c, t) in sorted(pair.items(), key=lambda x: -x[1][1])[:14]:
        print(f"    {a:32s} -> {b:32s} x{c:4d} {t:9.0f} us")


main()
