#!/usr/bin/env python3
"""Per-kernel averages from rocprofv3 CSV output: tools/pmc_summary.py <dir> -> JSON on stdout.
Reads *kernel_stats.csv (durations) and *counter_collection.csv (FETCH_SIZE / WRITE_SIZE, KiB per dispatch) found under
<dir> (several passes may live in sub-directories)."""
import collections, csv, glob, json, os, re, sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('fhelin::', '')
    m = re.match(r'(?:void )?(\w+(?:<[^>]*>)?)', n)
    return m.group(1) if m else n[:40]


def main():
    root = sys.argv[1]
    out = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(root, '**', '*kernel_stats.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Name'])
            out[k]['calls'] = int(r['Calls'])
            out[k]['avg_us'] = round(float(r['AverageNs']) / 1e3, 2)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        for c, v in cs.items():
            out[k][c + '_KiB_per_launch'] = round(sum(v) / len(v), 1)
            out[k][c + '_launches'] = len(v)
    print(json.dumps(out, indent=1, sort_keys=True))


main()
