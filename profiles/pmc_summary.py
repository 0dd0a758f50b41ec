#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean counter value per
dispatch).  usage: pmc_summary.py <counter_collection.csv> [substring filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace('sage::(anonymous namespace)::', '').replace('void ', '')
    if flt in name:
        acc[(name[:80], r['Counter_Name'], r['Grid_Size'] if 'Grid_Size' in r else '')].append(float(r['Counter_Value']))
for (name, ctr, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name:80s} grid={grid:>9s} {ctr:12s} n={len(v):4d} mean={sum(v) / len(v):14.1f} max={max(v):14.1f}")
