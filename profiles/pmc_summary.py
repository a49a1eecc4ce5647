#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, the sum over dispatches
and the per-dispatch average."""
import collections
import csv
import sys

for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    disp = collections.defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            acc[(k, row["Counter_Name"])][0] += float(row["Counter_Value"])
            disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
    print(f"# {path}")
    for (k, c), (v, _) in sorted(acc.items()):
        n = len(disp[(k, c)])
        print(f"{k[:70]:70s} {c:22s} dispatches={n:4d} sum={v:.6g} per_dispatch={v / n:.6g}")
