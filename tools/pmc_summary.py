#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per launch)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else "srx"
for k, v in agg.items():
    if flt not in k:
        continue
    print(k)
    for c, vals in sorted(v.items()):
        print(f"    {c:32s} {sum(vals) / len(vals):16.1f}   (n={len(vals)})")
