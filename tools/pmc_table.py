#!/usr/bin/env python3
"""Per-kernel averages of every counter in one or more rocprofv3 *_counter_collection.csv files.
usage: pmc_table.py a_counter_collection.csv [b_counter_collection.csv ...]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
names = []
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:40]
        c = r["Counter_Name"]
        if c not in names:
            names.append(c)
        a = acc[k][c]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
print("| kernel | launches | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k, cs in sorted(acc.items()):
    n = max(v[0] for v in cs.values())
    print(f"| {k} | {n} | " + " | ".join(f"{cs[c][1] / cs[c][0]:.4g}" if c in cs else "" for c in names) + " |")
