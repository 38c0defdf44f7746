#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 runs: kernel-trace stats CSV + FETCH_SIZE / WRITE_SIZE counter-collection CSVs
(collected in separate --pmc passes, as MI355X_MICROARCH.md prescribes).  FETCH_SIZE/WRITE_SIZE are in KiB.
usage: pmc_summary.py <kt_kernel_stats.csv> <pmc_fetch_counter_collection.csv> <pmc_write_counter_collection.csv>"""
import csv
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}


def main():
    stats = list(csv.DictReader(open(sys.argv[1])))
    fetch = per_kernel(sys.argv[2], "FETCH_SIZE")
    write = per_kernel(sys.argv[3], "WRITE_SIZE")
    print("| kernel | calls | avg us | % GPU time | FETCH_SIZE MiB/launch (raw) | x2 (gfx950 wide-load correction) | WRITE_SIZE MiB/launch |")
    print("|---|---|---|---|---|---|---|")
    for r in stats:
        n = r["Name"]
        f, w = fetch.get(n), write.get(n)
        short = n.split("(")[0][:48]
        print(f"| {short} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} | "
              f"{'' if f is None else f'{f/1024:.2f}'} | {'' if f is None else f'{2*f/1024:.2f}'} | {'' if w is None else f'{w/1024:.2f}'} |")


if __name__ == "__main__":
    main()
