#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table (markdown/CSV-ish).
usage: python tools/rocpd_stats.py <results.db> [--per-dispatch KERNEL_SUBSTR]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, start, end, grid_size_x, workgroup_size_x from kernels"))
    if "--per-dispatch" in sys.argv:
        key = sys.argv[sys.argv.index("--per-dispatch") + 1]
        sel = [r for r in rows if key in r[0]]
        sel.sort(key=lambda r: r[1])
        for r in sel[:200]:
            print(f"{(r[2]-r[1])/1e3:10.1f} us  grid={r[3]} wg={r[4]}  {r[0][:90]}")
        return
    stats = {}
    for name, s, e, *_ in rows:
        d = stats.setdefault(name, [0, 0, 1 << 62, 0])
        dur = e - s
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    total = sum(d[1] for d in stats.values()) or 1
    print("| kernel | calls | total_ms | avg_us | min_us | max_us | pct |")
    print("|---|---|---|---|---|---|---|")
    for name, d in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print(f"| {name[:100]} | {d[0]} | {d[1]/1e6:.3f} | {d[1]/d[0]/1e3:.2f} | {d[2]/1e3:.2f} | {d[3]/1e3:.2f} | {100*d[1]/total:.1f} |")


if __name__ == "__main__":
    main()
