#!/usr/bin/env python3
"""Prints the kernel sequence of one steady-state frame from a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# frame marker: the resolve (one launch per frame; the blit is fused into it)
idx = [i for i, r in enumerate(rows) if 'k_resolve_opaque' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
tot = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tot += e - s
    if (e - s) > 8000 or 'raster' in r['Kernel_Name'] or 'cull' in r['Kernel_Name']:
        print(f"+{(s - int(rows[a]['End_Timestamp']))/1e3:8.1f} us  {(e-s)/1e3:7.1f} us  queue {r['Queue_Id']:>2}  grid {r['Grid_Size_X']:>8}  {r['Kernel_Name'][:48]}")
print("sum of kernel durations in frame: %.1f us, wall %.1f us" % (tot / 1e3, (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3))
