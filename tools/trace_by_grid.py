#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv: per kernel name and grid size, the number of dispatches, mean duration and the mean gap
to the previous dispatch on the same queue.  usage: trace_by_grid.py <dir-or-csv> <kernel-substring>"""
import csv, glob, sys
from collections import defaultdict
path, want = sys.argv[1], sys.argv[2]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_end = {}
acc = defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    q = r.get("Queue_Id", "0")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - last_end.get(q, s)
    last_end[q] = e
    if want in r["Kernel_Name"]:
        key = (r["Kernel_Name"].split("(")[0][-24:], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)), r.get("LDS_Block_Size", ""))
        a = acc[key]; a[0] += 1; a[1] += e - s; a[2] += gap
for k in sorted(acc):
    n, d, g = acc[k]
    print("%-26s grid %8d lds %7s  n %6d  dur %8.2f us  gap-before %8.2f us" % (k[0], k[1], k[2], n, d / n / 1e3, g / n / 1e3))
