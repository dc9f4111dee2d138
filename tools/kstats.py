#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as a table (helper for reading gpurun_out/prof_*)."""
import csv, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = list(csv.DictReader(open(files[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    print("%-28s calls %6s  total %9.1f us  avg %8.2f us  %5.1f%%" % (name[:28], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
