#!/usr/bin/env python3
"""Condense a rocprofv3 --pmc output directory (…_counter_collection.csv) into per-kernel means per launch (JSON on stdout).
usage: pmc_summary.py DIR [DIR ...]   -- several directories (separate passes) are merged by kernel name."""
import csv, glob, json, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])          # one row per XCD / dimension: sum them
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, cname), v in per_dispatch.items():
            k = names[disp].replace("(anonymous namespace)::", "").split("(")[0]
            a = acc[k][cname]; a[0] += v; a[1] += 1
out = {k: {c: {"mean_per_launch": round(a[0] / a[1], 2), "launches": a[1]} for c, a in cs.items()} for k, cs in sorted(acc.items())}
print(json.dumps(out, indent=1))
