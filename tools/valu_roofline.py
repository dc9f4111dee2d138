#!/usr/bin/env python3
"""VALU-issue roofline of the kernels of a profile: a CDNA compute unit starts at most one vector instruction per cycle (four SIMDs,
one wave64 instruction per four cycles each), so  SQ_INSTS_VALU / (compute units x clock)  is the shortest time the kernel's vector
instructions can issue in; its ratio to the measured duration says how much of that roofline the kernel uses.
usage: valu_roofline.py <tag> [clock GHz, default 2.3]      (profiles/<tag>_pmc.json + profiles/<tag>_bench_kernel_stats.csv)"""
import csv, json, os, sys
tag = sys.argv[1]; clock = float(sys.argv[2]) if len(sys.argv) > 2 else 2.3
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc = json.load(open(os.path.join(root, "profiles", tag + "_pmc.json")))
kern = pmc.get("kernels", pmc)
dur = {}
for r in csv.DictReader(open(os.path.join(root, "profiles", tag + "_bench_kernel_stats.csv"))):
    dur[r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("<true>", "").replace("<false>", "")] = float(r["AverageNs"]) / 1e3
# compute units a launch may use: the queued front-end kernels of the timed loop leave 16 per XCD to the mapping solves
cus = lambda name: 128 if name.endswith("_q") or name == "k_pyr_bands" else 256
print("%-22s %12s %10s %10s %8s" % ("kernel", "VALU instr", "floor us", "measured", "frac"))
for name, v in sorted(kern.items(), key=lambda kv: -dur.get(kv[0], 0) * 1):
    if not isinstance(v, dict) or "SQ_INSTS_VALU" not in v or name not in dur:
        continue
    n = v["SQ_INSTS_VALU"]["mean_per_launch"]
    floor_us = n / (cus(name) * clock * 1e3)
    if dur[name] >= 20:
        print("%-22s %12.0f %10.1f %10.1f %8.2f" % (name, n, floor_us, dur[name], floor_us / dur[name]))
