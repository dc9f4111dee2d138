#!/usr/bin/env python3
"""VALU-issue roofline of the kernels of a profile, against the MEASURED issue ceiling.
tools/dev/valu_issue_bench.hip (profiles/r05_valu_issue.txt) measured what a gfx950 compute unit issues per shader-clock cycle:
1.75 wave64 instructions of the 32-bit encoded VOP1 / VOP2 kind (v_add_u32: from two wavefronts per SIMD on), 0.96 of the wide kinds
(VOP3 / VOP3P: v_perm_b32, v_pk_sub_u16, v_alignbyte_b32, v_fma_f32 / f64) -- neither the "one per cycle" round 4 assumed nor the "two
per cycle" of the guide's v_fma_f32 figure -- at a shader clock of 2.15 - 2.40 GHz under load.  A kernel's ceiling is the harmonic
mix over its instruction kinds (tools/valu_mix.py: static mix of the binary, profiles/<tag>_valu_mix.json);
floor_us = SQ_INSTS_VALU / (compute units x clock x ceiling).
usage: valu_roofline.py <profile tag> [clock GHz, default 2.3] [mix tag, default r05]   (profiles/<tag>_pmc.json + _bench_kernel_stats.csv)"""
import csv, json, os, sys
tag = sys.argv[1]; clock = float(sys.argv[2]) if len(sys.argv) > 2 else 2.3
mix_tag = sys.argv[3] if len(sys.argv) > 3 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc = json.load(open(os.path.join(root, "profiles", tag + "_pmc.json")))
mix = json.load(open(os.path.join(root, "profiles", mix_tag + "_valu_mix.json")))["kernels"]
kern = pmc.get("kernels", pmc)
dur = {}
for r in csv.DictReader(open(os.path.join(root, "profiles", tag + "_bench_kernel_stats.csv"))):
    dur[r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("<true>", "").replace("<false>", "")] = float(r["AverageNs"]) / 1e3
# compute units a launch may use: the queued front-end kernels of the timed loop leave 16 per XCD to the mapping solves
cus = lambda name: 128 if name.endswith("_q") or name == "k_pyr_bands" else 256
print("%-22s %12s %8s %10s %10s %8s %14s" % ("kernel", "VALU instr", "ceiling", "floor us", "measured", "frac", "instr/CU-cycle"))
for name, v in sorted(kern.items(), key=lambda kv: -dur.get(kv[0], 0) * 1):
    if not isinstance(v, dict) or "SQ_INSTS_VALU" not in v or name not in dur or name not in mix:
        continue
    n = v["SQ_INSTS_VALU"]["mean_per_launch"]
    ceil = mix[name]["issue_ceiling_per_cu_cycle"]
    floor_us = n / (cus(name) * clock * 1e3 * ceil)
    if dur[name] >= 20:
        print("%-22s %12.0f %8.3f %10.1f %10.1f %8.2f %14.3f" % (name, n, ceil, floor_us, dur[name], floor_us / dur[name], n / (cus(name) * clock * 1e3 * dur[name])))
