#!/usr/bin/env python3
"""Static vector-instruction mix of the front-end kernels by ENCODING class, from the compiler's gfx950 assembly of frontend.hip:
`e32` = 32-bit encoded VOP1 / VOP2 / VOPC forms (v_add_u32_e32, v_mov_b32_e32, ...), `wide` = everything else (VOP3, VOP3P packed,
SDWA / DPP forms, 64-bit operations).  tools/dev/valu_issue_bench.hip measured the two classes' issue rates on an MI355X
(profiles/r05_valu_issue.txt): 1.75 and 0.96 wave64 instructions per compute unit and cycle.  A kernel's issue ceiling is the
harmonic mix  1 / (f_e32 / 1.75 + f_wide / 0.96).  The mix is static (one count per instruction in the binary, loops not weighted):
a proxy, said so where it is used.  Writes profiles/<tag>_valu_mix.json.   usage: valu_mix.py <tag>"""
import json, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
RATE = {"e32": 1.75, "wide": 0.96}      # profiles/r05_valu_issue.txt: v_add_u32 at >= 2 wavefronts per SIMD; v_perm / v_pk_sub_u16 / v_alignbyte / v_fma (VOP3) at 4
with tempfile.TemporaryDirectory() as td:
    s_path = os.path.join(td, "fe.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", s_path,
                           os.path.join(root, "lpslam_amd", "csrc", "frontend.hip")], stderr=subprocess.DEVNULL)
    lines = open(s_path).read().split("\n")
cur, stats = None, {}
for ln in lines:
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
    if m:
        cur = m.group(1); stats[cur] = {"e32": 0, "wide": 0}; continue
    if cur is None:
        continue
    if ln.startswith(".Lfunc_end"):
        cur = None; continue
    t = ln.strip()
    if not t.startswith("v_"):
        continue
    op = t.split()[0]
    if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
        continue                                          # (lane moves: not counted by SQ_INSTS_VALU's vector-ALU work either way; left out)
    stats[cur]["e32" if op.endswith("_e32") and "f64" not in op else "wide"] += 1
out = {"rates_per_cu_cycle": RATE, "source": "profiles/r05_valu_issue.txt (tools/dev/valu_issue_bench.hip)", "kernels": {}}
names = {"k_fast_cells": "12k_fast_cells", "k_fast_cells_q": "14k_fast_cells_q", "k_describe": "10k_describe", "k_describe_q": "12k_describe_q",
         "k_pyr_bands": "11k_pyr_bandsILb1", "k_distribute": "12k_distribute", "k_distribute_q": "14k_distribute_q"}
for name, key in names.items():
    for k, v in stats.items():
        if key in k:
            n = v["e32"] + v["wide"]
            f = v["e32"] / n
            out["kernels"][name] = {"vector_instructions_static": n, "e32_fraction": round(f, 3), "issue_ceiling_per_cu_cycle": round(1.0 / (f / RATE["e32"] + (1 - f) / RATE["wide"]), 3)}
            break
json.dump(out, open(os.path.join(root, "profiles", tag + "_valu_mix.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
