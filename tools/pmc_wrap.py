#!/usr/bin/env python3
"""Wraps gpurun_out/<tag>_pmc_kernels.json (tools/pmc_summary.py) into profiles/<tag>_pmc.json with the command, the units and the
workload the counters belong to, and copies the bench line and kernel statistics of the same tag.  usage: pmc_wrap.py <tag>"""
import json, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
k = json.load(open(os.path.join(src, tag + "_pmc_kernels.json")))
k = {name.replace("void ", "").replace("<true>", "").replace("<false>", ""): v for name, v in k.items()}
# MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced read (16 B per lane); other
# access widths are uncalibrated.  Kernels whose global reads are 16-byte per lane get fetch_correction 2, the others stay as counted.
WIDE16 = {"k_chol_pair", "k_chol_wg", "k_ba_schur", "k_copy_from_host", "k_bs_copy_in", "k_ba_lin", "k_ba_trial", "k_ba_update", "k_distribute", "k_schur_group"}
for name, v in k.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        corr = 2.0 if name in WIDE16 else 1.0
        v["hbm_bytes_per_launch"] = {"raw": int(round(1024 * (v["FETCH_SIZE"]["mean_per_launch"] + v["WRITE_SIZE"]["mean_per_launch"]))),
                                     "corrected": int(round(1024 * (corr * v["FETCH_SIZE"]["mean_per_launch"] + v["WRITE_SIZE"]["mean_per_launch"]))),
                                     "fetch_correction": corr, "read_class": "16 B per lane (x2, guide)" if corr == 2.0 else "narrower than 16 B per lane (uncalibrated: as counted)"}
bench = json.loads(open(os.path.join(src, tag + "_bench.json")).read().strip().splitlines()[-1])
out = {
    "command": "rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 bench.py --no-cpu --no-extras --steps 4 --warmup 1; four separate passes: "
               "{SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES}, {FETCH_SIZE}, {WRITE_SIZE}, "
               "{SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAVES}; condensed by tools/pmc_summary.py "
               "(values summed over XCDs, mean per launch); script: tools/refresh_profiles.sh",
    "units": "FETCH_SIZE / WRITE_SIZE in KB as reported by rocprofv3; hbm_bytes_per_launch.corrected applies the guide's gfx950 rule (FETCH_SIZE x 2 for "
             "reads of 16 B per lane; narrower reads are uncalibrated and kept as counted); SQ_* in the counter's own units",
    "frames_per_launch": bench["config"]["frames_per_launch"],
    "workload": "%d images 1280x720 per front-end launch (%d stereo frames, 8 levels); BA 50 KF / 5000 landmarks / 40 000 observations, a fresh problem per keyframe; the timed steps once with random tracks (k_ba_schur / k_chol_pair / k_chol_xsolve) and once with contiguous tracks (k_schur_group / k_schur_band_reduce / k_chol_band); both behind k_ba_update"
                % (2 * bench["config"]["frames_per_launch"], bench["config"]["frames_per_launch"]),
    "kernels": k,
}
json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)
for suffix in ("_bench_kernel_stats.csv", "_bench_kernel_stats.txt"):
    shutil.copy(os.path.join(src, tag + suffix), os.path.join(dst, tag + suffix))
print("profiles/%s_* written" % tag)
