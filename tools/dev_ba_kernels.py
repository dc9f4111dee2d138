"""development: per-kernel HIP-event times of the config-3 window, random and contiguous tracks (optimize_profiled), and the
graph-replayed wall per iteration.  usage: dev_ba_kernels.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
for kind in ("random", "contiguous"):
    prob = synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0, tracks=kind, top_up=True)
    ba = hip.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hip.ba_obs_array(prob), prob["cam"])
    acc = {}
    for _ in range(reps):
        ba.reset()
        prof, iters, dim = ba.optimize_profiled(True, 10)
        for n, (ms, marks, per) in prof.items():
            acc[n] = acc.get(n, 0.0) + ms / iters
    ba.reset(); ba.optimize(True, 10); ba.reset(); ba.optimize(True, 10)
    tw = []
    for _ in range(8):
        ba.reset(); ba.state()
        t0 = time.perf_counter(); ba.optimize(True, 10); tw.append(time.perf_counter() - t0)
    print("%-10s %s  event sum %.1f us/iter, graph wall %.1f us/iter, chi2 last %.6f" % (kind, {n: round(1e3 * v / reps, 2) for n, v in acc.items()}, 1e3 * sum(acc.values()) / reps, 1e5 * np.median(tw), ba.optimize(True, 0) is None or 0))
    ba.close()
