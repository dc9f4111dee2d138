"""Scratch: per-kernel times of one local BA (bench shape) + wall time of the graph-replayed optimize."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
tracks = sys.argv[1] if len(sys.argv) > 1 else "random"
p = synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0, tracks=tracks, top_up=True)
b = hip.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], hip.ba_obs_array(p), p["cam"])
print(tracks, "solver", b.solver())
for _ in range(3):
    b.reset(); b.optimize(True, 10)
ts = []
for _ in range(10):
    b.reset(); ctx.sync() if hasattr(ctx, "sync") else None
    t0 = time.perf_counter(); b.optimize(True, 10); ts.append(time.perf_counter() - t0)
print("optimize(10 iterations): best %.3f ms, median %.3f ms" % (min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3))
b.reset()
k, iters, dim = b.optimize_profiled(True, 10)
print("dim", dim, "iterations", iters)
tot = 0
for name, (ms, marks, per) in k.items():
    print("  %-16s %8.2f us per iteration (%d marks, %d launches per mark)" % (name, ms * 1e3 / max(iters, 1), marks, per))
    tot += ms
print("  sum %.2f us per iteration" % (tot * 1e3 / max(iters, 1)))
