"""development: a batch of many banded (twisted) windows -- more helper / main workgroup pairs than compute units -- against single solves"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpslam_amd import hip, synth
ctx = hip.Context(320, 240, 400, 1.2, 4, max_images=1)
probs = [synth.ba_problem(34 + (i % 5), 1200, 7000, 640, 480, seq_id=200 + i % 7, tracks="contiguous", top_up=bool(i & 1)) for i in range(7)]
make = lambda pr: hip.BundleAdjuster(ctx, pr["poses"], pr["fixed"], pr["points"], hip.ba_obs_array(pr), pr["cam"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = [make(probs[i % 7]) for i in range(n)]
print(batch[0].solver(), "problems", n)
logs = hip.ba_optimize_batch(batch, True, 5)
singles = []
for i in range(7):
    b = make(probs[i]); l = b.optimize(True, 5); singles.append((l, b.state())); b.close()
bad = 0
for i, (b, lg) in enumerate(zip(batch, logs)):
    l, (p, x) = singles[i % 7]
    gp, gx = b.state()
    if lg.tobytes() != l.tobytes() or not np.array_equal(gp, p) or not np.array_equal(gx, x):
        bad += 1
print("mismatches:", bad)
for b in batch:
    b.close()
ctx.close()
