"""Scratch: batch vs single BA, and object-vs-object determinism."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
ctx = hip.Context(320, 240, 400, 1.2, 4, max_images=1)
specs = [(12, 600, 4000, 21), (5, 80, 320, 22), (33, 900, 6000, 23), (6, 150, 800, 6), (2, 20, 40, 25)]
probs = []
for kf, pts, obs, seq in specs:
    kw = dict(pose_noise=(0.08, 0.5), point_noise=0.5) if seq == 6 else {}
    probs.append(synth.ba_problem(kf, pts, obs, 640, 480, seq_id=seq, **kw))
make = lambda pr: hip.BundleAdjuster(ctx, pr["poses"], pr["fixed"], pr["points"], hip.ba_obs_array(pr), pr["cam"])
a = [make(p) for p in probs]; la = [b.optimize(True, 8) for b in a]
b2 = [make(p) for p in probs]; lb = [b.optimize(True, 8) for b in b2]
for i, (x, y) in enumerate(zip(la, lb)):
    print("single vs single", i, x.tobytes() == y.tobytes(), np.abs(x["chi2_after"] / y["chi2_after"] - 1).max())
c = [make(p) for p in probs]; lc = hip.ba_optimize_batch(c, True, 8)
for i, (x, y) in enumerate(zip(la, lc)):
    print("single vs batch ", i, x.tobytes() == y.tobytes(), len(x), len(y), np.abs(x["chi2_after"][:len(y)] / y["chi2_after"][:len(x)] - 1).max(), x["trials"], y["trials"])
for n in (1, 2):
    d = [make(p) for p in probs[:n]]; ld = hip.ba_optimize_batch(d, True, 8)
    for i, (x, y) in enumerate(zip(la, ld)):
        print("batch of", n, i, x.tobytes() == y.tobytes())
