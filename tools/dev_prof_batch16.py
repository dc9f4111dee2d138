"""Scratch: one batch of 16 local-BA problems, 3 solves (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpslam_amd import hip, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
kind = sys.argv[2] if len(sys.argv) > 2 else "random"
probs = [synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=s, tracks=kind, top_up=True) for s in range(4)]
obs = [hip.ba_obs_array(p) for p in probs]
bas = [hip.BundleAdjuster(ctx, probs[i % 4]["poses"], probs[i % 4]["fixed"], probs[i % 4]["points"], obs[i % 4], probs[i % 4]["cam"]) for i in range(B)]
import time
for rep in range(4):
    hip.ba_reset_batch(bas)
    for b in bas: b.state()
    t0 = time.perf_counter()
    hip.ba_optimize_batch(bas, True, 10)
    print("batch of %d %s windows, 10 iterations: %.3f ms" % (B, kind, 1e3 * (time.perf_counter() - t0)))
