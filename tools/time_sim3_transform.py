#!/usr/bin/env python3
"""Times lpslam_hip_sim3_transform_optimize on a batch of loop candidates (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpslam_amd import hip, synth
hip.load()
ctx = hip.Context(320, 240, 400, 1.2, 4, max_images=1)
probs = [synth.sim3_pair_problem(300, sid, init_noise=(0.02, 0.15, 0.0)) for sid in range(32)]
s0 = np.array([p["s12"] for p in probs]); pl = [hip.sim3_pairs(p) for p in probs]
for _ in range(2): hip.sim3_transform_optimize(ctx, s0, pl, probs[0]["cam1"], probs[0]["cam2"], 10.0, True)
t = time.perf_counter()
for _ in range(10): s, inl, cnt = hip.sim3_transform_optimize(ctx, s0, pl, probs[0]["cam1"], probs[0]["cam2"], 10.0, True)
dt = (time.perf_counter() - t) / 10
print("32 candidates x 300 pairs: %.3f ms per batch, inliers %s" % (1e3 * dt, cnt[:4]))
t = time.perf_counter()
for _ in range(10): hip.sim3_transform_optimize(ctx, s0[:1], pl[:1], probs[0]["cam1"], probs[0]["cam2"], 10.0, True)
print("1 candidate: %.3f ms" % (1e3 * (time.perf_counter() - t) / 10))
