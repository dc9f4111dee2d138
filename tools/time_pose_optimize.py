#!/usr/bin/env python3
"""Times lpslam_hip_pose_optimize (one-workgroup motion-only BA) for a few observation counts (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpslam_amd import hip, synth
hip.LIB_PATH = os.environ.get("LPSLAM_LIB", hip.LIB_PATH); hip.load()
ctx = hip.Context(320, 240, 400, 1.2, 4, max_images=1)
for n in (30, 60, 100, 120, 180, 200, 250, 300, 500, 1000, 2000):
    prob = synth.ba_problem(2, n, 2 * n, 640, 480, seq_id=n)
    sel = prob["obs_pose"] == 1
    obs = hip.ba_obs_array(prob)[sel]
    pose = prob["poses"][1].copy()
    for _ in range(2): hip.pose_optimize(ctx, pose, prob["points"], obs, prob["cam"])
    t = time.perf_counter()
    for _ in range(20): p7, out, inl = hip.pose_optimize(ctx, pose, prob["points"], obs, prob["cam"])
    dt = (time.perf_counter() - t) / 20
    print("%5d observations: %.3f ms per call, %d passes, %.2f us per pass, %d inliers" % (len(obs), 1e3 * dt, ctx.pose_optimize_passes(), 1e6 * dt / ctx.pose_optimize_passes(), inl))
