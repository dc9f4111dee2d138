"""Scratch: wall time of lpslam_hip_pose_optimize (one launch: 4 x 10 LM iterations) for tracker-sized inputs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
for n_obs_target in (300, 600, 1200):
    prob = synth.ba_problem(6, 2000, 6 * n_obs_target, 1280, 720, seq_id=12)
    kf = 4
    sel = prob["obs_pose"] == kf
    obs = hip.ba_obs_array(prob)[sel].copy()
    obs["pose"] = 0
    bad = np.arange(0, len(obs), 9)
    obs["v"][bad] += 25.0
    pts = prob["points_gt"] + np.random.default_rng(5).normal(0, 0.01, prob["points_gt"].shape)
    start = prob["poses"][kf]
    for _ in range(3):
        pose, out, n_in = hip.pose_optimize(ctx, start, pts, obs, prob["cam"])
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); hip.pose_optimize(ctx, start, pts, obs, prob["cam"]); ts.append(time.perf_counter() - t0)
    print("n_obs %4d: %.1f us per call (best %.1f), inliers %d, pose sum %.15f" % (len(obs), 1e6 * np.median(ts), 1e6 * min(ts), n_in, float(np.sum(pose))))
