#!/usr/bin/env python3
"""Fuzz of lpslam_hip_pose_optimize against the oracle's pose_optimize (GPU box): random observation counts (0 ... beyond what the
kernel keeps in LDS), mono / stereo mixes, gross outliers, start poses off the truth.  usage: fuzz_pose.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
from oracle import oracle as O
O.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = hip.Context(640, 480, 500, 1.2, 4, max_images=2)
bad = 0; worst_r = worst_t = 0.0
t0 = time.time()
sizes = [0, 1, 4, 5, 6, 30, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 2600, 2700, 2800, 3500]
for case in range(cases):
    n = int(sizes[case % len(sizes)] if case < 2 * len(sizes) else rng.integers(0, 1500))
    cam = dict(synth.intrinsics(640, 480)); cam_stereo = rng.random() < 0.7
    if not cam_stereo: cam["fxb"] = 0.0
    pts = np.stack([rng.uniform(-4, 4, max(n, 1)), rng.uniform(-3, 3, max(n, 1)), rng.uniform(2, 20, max(n, 1))], axis=1)
    yaw = rng.normal(0, 0.02); q = np.array([np.cos(yaw / 2), 0, np.sin(yaw / 2), 0]); t = rng.normal(0, 0.05, 3)
    R = np.array([[1 - 2 * q[2] ** 2, 0, 2 * q[0] * q[2]], [0, 1, 0], [-2 * q[0] * q[2], 0, 1 - 2 * q[2] ** 2]])
    pc = pts @ R.T + t
    obs = np.zeros(n, hip.BA_OBS_DTYPE)
    obs["point"] = np.arange(n)
    obs["u"] = cam["fx"] * pc[:n, 0] / pc[:n, 2] + cam["cx"] + rng.normal(0, 0.5, n)
    obs["v"] = cam["fy"] * pc[:n, 1] / pc[:n, 2] + cam["cy"] + rng.normal(0, 0.5, n)
    stereo = (rng.random(n) < 0.6) & cam_stereo
    obs["ur"] = np.where(stereo, obs["u"] - cam["fxb"] / pc[:n, 2] + rng.normal(0, 0.5, n), -1.0)
    obs["inv_sigma2"] = 1.0 / (1.2 ** (2 * rng.integers(0, 4, n)))
    out_idx = rng.random(n) < rng.choice([0.0, 0.1, 0.3])
    obs["v"][out_idx] += rng.choice([-1, 1], out_idx.sum()) * rng.uniform(15, 60, out_idx.sum())
    start = np.concatenate([[1, 0, 0, 0], rng.normal(0, 0.02, 3)])
    oobs = np.zeros(n, O.OBS_DTYPE)
    for f in oobs.dtype.names: oobs[f] = obs[f]
    opose, oout, oin = O.pose_optimize(start, pts, oobs, cam)
    kpose, kout, kin = hip.pose_optimize(ctx, start, pts, obs, cam)
    dr = float(np.abs(kpose[:4] - opose[:4]).max()); dt = float(np.abs(kpose[4:] - opose[4:]).max())
    same = kin == oin and np.array_equal(kout, oout.astype(bool))
    # a classification can sit on its threshold: tolerate a handful of flipped observations when the poses agree
    flips = int(np.sum(kout != oout.astype(bool)))
    ok = (same or flips <= max(1, n // 200)) and dr < 1e-6 and dt < 1e-5
    worst_r = max(worst_r, dr); worst_t = max(worst_t, dt)
    if not ok:
        bad += 1
        print("BAD  case %d: n %d stereo %s inliers %d / %d flips %d |dq| %.2e |dt| %.2e passes %d" % (case, n, cam_stereo, kin, oin, flips, dr, dt, ctx.pose_optimize_passes()))
    elif case % 20 == 0:
        print("ok   case %d: n %d stereo %s inliers %d flips %d |dq| %.1e |dt| %.1e passes %d" % (case, n, cam_stereo, kin, flips, dr, dt, ctx.pose_optimize_passes()))
print("%d cases, %d bad, worst |dq| %.2e |dt| %.2e, %.1f s" % (cases, bad, worst_r, worst_t, time.time() - t0))
sys.exit(1 if bad else 0)
