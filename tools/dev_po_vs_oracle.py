"""development: deviation of lpslam_hip_pose_optimize from the oracle's pose_optimize on tracker-sized problems (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
from oracle import oracle as oba
oba.build()
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
worst = 0.0
for seq, n in ((12, 150), (12, 300), (12, 500), (3, 100), (4, 200), (5, 400), (6, 800), (7, 1500)):
    prob = synth.ba_problem(2, n, 2 * n, 1280, 720, seq_id=seq)
    sel = prob["obs_pose"] == 1
    hobs = hip.ba_obs_array(prob)[sel].copy(); hobs["pose"] = 0
    bad = np.arange(0, len(hobs), 9); hobs["v"][bad] += 25.0
    pts = prob["points_gt"] + np.random.default_rng(5).normal(0, 0.01, prob["points_gt"].shape)
    start = prob["poses"][1]
    oobs = np.zeros(len(hobs), oba.OBS_DTYPE)
    for f in oobs.dtype.names: oobs[f] = hobs[f]
    opose, oout, oin = oba.pose_optimize(start, pts, oobs, prob["cam"])
    kpose, kout, kin = hip.pose_optimize(ctx, start, pts, hobs, prob["cam"])
    dq = np.abs(kpose[:4] - opose[:4]).max(); dt = np.abs(kpose[4:] - opose[4:]).max()
    worst = max(worst, dq, dt)
    print("n %4d: inliers %d / %d, outlier masks equal %s, |dq| %.2e, |dt| %.2e, passes %d" % (len(hobs), kin, oin, np.array_equal(kout, oout.astype(bool)), dq, dt, ctx.pose_optimize_passes()))
print("worst %.2e" % worst)
