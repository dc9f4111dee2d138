"""development: phase times inside k_pose_optimize (build with LPSLAM_HIP_EXTRA_FLAGS=-DLPSLAM_PO_STAMPS)"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
ctx = hip.Context(1280, 720, 2000, 1.2, 8, max_images=2)
names = ["prologue", "accumulate+loop end", "reduce28", "decision", "solve", "oplus", "barrier", "-", "classification", "quat->R", "loads", "residual", "chi+huber", "jacobian"]
for n_obs_target in [int(a) for a in sys.argv[1:]] or (150, 300, 500):
    prob = synth.ba_problem(2, n_obs_target, 2 * n_obs_target, 1280, 720, seq_id=12)
    kf = 1
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
    buf = np.zeros(16)
    if hasattr(ctx.lib, "lpslam_hip_debug_po_stamps"):
        ctx.lib.lpslam_hip_debug_po_stamps(buf.ctypes.data_as(C.c_void_p))
    print("n_obs %4d: %.1f us per call (best %.1f), %d passes, %.2f us per pass, inliers %d, pose sum %.15f" % (len(obs), 1e6 * np.median(ts), 1e6 * min(ts), ctx.pose_optimize_passes(), 1e6 * np.median(ts) / ctx.pose_optimize_passes(), n_in, float(np.sum(pose))))
    print("   passes %d;  cycles per pass (total %.0f kcycles): " % (buf[15], buf[:14].sum() / 1e3) + ", ".join("%s %.0f" % (nm, buf[k] / buf[15]) for k, nm in enumerate(names)))
