"""Scratch: two frames of the ring scene 3 degrees apart: BF matches + pose optimiser through the HIP API."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpslam_amd import hip, synth
w, h = 640, 480
k = synth.intrinsics(w, h)
seq = synth.StereoSequence(w, h, 4, n_points=9000)
rng = np.random.default_rng(21)
az = rng.uniform(0, 2 * np.pi, 9000); rad = rng.uniform(5.0, 25.0, 9000)
seq.pts = np.stack([rad * np.sin(az), rng.uniform(-4, 4, 9000), rad * np.cos(az)], axis=1)
def frame(i, deg=3.0):
    yaw = math.radians(deg) * i
    c, s_ = math.cos(yaw), math.sin(yaw)
    R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]).T
    nr = np.random.Generator(np.random.PCG64([5, i]))
    return seq._render(R, np.zeros(3), nr), seq._render(R, -np.array([k["baseline"], 0.0, 0.0]), nr), R
ctx = hip.Context(w, h, 1000, 1.2, 4, max_images=4)
l0, r0, R0 = frame(0); l1, r1, R1 = frame(1)
for i, im in enumerate((l0, r0, l1, r1)):
    ctx.upload(i, im)
ctx.extract(4)
ctx.match_stereo_strided(0, 1, 2, 2, k["fxb"], k["baseline"])
kp0, d0 = ctx.keypoints(0); kp1, d1 = ctx.keypoints(2)
xr0, dep0, _ = ctx.stereo(0); xr1, dep1, _ = ctx.stereo(2)
print("kpts", len(kp0), len(kp1), "depths", (dep0 > 0).sum(), (dep1 > 0).sum())
ctx.match_bf(2, 0)
mq, mt, md = ctx.bf_matches(2, 0, 50, 0.9, True)
print("bf matches", len(mq))
good = dep0[mt] > 0
mq, mt = mq[good], mt[good]
# landmarks = frame-0 keypoints back-projected (camera 0 = world)
z = dep0[mt]; X = np.stack([(kp0["x"][mt] - k["cx"]) * z / k["fx"], (kp0["y"][mt] - k["cy"]) * z / k["fy"], z], axis=1).astype(np.float64)
obs = np.zeros(len(mq), hip.BA_OBS_DTYPE)
obs["point"] = np.arange(len(mq)); obs["u"] = kp1["x"][mq]; obs["v"] = kp1["y"][mq]; obs["ur"] = np.where(xr1[mq] >= 0, xr1[mq], -1.0)
obs["inv_sigma2"] = 1.0 / (np.float32(1.2) ** kp1["octave"][mq]).astype(np.float64) ** 2
pose, out, inl = hip.pose_optimize(ctx, np.array([1.0, 0, 0, 0, 0, 0, 0]), X, obs, k)
print("pose", pose, "inliers", inl, "of", len(obs))
print("expected yaw quaternion y component", math.sin(math.radians(3.0) / 2), "(world->camera: negative)")
