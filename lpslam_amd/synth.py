"""Seeded synthetic stereo sequences and bundle-adjustment problems.

Host-side test/bench input generation only (numpy); nothing here is on the measured path.
PRNG: numpy PCG64 seeded with 0x5EED0000 + sequence_id.
"""
import math
import numpy as np

SEED_BASE = 0x5EED0000

INTRINSICS = {
    (640, 480): dict(fx=525.0, fy=525.0, cx=320.0, cy=240.0),
    (1280, 720): dict(fx=700.0, fy=700.0, cx=640.0, cy=360.0),
    (1920, 1080): dict(fx=1050.0, fy=1050.0, cx=960.0, cy=540.0),
}
BASELINE_M = 0.12  # cf. /root/reference/src/Sources/OpenCVCameraSource.cpp:68-75


def intrinsics(width, height):
    if (width, height) in INTRINSICS:
        k = dict(INTRINSICS[(width, height)])
    else:  # small test sizes: same field of view as 640x480
        f = 525.0 * width / 640.0
        k = dict(fx=f, fy=f, cx=width / 2.0, cy=height / 2.0)
    k["fxb"] = k["fx"] * BASELINE_M
    k["baseline"] = BASELINE_M
    return k


def _value_noise(rng, h, w, octaves=3, mean=110.0, sigma=25.0):
    """Three octaves of bilinearly interpolated random lattices."""
    out = np.zeros((h, w), np.float64)
    amp, total = 1.0, 0.0
    for o in range(octaves):
        cell = 64 >> o
        gh, gw = h // cell + 2, w // cell + 2
        g = rng.standard_normal((gh, gw))
        ys = np.arange(h) / cell
        xs = np.arange(w) / cell
        y0 = ys.astype(int); x0 = xs.astype(int)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        out += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
        total += amp * amp
        amp *= 0.5
    out *= sigma / math.sqrt(total) / 0.6
    return out + mean


class StereoSequence:
    """Rectified, distortion-free synthetic stereo sequence."""

    def __init__(self, width, height, seq_id=0, n_points=20000):
        self.w, self.h = int(width), int(height)
        self.k = intrinsics(width, height)
        self.rng = np.random.Generator(np.random.PCG64(SEED_BASE + int(seq_id)))
        r = self.rng
        self.pts = np.stack([r.uniform(-20, 20, n_points), r.uniform(-5, 5, n_points),
                             r.uniform(1.0, 41.0, n_points)], axis=1)
        self.amp = r.uniform(60, 120, n_points)
        self.quad = r.integers(0, 2, (n_points, 2, 2)) * 2 - 1     # +-1 per quadrant
        same = (self.quad.reshape(n_points, 4) == self.quad[:, :1, 0]).all(axis=1)
        self.quad[same, 0, 0] *= -1                                 # never a flat patch
        self.bg = _value_noise(r, self.h, self.w)

    def pose(self, k):
        """World->camera (R, t) of the left camera at frame k: +z at 0.05 m/frame, small yaw."""
        yaw = math.radians(0.2) * math.sin(2 * math.pi * k / 100.0)
        c, s = math.cos(yaw), math.sin(yaw)
        R_wc = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])        # camera->world
        C = np.array([0.0, 0.0, 0.05 * k])
        R = R_wc.T
        return R, -R @ C

    def _render(self, R, t, noise_rng):
        img = self.bg.copy()
        pc = self.pts @ R.T + t
        z = pc[:, 2]
        ok = z > 0.5
        u = np.where(ok, self.k["fx"] * pc[:, 0] / np.where(ok, z, 1) + self.k["cx"], -100)
        v = np.where(ok, self.k["fy"] * pc[:, 1] / np.where(ok, z, 1) + self.k["cy"], -100)
        ui = np.rint(u).astype(int); vi = np.rint(v).astype(int)
        ok &= (ui >= 4) & (ui < self.w - 4) & (vi >= 4) & (vi < self.h - 4)
        idx = np.nonzero(ok)[0]
        idx = idx[np.argsort(-z[idx], kind="stable")]              # far first, near overwrites
        flat = img.reshape(-1)
        for dy in range(-3, 4):
            for dx in range(-3, 4):
                q = self.quad[idx, int(dy >= 0), int(dx >= 0)]
                flat[(vi[idx] + dy) * self.w + ui[idx] + dx] = 110.0 + q * self.amp[idx]
        img += noise_rng.normal(0.0, 2.0, img.shape)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def frame(self, k):
        """Returns (left, right) uint8 images of frame k (deterministic per (seq_id, k))."""
        R, t = self.pose(k)
        nrng = np.random.Generator(np.random.PCG64([SEED_BASE, 77, int(k)]))
        left = self._render(R, t, nrng)
        right = self._render(R, t - np.array([self.k["baseline"], 0.0, 0.0]), nrng)
        return left, right


def turning_sequence(width, height, n_frames=156, step_deg=3.0, seq_id=4, n_points=9000, radius=(5.0, 25.0)):
    """A turn on the spot, once around and 108 degrees more (the loop detector wants its candidate at four keyframes in a row), inside a ring of structure (loop-closure tests, tests/golden/g14_track_loop.npz): stereo frames of a
    camera at the origin turning by `step_deg` per frame about its y axis; the background is a panorama fixed to the WORLD (the
    generator's own background is fixed to the image).  Returns (frames, yaws)."""
    k = intrinsics(width, height)
    seq = StereoSequence(width, height, seq_id, n_points=n_points)
    rng = np.random.default_rng(21)
    az = rng.uniform(0, 2 * np.pi, n_points); rad = rng.uniform(radius[0], radius[1], n_points)
    seq.pts = np.stack([rad * np.sin(az), rng.uniform(-4, 4, n_points), rad * np.cos(az)], axis=1)       # structure all around the camera
    pano = _value_noise(np.random.Generator(np.random.PCG64(77)), 1200, 7200)                            # 0.05 degrees per pixel
    uu, vv = np.meshgrid((np.arange(width) - k["cx"]) / k["fx"], (np.arange(height) - k["cy"]) / k["fy"])

    def world_background(R):
        d = np.stack([uu, vv, np.ones_like(uu)], axis=-1) @ R                  # camera ray -> world (R is world -> camera)
        a = (np.arctan2(d[..., 0], d[..., 2]) + np.pi) * (7200 / (2 * np.pi))
        e = (np.arctan2(d[..., 1], np.hypot(d[..., 0], d[..., 2])) + np.pi / 6) * (1200 / (np.pi / 3))
        a0 = np.floor(a).astype(int); e0 = np.clip(np.floor(e).astype(int), 0, 1198)
        fa = a - a0; fe = np.clip(e - e0, 0, 1)
        a0 %= 7200; a1 = (a0 + 1) % 7200
        return (pano[e0, a0] * (1 - fa) + pano[e0, a1] * fa) * (1 - fe) + (pano[e0 + 1, a0] * (1 - fa) + pano[e0 + 1, a1] * fa) * fe
    step = math.radians(step_deg)
    frames, yaws = [], []
    for i in range(n_frames):
        yaw = step * i
        c, s_ = math.cos(yaw), math.sin(yaw)
        R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]).T          # world -> camera for a camera turned by `yaw` about y
        nr = np.random.Generator(np.random.PCG64([5, i]))
        seq.bg = world_background(R)
        frames.append((seq._render(R, np.zeros(3), nr), seq._render(R, -np.array([k["baseline"], 0.0, 0.0]), nr)))
        yaws.append(yaw)
    return frames, yaws


class WallSequence:
    """Monocular test scene: three fronto-parallel textured walls at 14, 9 and 6 m, one per horizontal band of the image, seen by a
    camera that moves sideways by `step` metres per frame.  Every wall slides by a whole number of pixels (f * x / depth, rounded),
    so descriptors do not change between frames, while the three depths give the parallax a two-view initialisation needs.

    rectangle = (a, b): the camera goes a frames to the right, b frames up, a frames to the left and b frames down -- never turning --
    and is back where it started at frame 2 (a + b): a real loop for the monocular loop-closure tests.  The walls then slide
    vertically as well, behind the same three bands of the image."""

    def __init__(self, width, height, seq_id=0, step=0.1, depths=(14.0, 9.0, 6.0), margin=None, rectangle=None):
        self.w, self.h, self.step, self.depths = int(width), int(height), float(step), tuple(depths)
        self.k = intrinsics(width, height)
        rng = np.random.default_rng(SEED_BASE + 7919 * int(seq_id) + 3)
        self.band_h = self.h // len(self.depths)
        self.margin = int(margin if margin is not None else 0.75 * self.w)
        self.rectangle = rectangle
        self.walls = []
        for z in self.depths:
            extra_x, extra_y = self.margin, 0
            if rectangle:
                extra_x = int(round(self.k["fx"] * self.step * rectangle[0] / z)) + 1
                extra_y = int(round(self.k["fy"] * self.step * rectangle[1] / z)) + 1
            t = rng.integers(0, 256, (self.band_h + extra_y, self.w + extra_x)).astype(np.float64)
            t = (t + np.roll(t, 1, 0) + np.roll(t, 1, 1) + np.roll(t, (1, 1), (0, 1))) / 4.0        # 2x2 box: corners FAST still likes
            self.walls.append(60 + (t - t.min()) * (150.0 / (t.max() - t.min())))

    def centre(self, i):
        if not self.rectangle:
            return np.array([self.step * i, 0.0, 0.0])
        a, b = self.rectangle
        i = int(i) % (2 * (a + b))
        if i <= a:
            return np.array([self.step * i, 0.0, 0.0])
        if i <= a + b:
            return np.array([self.step * a, -self.step * (i - a), 0.0])                  # image y points down: "up" is -y
        if i <= 2 * a + b:
            return np.array([self.step * (2 * a + b - i), -self.step * b, 0.0])
        return np.array([0.0, -self.step * (2 * (a + b) - i), 0.0])

    def frame(self, i):
        img = np.zeros((self.h, self.w))
        c = self.centre(i)
        for b, (z, wall) in enumerate(zip(self.depths, self.walls)):
            if self.rectangle:
                sx = int(round(self.k["fx"] * c[0] / z))                 # the wall moves left as the camera moves right
            else:
                sx = min(int(round(self.k["fx"] * self.step * i / z)), self.margin)
            sy = wall.shape[0] - self.band_h - int(round(self.k["fy"] * -c[1] / z))       # ... and down as the camera moves up
            img[b * self.band_h:(b + 1) * self.band_h] = wall[sy:sy + self.band_h, sx:sx + self.w]
        img[len(self.depths) * self.band_h:] = 110.0
        img += np.random.Generator(np.random.PCG64([SEED_BASE, 91, int(i)])).normal(0, 1.0, img.shape)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def random_image(width, height, seed=0):
    """Left image of frame 0 of a small sequence (unit tests)."""
    n = max(200, int(20000 * (width * height) / (1280.0 * 720.0)))
    return StereoSequence(width, height, seq_id=seed, n_points=n).frame(0)[0]


def rot_to_quat(R):
    tr = np.trace(R)
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0, 0.0, 0.0, 0.0]
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


def _small_rot(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def ba_problem(n_kf=50, n_points=5000, n_obs=40000, width=1280, height=720, seq_id=0,
               kf_stride=6, pose_noise=(0.01, 0.05), point_noise=0.05, pix_noise=1.0, tracks="random", top_up=False):
    """Bundle-adjustment problem of SURVEY.md section 8(d) config 3 / 5.

    Returns dict with ground-truth and perturbed poses (n_kf x 7: qw qx qy qz tx ty tz, world->camera),
    points (n x 3), observations (pose, point, u, v, ur, inv_sigma2), `fixed` flags (first KF fixed)
    and the camera dict.  Stereo observations throughout; octave drawn to give sigma^2 = 1.2^(2*level).

    `tracks`: which of the keyframes that see a landmark keep their observation when there are more than the cap
    (ceil(n_obs / n_points)).  "random": a random subset -- every landmark couples keyframes all over the window and the reduced
    system is dense (the stress case, and the round-1..3 workload).  "contiguous": a run of consecutive visible keyframes starting
    at a random one -- what a tracker produces (a landmark is followed from its first sighting until it is lost), giving a
    block-banded reduced system ("track length capped"; the reference solves it with CSparse, row a21).
    `top_up`: tracks of cap + 1 keyframes for as many landmarks as it takes to reach n_obs (the cap alone ends ~3 % short of
    config 3's 40 000 +- 2 %, because landmarks near the window's ends are seen by fewer keyframes than the cap).
    """
    rng = np.random.Generator(np.random.PCG64([SEED_BASE + int(seq_id), 3]))
    k = intrinsics(width, height)
    seq = StereoSequence(width, height, seq_id, n_points=1)
    Rs, ts = zip(*[seq.pose(i * kf_stride) for i in range(n_kf)])
    depth_span = 0.05 * kf_stride * n_kf
    cand = np.stack([rng.uniform(-20, 20, n_points * 6), rng.uniform(-5, 5, n_points * 6),
                     rng.uniform(1.0, 41.0 + depth_span, n_points * 6)], axis=1)
    vis = np.zeros((len(cand), n_kf), bool)
    uvs = np.zeros((len(cand), n_kf, 3))
    for i in range(n_kf):
        pc = cand @ Rs[i].T + ts[i]
        z = pc[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = k["fx"] * pc[:, 0] / z + k["cx"]
            v = k["fy"] * pc[:, 1] / z + k["cy"]
        vis[:, i] = (z > 1.0) & (z < 40.0) & (u >= 20) & (u < width - 20) & (v >= 20) & (v < height - 20)
        uvs[:, i, 0] = u; uvs[:, i, 1] = v; uvs[:, i, 2] = u - k["fxb"] / np.where(z > 0, z, 1)
    min_views = min(3, n_kf)
    good = np.nonzero(vis.sum(1) >= min_views)[0]
    if len(good) < n_points:
        raise RuntimeError("not enough visible landmarks: %d" % len(good))
    good = good[:n_points]
    if tracks not in ("random", "contiguous"):
        raise ValueError("tracks must be 'random' or 'contiguous'")
    cap = max(min_views, int(math.ceil(n_obs / n_points)))
    picked = []
    for pi, ci in enumerate(good):
        kfs = np.nonzero(vis[ci])[0]
        if len(kfs) > cap:
            if tracks == "random":
                kfs_sel = np.sort(rng.choice(kfs, cap, replace=False))
            else:
                start = int(rng.integers(0, len(kfs) - cap + 1))
                kfs_sel = kfs[start:start + cap]
        else:
            kfs_sel = kfs
        picked.append((kfs, kfs_sel))
    if top_up:
        # its own generator: the draws above and below are those of the problem without the top-up
        rng2 = np.random.Generator(np.random.PCG64([SEED_BASE + int(seq_id), 5]))
        missing = n_obs - sum(len(sel) for _, sel in picked)
        for pi in rng2.permutation(len(picked)):
            if missing <= 0:
                break
            kfs, sel = picked[pi]
            if len(sel) < cap or len(kfs) <= len(sel):
                continue
            if tracks == "random":
                extra = rng2.choice(np.setdiff1d(kfs, sel))
            else:                       # the run grows by one keyframe, at the end where there is one
                at = int(np.searchsorted(kfs, sel[-1]))
                extra = kfs[at + 1] if at + 1 < len(kfs) else kfs[int(np.searchsorted(kfs, sel[0])) - 1]
            picked[pi] = (kfs, np.sort(np.append(sel, extra)))
            missing -= 1
    obs = np.array([(f, pi, good[pi]) for pi, (_, sel) in enumerate(picked) for f in sel])
    if len(obs) > n_obs:     # trim observations of the longest tracks but keep >= 3 per landmark
        counts = np.bincount(obs[:, 1], minlength=n_points)
        order = rng.permutation(len(obs))
        keep = np.ones(len(obs), bool)
        excess = len(obs) - n_obs
        for j in order:
            if excess == 0:
                break
            p = obs[j, 1]
            if counts[p] > min_views:
                counts[p] -= 1; keep[j] = False; excess -= 1
        obs = obs[keep]
    obs = obs[np.lexsort((obs[:, 0], obs[:, 1]))]      # by landmark, then by keyframe
    level = rng.integers(0, 8, len(obs))
    sigma = 1.2 ** level
    meas = uvs[obs[:, 2], obs[:, 0]] + rng.normal(0, pix_noise, (len(obs), 3)) * sigma[:, None]
    poses_gt = np.array([np.concatenate([rot_to_quat(Rs[i]), ts[i]]) for i in range(n_kf)])
    poses = poses_gt.copy()
    for i in range(1, n_kf):
        dR = _small_rot(rng.normal(0, pose_noise[0], 3))
        R = dR @ Rs[i]
        t = dR @ ts[i] + rng.normal(0, pose_noise[1], 3)
        poses[i] = np.concatenate([rot_to_quat(R), t])
    pts_gt = cand[good]
    pts = pts_gt + rng.normal(0, point_noise, pts_gt.shape)
    fixed = np.zeros(n_kf, np.uint8); fixed[0] = 1
    return dict(poses_gt=poses_gt, poses=poses, points_gt=pts_gt, points=pts, fixed=fixed,
                obs_pose=obs[:, 0].astype(np.int32), obs_point=obs[:, 1].astype(np.int32),
                obs_uvr=meas.astype(np.float64), obs_inv_sigma2=(1.0 / sigma ** 2).astype(np.float64),
                cam=dict(fx=k["fx"], fy=k["fy"], cx=k["cx"], cy=k["cy"], fxb=k["fxb"]))


def pose_graph_problem(n_kf=200, seq_id=0, radius=20.0, drift_rot=0.002, drift_trans=0.02, drift_scale=0.0,
                       meas_noise=1e-3, covis=3, n_loop=5):
    """Sim3 essential graph after a loop closure (BASELINE config 5's keyframe count): keyframes on a closed circuit,
    vertex estimates from drifting odometry (random walk in rotation / translation / log-scale), edges = spanning
    chain + `covis` covisibility neighbours + `n_loop` loop edges joining the end of the circuit to its start, with
    measurements S_j S_i^-1 taken from the ground truth plus a small perturbation.  Vertices are world->camera Sim3
    as (qw, qx, qy, qz, tx, ty, tz, s); keyframe 0 is fixed (the loop keyframe of [UPSTREAM] graph_optimizer)."""
    rng = np.random.default_rng(0x5EED0000 + 7919 * seq_id + 13)

    def compose(a, b):      # (R, t, s) * (R, t, s)
        return a[0] @ b[0], a[2] * (a[0] @ b[1]) + a[1], a[2] * b[2]

    def inverse(a):
        Ri = a[0].T
        return Ri, Ri @ (-a[1] / a[2]), 1.0 / a[2]

    gt = []
    for i in range(n_kf):
        ang = 2 * math.pi * i / n_kf
        c = np.array([radius * math.sin(ang), 0.3 * math.sin(3 * ang), radius * (1 - math.cos(ang))])   # camera centre
        R_wc = _small_rot(np.array([0.0, ang, 0.0]))        # heading along the circuit
        R = R_wc.T
        gt.append((R, -R @ c, 1.0))
    est = [gt[0]]
    for i in range(1, n_kf):
        rel = compose(gt[i], inverse(gt[i - 1]))              # S_i S_{i-1}^-1
        d = (_small_rot(rng.normal(0, drift_rot, 3)), rng.normal(0, drift_trans, 3), math.exp(rng.normal(0, drift_scale)))
        est.append(compose(compose(d, rel), est[i - 1]))
    ei, ej = [], []
    for i in range(n_kf):
        for d in range(1, covis + 1):
            if i + d < n_kf:
                ei.append(i); ej.append(i + d)
    for l in range(n_loop):
        ei.append(n_kf - 1 - l); ej.append(l)
    meas = []
    for i, j in zip(ei, ej):
        m = compose(gt[j], inverse(gt[i]))
        d = (_small_rot(rng.normal(0, meas_noise, 3)), rng.normal(0, meas_noise, 3), 1.0)
        meas.append(compose(d, m))

    def pack(a):
        return np.concatenate([rot_to_quat(a[0]), a[1], [a[2]]])

    fixed = np.zeros(n_kf, np.uint8); fixed[0] = 1
    return dict(verts_gt=np.array([pack(a) for a in gt]), verts=np.array([pack(a) for a in est]), fixed=fixed,
                edge_i=np.array(ei, np.int32), edge_j=np.array(ej, np.int32), meas=np.array([pack(a) for a in meas]))


def sim3_pair_problem(n=120, seq_id=0, width=1280, height=720, scale=1.0, outlier_frac=0.1, pix_noise=0.7, init_noise=(0.02, 0.15, 0.0)):
    """Loop-candidate alignment ([UPSTREAM] transform_optimizer): landmarks seen by keyframe 2 (camera-2 coordinates p2c) and
    their matches in keyframe 1 (p1c = S12 p2c), keypoints of both with level-dependent noise, a fraction of wrong matches,
    and a perturbed initial S12 = (qw qx qy qz tx ty tz s)."""
    rng = np.random.default_rng(0x5EED0000 + 104729 * seq_id + 29)
    k = intrinsics(width, height)
    cam = np.array([k["fx"], k["fy"], k["cx"], k["cy"]])
    R = _small_rot(rng.normal(0, 0.1, 3)); t = rng.normal(0, 0.5, 3)
    p2 = np.stack([rng.uniform(-6, 6, n), rng.uniform(-3, 3, n), rng.uniform(4, 25, n)], axis=1)
    p1 = scale * (p2 @ R.T) + t
    ok = p1[:, 2] > 1.0
    p1, p2 = p1[ok], p2[ok]; n = len(p1)

    def proj(p):
        return np.stack([cam[0] * p[:, 0] / p[:, 2] + cam[2], cam[1] * p[:, 1] / p[:, 2] + cam[3]], axis=1)

    lv1, lv2 = rng.integers(0, 8, n), rng.integers(0, 8, n)
    s1, s2 = 1.2 ** lv1, 1.2 ** lv2
    o1 = proj(p1) + rng.normal(0, pix_noise, (n, 2)) * s1[:, None]
    o2 = proj(p2) + rng.normal(0, pix_noise, (n, 2)) * s2[:, None]
    bad = rng.random(n) < outlier_frac
    o1[bad] += rng.uniform(-80, 80, (int(bad.sum()), 2))
    dR = _small_rot(rng.normal(0, init_noise[0], 3))
    s12 = np.concatenate([rot_to_quat(dR @ R), dR @ t + rng.normal(0, init_noise[1], 3), [scale * math.exp(rng.normal(0, init_noise[2])) if init_noise[2] else scale]])
    return dict(p1c=p1 + rng.normal(0, 0.01, p1.shape), p2c=p2 + rng.normal(0, 0.01, p2.shape), obs1=o1, obs2=o2,
                inv_sigma2_1=1.0 / s1 ** 2, inv_sigma2_2=1.0 / s2 ** 2, cam1=cam, cam2=cam.copy(), s12=s12,
                s12_gt=np.concatenate([rot_to_quat(R), t, [scale]]), outlier=bad)
