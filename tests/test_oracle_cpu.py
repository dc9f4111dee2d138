"""CPU tests of the oracle: golden vectors (tests/golden, made by tools/make_golden.py) and independent
cross-checks of every building block against plain numpy/scipy restatements of the published definitions.
The reference has no fixtures for this path (parity unpinned)."""
import numpy as np
import pytest

from conftest import golden
from lpslam_amd import synth


def test_geometry_matches_survey(oracle):
    p = oracle.params(2000, 1.2, 8)
    lw, lh = oracle.pyramid_sizes(1280, 720, p)
    assert lw == [1280, 1067, 889, 741, 617, 514, 429, 357]
    assert lh == [720, 600, 500, 417, 347, 289, 241, 201]
    assert oracle.keypts_per_level(p) == [434, 362, 302, 251, 209, 175, 145, 122]
    assert sum(oracle.keypts_per_level(p)) == 2000


def test_golden_pyramid(oracle):
    g = golden("g1_pyramid.npz")
    p = oracle.params(60, 1.2, 3)
    lw, lh = oracle.pyramid_sizes(96, 96, p)
    l1 = oracle.resize(g["image"], lw[1], lh[1])
    l2 = oracle.resize(l1, lw[2], lh[2])
    assert np.array_equal(l1, g["level1"]) and np.array_equal(l2, g["level2"])


def test_resize_close_to_float_bilinear(oracle):
    img = synth.random_image(200, 150, seed=3)
    dw, dh = 167, 125
    out = oracle.resize(img, dw, dh).astype(np.float64)
    sx, sy = 200 / dw, 150 / dh
    xs = np.clip((np.arange(dw) + 0.5) * sx - 0.5, 0, 199); ys = np.clip((np.arange(dh) + 0.5) * sy - 0.5, 0, 149)
    x0 = np.floor(xs).astype(int); y0 = np.floor(ys).astype(int)
    x1 = np.minimum(x0 + 1, 199); y1 = np.minimum(y0 + 1, 149)
    fx = (xs - x0)[None, :]; fy = (ys - y0)[:, None]
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx) * (1 - fy) + (f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx) * fy
    assert np.abs(out - ref).max() <= 1.0


def _is_fast_corner(img, x, y, t):
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
            (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    v = int(img[y, x])
    vals = [int(img[y + dy, x + dx]) for dx, dy in ring]
    for sign in (1, -1):
        flags = [(sign * (v - p)) > t for p in vals]
        ff = flags + flags
        run = 0
        for b in ff:
            run = run + 1 if b else 0
            if run >= 9:
                return True
    return False


def test_fast_matches_definition(oracle):
    img = synth.random_image(70, 70, seed=5)
    got = oracle.fast(img, 20, nms=False)
    got_set = {(int(c["x"]), int(c["y"])) for c in got}
    ref = {(x, y) for y in range(3, 67) for x in range(3, 67) if _is_fast_corner(img, x, y, 20)}
    assert got_set == ref and len(ref) > 5
    # score = largest threshold for which the pixel is still a corner
    for c in got[:40]:
        s = int(c["score"])
        assert _is_fast_corner(img, c["x"], c["y"], s) and not _is_fast_corner(img, c["x"], c["y"], s + 1)


def test_fast_nms_strict(oracle):
    img = synth.random_image(70, 70, seed=6)
    allc = oracle.fast(img, 7, nms=False); kept = oracle.fast(img, 7, nms=True)
    score = np.zeros((70, 70), int)
    for c in allc:
        score[c["y"], c["x"]] = c["score"]
    ref = []
    for c in allc:
        x, y, s = c["x"], c["y"], c["score"]
        nb = score[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
        if s > nb.max():
            ref.append((x, y, s))
    assert [(c["x"], c["y"], c["score"]) for c in kept] == ref


def test_golden_fast_level(oracle):
    g = golden("g2_fast.npz")
    c = oracle.fast_level(g["image"], 20, 7)
    assert np.array_equal(c["x"], g["x"]) and np.array_equal(c["y"], g["y"]) and np.array_equal(c["score"], g["score"])


def test_fast_level_empty_and_fallback(oracle):
    flat = np.full((120, 160), 100, np.uint8)
    assert len(oracle.fast_level(flat)) == 0
    weak = flat.copy(); weak[60:, 80:] = 112          # a 12-level step: invisible at 20, a corner at 7
    yy, xx = np.mgrid[0:120, 0:160]
    weak = (weak + (xx * 7 + yy * 13) % 3).astype(np.uint8)           # break score plateaus (strict NMS)
    c = oracle.fast_level(weak, 20, 7)
    assert len(c) > 0 and (c["score"] < 20).all()


def test_distribute_properties(oracle):
    img = synth.random_image(320, 240, seed=8)
    c = oracle.fast_level(img)
    for quota in (1, 7, 50, 200):
        sel = oracle.distribute(c, 320, 240, quota)
        assert len(set(sel.tolist())) == len(sel)
        assert quota <= len(sel) <= max(quota + 3, 8) or len(sel) == len(c)
    far = c[[0, len(c) // 2, len(c) - 1]]                      # fewer, well separated candidates than the quota: all kept
    assert sorted(oracle.distribute(far, 320, 240, 50).tolist()) == [0, 1, 2]
    # upstream stops as soon as a pass creates no new node: neighbours sharing one quadrant collapse to the best one
    near = far[[0, 0, 0]].copy(); near["x"] += [0, 1, 0]; near["y"] += [0, 0, 1]; near["score"] = [10, 30, 20]
    assert oracle.distribute(near, 320, 240, 50).tolist() == [1]
    assert len(oracle.distribute(c[:0], 320, 240, 50)) == 0


def test_gauss_close_to_float(oracle):
    from scipy import ndimage
    img = synth.random_image(96, 80, seed=9)
    out = oracle.gauss7(img).astype(np.float64)
    k = np.exp(-np.arange(-3, 4) ** 2 / 8.0); k /= k.sum()
    ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    assert np.abs(out - ref).max() <= 2.0
    assert oracle.gauss7(np.full((40, 40), 77, np.uint8)).min() == 77     # weights sum to exactly 1


def test_atan2_and_sincos(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        y, x = rng.normal(size=2) * 1000
        ref = np.degrees(np.arctan2(y, x)) % 360
        got = oracle.lib().ora_fast_atan2(np.float32(y), np.float32(x))
        assert abs((got - ref + 180) % 360 - 180) < 0.3
    for a in np.linspace(0, 360, 721):
        s, c = oracle.sincos_deg(a)
        r = np.radians(np.float64(np.float32(a)))
        assert abs(s - np.sin(r)) < 7e-8 and abs(c - np.cos(r)) < 7e-8        # float rounding of the exact value


def test_ic_angle_follows_gradient(oracle):
    yy, xx = np.mgrid[0:64, 0:64]
    img = np.clip(100 + 3 * (xx - 32), 0, 255).astype(np.uint8)       # brighter to the right -> 0 degrees
    assert min(oracle.ic_angle(img, 32, 32), 360 - oracle.ic_angle(img, 32, 32)) < 1.0
    assert abs(oracle.ic_angle(img.T.copy(), 32, 32) - 90) < 1.0        # brighter downwards -> 90 degrees


def test_golden_orb(oracle):
    g = golden("g3_orb.npz")
    kp, desc, cc, _ = oracle.extract(g["image"], oracle.params(150, 1.2, 3))
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(kp[f], g[f]), f
    assert np.array_equal(desc, g["desc"]) and np.array_equal(cc, g["cand_count"])


def test_descriptor_rotation_invariance(oracle):
    """A keypoint's descriptor on a 90-degree rotated image stays close (rBRIEF steers by the patch orientation)."""
    img = synth.random_image(200, 200, seed=21)
    p = oracle.params(120, 1.2, 1)
    k0, d0, _, _ = oracle.extract(img, p)
    rot = np.ascontiguousarray(np.rot90(img))                         # (x, y) -> (y, W-1-x)
    k1, d1, _, _ = oracle.extract(rot, p)
    pos1 = {(int(a["x"]), int(a["y"])): i for i, a in enumerate(k1)}
    dists = []
    for i, a in enumerate(k0):
        j = pos1.get((int(a["y"]), 199 - int(a["x"])))
        if j is not None:
            dists.append(int(np.unpackbits(d0[i] ^ d1[j]).sum()))
    assert len(dists) > 30 and np.median(dists) < 40


def test_hamming_and_golden_bf(oracle):
    g = golden("g4_bf.npz")
    q, t = g["q"], g["t"]
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
    bi, bd, sd = oracle.match_bf_knn2(q, t)
    assert np.array_equal(bi, d.argmin(axis=1)) and np.array_equal(bd, d.min(axis=1))
    assert np.array_equal(sd, np.sort(d, axis=1)[:, 1])
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_dist"]) and np.array_equal(sd, g["second_dist"])
    mq, mt, md = oracle.match_bf(q, t, 50, 0.9, True)
    assert np.array_equal(mq, g["mq"]) and np.array_equal(mt, g["mt"]) and np.array_equal(md, g["md"])
    e = oracle.match_bf_knn2(q[:0], t)
    assert len(e[0]) == 0
    bi0, bd0, sd0 = oracle.match_bf_knn2(q, t[:0])
    assert (bi0 == -1).all() and (bd0 == 257).all()


def test_golden_bf_full_size(oracle):
    """G8, the benchmark's 2000 x 2000 shape: the oracle against numpy's unpackbits Hamming matrix and the committed vector."""
    g = golden("g8_bf2000.npz")
    q, t = g["q"], g["t"]
    bi, bd, sd = oracle.match_bf_knn2(q, t)
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_dist"]) and np.array_equal(sd, g["second_dist"])
    pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
    for lo in range(0, 2000, 250):                      # independent restatement, in slabs to bound memory
        d = pop[q[lo:lo + 250, None, :] ^ t[None, :, :]].sum(axis=2)
        assert np.array_equal(bi[lo:lo + 250], d.argmin(axis=1)) and np.array_equal(bd[lo:lo + 250], d.min(axis=1))
        assert np.array_equal(sd[lo:lo + 250], np.partition(d, 1, axis=1)[:, 1])
    assert bi[0] == 5 and bd[0] == 0 and sd[0] == 0 and bi[1] == 1023 and bi[3] == 0       # planted ties: first minimum wins
    mq, mt, md = oracle.match_bf(q, t, 100, 0.9, True)
    assert np.array_equal(mq, g["mq"]) and np.array_equal(mt, g["mt"]) and np.array_equal(md, g["md"])


def test_golden_stereo_720(oracle):
    """G9: a 1280x720 pair at the benchmark's configuration; images regenerated by the committed generator and pinned by hash."""
    import hashlib
    g = golden("g9_stereo720.npz")
    l, r = synth.StereoSequence(1280, 720, 9).frame(2)
    assert hashlib.sha256(l.tobytes()).hexdigest() == str(g["sha_left"]) and hashlib.sha256(r.tobytes()).hexdigest() == str(g["sha_right"])
    p = oracle.params(2000, 1.2, 8)
    kl, dl, _, pl = oracle.extract(l, p, True)
    kr, dr, _, pr = oracle.extract(r, p, True)
    for f in kl.dtype.names:
        assert np.array_equal(kl[f], g["kl"][f]) and np.array_equal(kr[f], g["kr"][f])
    assert np.array_equal(dl, g["dl"]) and np.array_equal(dr, g["dr"])
    k = synth.intrinsics(1280, 720)
    xr, dep, bi, nv = oracle.match_stereo(pl, pr, p, kl, dl, kr, dr, k["fxb"], k["baseline"])
    assert np.array_equal(xr, g["x_right"]) and np.array_equal(dep, g["depth"]) and np.array_equal(bi, g["best_idx"]) and nv == int(g["n_valid"])
    ok = dep > 0
    assert ok.sum() > 300 and np.allclose(dep[ok], k["fxb"] / (kl["x"][ok] - xr[ok]), rtol=1e-5)


def test_golden_stereo_and_depth_truth(oracle):
    g = golden("g6_stereo.npz")
    p = oracle.params(400, 1.2, 4)
    kl, dl, _, pl = oracle.extract(g["left"], p, True)
    kr, dr, _, pr = oracle.extract(g["right"], p, True)
    k = synth.intrinsics(320, 240)
    xr, dep, bi, nv = oracle.match_stereo(pl, pr, p, kl, dl, kr, dr, k["fxb"], k["baseline"])
    assert np.array_equal(xr, g["x_right"]) and np.array_equal(dep, g["depth"]) and np.array_equal(bi, g["best_idx"])
    ok = dep > 0
    assert ok.sum() > 30
    disp = kl["x"][ok] - xr[ok]
    assert np.allclose(dep[ok], k["fxb"] / disp, rtol=1e-5) and (disp > 0).all()


def test_golden_ba(oracle):
    g = golden("g5_ba.npz")
    cam = dict(zip(("fx", "fy", "cx", "cy", "fxb"), g["cam"]))
    obs = np.zeros(len(g["obs_pose"]), oracle.OBS_DTYPE)
    obs["pose"] = g["obs_pose"]; obs["point"] = g["obs_point"]; obs["u"] = g["obs_uvr"][:, 0]; obs["v"] = g["obs_uvr"][:, 1]
    obs["ur"] = g["obs_uvr"][:, 2]; obs["inv_sigma2"] = g["obs_inv_sigma2"]
    poses, points, log = oracle.ba_optimize(g["poses0"], g["fixed"], g["points0"], obs, cam, True, 10)
    assert np.allclose(log["chi2_after"], g["chi2_after"], rtol=1e-12) and np.array_equal(log["trials"], g["trials"])
    assert np.allclose(poses, g["poses"], atol=1e-12) and np.allclose(points, g["points"], atol=1e-10)
    assert (np.diff(np.r_[log["chi2_before"][0], log["chi2_after"]]) <= 0).all()


def test_ba_jacobians_by_finite_differences(oracle):
    """One Gauss-Newton step with lambda ~ 0 on a noise-free, slightly perturbed problem must land on the optimum:
    this only holds if residual and analytic Jacobians (sign conventions of SURVEY.md section 8(a)) are consistent."""
    prob = synth.ba_problem(5, 80, 320, 640, 480, seq_id=2, pose_noise=(1e-3, 1e-2), point_noise=1e-2, pix_noise=0.0)
    obs = oracle.ba_obs(prob)
    chi0, _ = oracle.ba_chi2(prob["poses"], prob["points"], obs, prob["cam"])
    poses, points, log = oracle.ba_optimize(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"], False, 8)
    chi1, pos = oracle.ba_chi2(poses, points, obs, prob["cam"])
    assert chi0.sum() > 1.0 and chi1.sum() < 1e-5 * chi0.sum() and pos.all()
    # every damped Gauss-Newton step is accepted and contracts strongly: needs residual-consistent Jacobians
    assert (log["trials"] == 1).all() and log["chi2_after"][2] < 1e-4 * log["chi2_before"][0]


def test_ba_matches_scipy_least_squares(oracle):
    from scipy.optimize import least_squares
    prob = synth.ba_problem(3, 25, 70, 640, 480, seq_id=4, pose_noise=(2e-3, 1e-2), point_noise=1e-2)
    obs = oracle.ba_obs(prob)
    cam = prob["cam"]
    poses, points, log = oracle.ba_optimize(prob["poses"], prob["fixed"], prob["points"], obs, cam, False, 60)

    def unpack(x):
        P = prob["poses"].copy(); P[1:] = x[:14].reshape(2, 7); P[1:, :4] /= np.linalg.norm(P[1:, :4], axis=1, keepdims=True)
        return P, x[14:].reshape(-1, 3)

    def resid(x):        # independent numpy restatement of the stereo reprojection residual
        P, X = unpack(x)
        q = P[obs["pose"], :4]; t = P[obs["pose"], 4:]
        w, qx, qy, qz = q.T
        R = np.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - w * qz), 2 * (qx * qz + w * qy),
                      2 * (qx * qy + w * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - w * qx),
                      2 * (qx * qz - w * qy), 2 * (qy * qz + w * qx), 1 - 2 * (qx * qx + qy * qy)], axis=1).reshape(-1, 3, 3)
        pc = np.einsum("nij,nj->ni", R, X[obs["point"]]) + t
        u = cam["fx"] * pc[:, 0] / pc[:, 2] + cam["cx"]; v = cam["fy"] * pc[:, 1] / pc[:, 2] + cam["cy"]
        ur = u - cam["fxb"] / pc[:, 2]
        sw = np.sqrt(obs["inv_sigma2"])
        return np.r_[(obs["u"] - u) * sw, (obs["v"] - v) * sw, (obs["ur"] - ur) * sw]
    # start MINPACK's LM at the oracle's answer: an independent optimiser must not find a lower cost nearby
    x0 = np.r_[poses[1:].ravel(), points.ravel()]
    assert abs(resid(x0).dot(resid(x0)) - log["chi2_after"][-1]) < 1e-9 * log["chi2_after"][-1]
    sol = least_squares(resid, x0, method="lm", max_nfev=4000)
    assert 2 * sol.cost > log["chi2_after"][-1] * (1 - 1e-5)
    assert log["chi2_after"][-1] < 0.5 * log["chi2_before"][0]


def test_ba_local_and_pose_optimizer(oracle):
    prob = synth.ba_problem(6, 150, 700, 640, 480, seq_id=7)
    obs = oracle.ba_obs(prob)
    bad = np.arange(0, len(obs), 37)
    obs["u"][bad] += 40.0                                              # gross outliers
    poses, points, out = oracle.ba_local(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"])
    assert out[bad].mean() > 0.9 and out.mean() < 0.3
    # motion-only: recover one keyframe's pose from the ground-truth landmarks
    sel = obs[obs["pose"] == 3].copy(); sel["pose"] = 0
    pose, outl, n_in = oracle.pose_optimize(prob["poses"][3], prob["points_gt"], sel, prob["cam"])
    assert np.abs(pose[4:] - prob["poses_gt"][3, 4:]).max() < 0.05 and n_in > 0.6 * len(sel)


# ---- Sim3 pose graph (ora_sim3.c) ------------------------------------------------------------------------------------
def _sim3_matrix(s):
    q = s[:4] / np.linalg.norm(s[:4])
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    M = np.eye(4); M[:3, :3] = s[7] * R; M[:3, 3] = s[4:7]
    return M


def test_sim3_exp_is_the_matrix_exponential_and_log_inverts_it(oracle):
    from scipy.linalg import expm
    rng = np.random.default_rng(3)
    for scale in (1.0, 0.3, 1e-3):
        for _ in range(5):
            u = rng.normal(0, scale, 7)
            th = np.linalg.norm(u[:3])
            if th > 3.0:                  # log returns the rotation below pi
                u[:3] *= 3.0 / th
            G = np.zeros((4, 4))
            G[:3, :3] = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]]) + u[6] * np.eye(3)
            G[:3, 3] = u[3:6]
            s = oracle.sim3_exp(u)
            assert np.allclose(_sim3_matrix(s), expm(G), atol=1e-12)
            assert np.allclose(oracle.sim3_log(s), u, atol=1e-10)
    # pure scale / pure rotation branches of the coefficient table
    for u in ([0, 0, 0, 1, 2, 3, 0.5], [0.4, 0, 0, 1, 2, 3, 0.0], [0, 0, 0, 1, 2, 3, 0.0]):
        u = np.array(u, float)
        assert np.allclose(oracle.sim3_log(oracle.sim3_exp(u)), u, atol=1e-12)


def test_sim3_group_operations(oracle):
    rng = np.random.default_rng(4)
    a, b = oracle.sim3_exp(rng.normal(0, 0.5, 7)), oracle.sim3_exp(rng.normal(0, 0.5, 7))
    assert np.allclose(_sim3_matrix(oracle.sim3_mul(a, b)), _sim3_matrix(a) @ _sim3_matrix(b), atol=1e-12)
    assert np.allclose(_sim3_matrix(oracle.sim3_inv(a)), np.linalg.inv(_sim3_matrix(a)), atol=1e-12)


def test_golden_sim3_graph(oracle):
    g = golden("g7_sim3.npz")
    assert np.allclose(oracle.sim3_exp(g["exp_in"]), g["exp_out"], atol=1e-15)
    e = oracle.sim3_edges(g["edge_i"], g["edge_j"], g["meas"])
    v, log = oracle.sim3_graph_optimize(g["verts0"], g["fixed"], e, True, 10)
    assert np.allclose(log["chi2_after"], g["chi2_after"], rtol=1e-9) and np.array_equal(log["trials"], g["trials"])
    assert np.allclose(v, g["verts"], atol=1e-9)
    assert np.array_equal(v[0], g["verts0"][0]) and np.all(v[:, 7] == 1.0)        # fixed vertex, fixed scale
    ef = oracle.sim3_edges(g["f_edge_i"], g["f_edge_j"], g["f_meas"])
    vf, logf = oracle.sim3_graph_optimize(g["f_verts0"], g["fixed"], ef, False, 10)
    assert np.allclose(logf["chi2_after"], g["f_chi2_after"], rtol=1e-9) and np.allclose(vf, g["f_verts"], atol=1e-9)
    assert np.abs(vf[1:, 7] - 1.0).max() > 1e-4                                   # free scale moves


def test_sim3_graph_recovers_the_loop(oracle):
    from lpslam_amd import synth
    p = synth.pose_graph_problem(30, 2, meas_noise=0.0, drift_scale=0.01)
    e = oracle.sim3_edges(p["edge_i"], p["edge_j"], p["meas"])
    chi0 = oracle.sim3_graph_chi2(p["verts"], e)
    v, log = oracle.sim3_graph_optimize(p["verts"], p["fixed"], e, False, 50)
    assert (np.diff(np.r_[chi0, log["chi2_after"]]) <= 1e-15).all()
    assert log["chi2_after"][-1] < 1e-12 * chi0 + 1e-14
    # exact measurements: the optimum is the ground truth (up to the quaternion sign)
    sign = np.sign(np.sum(v[:, :4] * p["verts_gt"][:, :4], axis=1))[:, None]
    assert np.abs(v[:, :4] * sign - p["verts_gt"][:, :4]).max() < 1e-6
    assert np.abs(v[:, 4:] - p["verts_gt"][:, 4:]).max() < 1e-5


def test_sim3_transform_optimizer_flow(oracle):
    from lpslam_amd import synth
    p = synth.sim3_pair_problem(150, 1)
    s, inl, n = oracle.sim3_transform_optimize(p["s12"], oracle.sim3_pairs(p), p["cam1"], p["cam2"], 10.0, True)
    assert n == inl.sum() and n >= 0.8 * len(inl)
    assert not (inl.astype(bool) & p["outlier"]).sum() > 1            # wrong matches are cut (one may sit inside the gate)
    assert np.abs(s - p["s12_gt"]).max() < 0.1 * np.abs(p["s12"] - p["s12_gt"]).max() and s[7] == 1.0
    pf = synth.sim3_pair_problem(150, 2, scale=1.15, init_noise=(0.02, 0.15, 0.03))
    sf, _, nf = oracle.sim3_transform_optimize(pf["s12"], oracle.sim3_pairs(pf), pf["cam1"], pf["cam2"], 10.0, False)
    assert nf > 100 and abs(sf[7] - 1.15) < 0.02
    bad = synth.sim3_pair_problem(30, 9, outlier_frac=0.8)
    assert oracle.sim3_transform_optimize(bad["s12"], oracle.sim3_pairs(bad), bad["cam1"], bad["cam2"], 10.0, True)[2] == 0


def test_projection_match_against_a_plain_restatement(oracle):
    """ora_match_projection vs a direct Python loop of the published algorithm (cell-scan order, sequential assignment)."""
    rng = np.random.default_rng(11)
    w, h, n = 320, 240, 150
    kp = np.zeros(n, oracle.KP_DTYPE)
    kp["x"] = rng.uniform(0, w, n).astype(np.float32); kp["y"] = rng.uniform(0, h, n).astype(np.float32); kp["octave"] = rng.integers(0, 4, n)
    desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    desc[1::2] = desc[0::2]                                     # pairs of identical descriptors: ties decided by scan order
    xr = np.where(rng.random(n) < 0.7, kp["x"] - rng.uniform(2, 20, n), -1).astype(np.float32)
    nq = 80
    q = np.zeros(nq, oracle.PROJ_QUERY_DTYPE)
    src = rng.integers(0, n, nq)
    q["x"] = kp["x"][src] + rng.normal(0, 5, nq); q["y"] = kp["y"][src] + rng.normal(0, 5, nq)
    q["x_right"] = np.where(xr[src] > 0, xr[src] + rng.normal(0, 5, nq), -1); q["radius"] = 40.0
    q["min_level"] = kp["octave"][src] - 1; q["max_level"] = kp["octave"][src]
    qd = desc[src].copy(); qd[:, 0] ^= rng.integers(0, 8, nq).astype(np.uint8)
    got_i, got_d, got_n = oracle.match_projection(kp, desc, xr, w, h, q, qd, 60, 0.9)
    cell = np.minimum(np.floor(kp["x"] * (64.0 / w)).astype(int), 63) * 48 + np.minimum(np.floor(kp["y"] * (48.0 / h)).astype(int), 47)
    order = np.lexsort((np.arange(n), cell))
    taken = np.zeros(n, bool); want = np.full(nq, -1)
    for k in range(nq):
        best, second, bl, sl, bi = 256, 256, -1, -1, -1
        for i in order:
            if not (abs(kp["x"][i] - q["x"][k]) < q["radius"][k] and abs(kp["y"][i] - q["y"][k]) < q["radius"][k]):
                continue
            if kp["octave"][i] < q["min_level"][k] or kp["octave"][i] > q["max_level"][k] or taken[i]:
                continue
            if xr[i] > 0 and q["x_right"][k] >= 0 and q["radius"][k] < abs(q["x_right"][k] - xr[i]):
                continue
            d = int(np.unpackbits(qd[k] ^ desc[i]).sum())
            if d < best:
                second, sl, best, bl, bi = best, bl, d, kp["octave"][i], i
            elif d < second:
                second, sl = d, kp["octave"][i]
        if bi >= 0 and best <= 60 and not (bl == sl and best > np.float32(0.9) * np.float32(second)):
            want[k] = bi; taken[bi] = True
    assert np.array_equal(got_i, want) and got_n == (want >= 0).sum() and got_n > 20
    # orientation filter keeps the dominant bins only
    aq = rng.uniform(0, 360, nq).astype(np.float32); at = rng.uniform(0, 360, n).astype(np.float32)
    good = got_i >= 0
    aq[good] = (at[got_i[good]] + 33.0) % 360                  # consistent rotation ...
    spoil = np.flatnonzero(good)[:3]; aq[spoil] = (aq[spoil] + 170.0) % 360      # ... except three matches
    filt, kept = oracle.match_orientation_filter(aq, at, got_i)
    assert kept == good.sum() - 3 and (filt[spoil] == -1).all()


def test_fuse_and_area_match_semantics(oracle):
    rng = np.random.default_rng(21)
    w, h, n = 320, 240, 120
    kp = np.zeros(n, oracle.KP_DTYPE)
    kp["x"] = rng.uniform(0, w, n).astype(np.float32); kp["y"] = rng.uniform(0, h, n).astype(np.float32); kp["octave"] = rng.integers(0, 3, n)
    desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    isq = (1.0 / (1.2 ** np.arange(8)) ** 2).astype(np.float32)
    # fuse: a query exactly on a keypoint with its descriptor matches it; 3 px off at level 0 fails the 5.99 gate (9 > 5.99)
    q = np.zeros(2, oracle.PROJ_QUERY_DTYPE)
    k0 = int(np.flatnonzero(kp["octave"] == 0)[0])
    q["x"] = kp["x"][k0] + np.array([0.5, 3.0], np.float32); q["y"] = kp["y"][k0]; q["x_right"] = -1; q["radius"] = 10
    q["min_level"] = 0; q["max_level"] = 0
    idx, dist, nm = oracle.match_fuse(kp, desc, None, w, h, isq, q, desc[[k0, k0]], 50)
    assert idx[0] == k0 and dist[0] == 0 and idx[1] != k0
    # area: two queries for the same keypoint, the second one closer -> the first loses it
    q2 = np.zeros(2, oracle.PROJ_QUERY_DTYPE)
    q2["x"] = kp["x"][k0]; q2["y"] = kp["y"][k0]; q2["radius"] = 5; q2["min_level"] = 0; q2["max_level"] = 0
    worse = desc[k0].copy(); worse[0] ^= 0x07
    idx2, n2 = oracle.match_area(kp, desc, w, h, q2, np.stack([worse, desc[k0]]), 50, 0.9)
    assert list(idx2) == [-1, k0] and n2 == 1
    idx3, n3 = oracle.match_area(kp, desc, w, h, q2, np.stack([desc[k0], worse]), 50, 0.9)
    assert list(idx3) == [k0, -1] and n3 == 1


def test_closed_loop_oracle_is_reproducible(oracle):
    """the golden is what oracle/tracker.py makes today (first 8 frames; the whole sequence is regenerated by tools/make_golden_track.py)"""
    from oracle import tracker as T
    g = golden("g10_track.npz")
    W, H = 640, 480
    k = synth.intrinsics(W, H)
    seq = synth.StereoSequence(W, H, 4, n_points=6000)
    trk = T.StereoTracker(W, H, k, max_keypoints=1000, num_levels=4, scale_factor=1.2, keyframe_interval=4, local_window=10)
    for i in range(8):
        pose = trk.feed(*seq.frame(i))
        assert np.allclose(pose, g["poses"][i], rtol=0, atol=1e-12), i
