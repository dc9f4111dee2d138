"""CPU tests of the monocular two-view initialisation (lpslam_amd/host/two_view.cpp through its C shim) against the numpy
restatement (oracle/two_view.py) and against the synthetic ground truth.  [UPSTREAM] initialize::perspective; reference call site
src/Trackers/OpenVSLAMTracker.cpp:120 (feed_monocular_frame)."""
import ctypes as C

import numpy as np
import pytest

K = np.array([525.0, 525.0, 320.0, 240.0])


@pytest.fixture(scope="module")
def lib():
    from lpslam_amd import _build
    l = C.CDLL(_build.host_library())
    l.lpslam_two_view_initialize.restype = C.c_int
    l.lpslam_two_view_initialize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_uint32] + [C.c_void_p] * 10
    return l


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * kx + (1 - np.cos(ang)) * kx @ kx


def scene(n=300, planar=False, seed=0, r=None, t=None, noise=0.3, outliers=0.0, slope=(0.25, -0.1)):
    rng = np.random.default_rng(seed)
    r = _rot([0.1, 1.0, 0.2], 0.06) if r is None else r
    t = np.array([-0.4, 0.05, 0.1]) if t is None else np.asarray(t, float)
    x = np.column_stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)])
    if planar:
        x[:, 2] = 6.0 + slope[0] * x[:, 0] + slope[1] * x[:, 1]
    x2 = x @ r.T + t
    p1 = np.column_stack([K[0] * x[:, 0] / x[:, 2] + K[2], K[1] * x[:, 1] / x[:, 2] + K[3]]) + rng.normal(0, noise, (n, 2))
    p2 = np.column_stack([K[0] * x2[:, 0] / x2[:, 2] + K[2], K[1] * x2[:, 1] / x2[:, 2] + K[3]]) + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outliers
    p2[bad] = np.column_stack([rng.uniform(0, 640, bad.sum()), rng.uniform(0, 480, bad.sum())])
    perm = rng.permutation(n)                        # the current frame lists its keypoints in another order
    kp_cur = np.zeros((n, 2), np.float32); kp_cur[perm] = p2
    matches = np.column_stack([np.arange(n), perm]).astype(np.int32)
    return p1.astype(np.float32), kp_cur, matches, r, t / np.linalg.norm(t), x, bad


def run(lib, kp_ref, kp_cur, matches, sigma=1.0, iters=100, seed=12345):
    n = len(matches)
    out = dict(R=np.zeros(9), t=np.zeros(3), H=np.zeros(9), F=np.zeros(9), scores=np.zeros(2), par=C.c_double(), model=C.c_int32(),
               inl=np.zeros(max(n, 1), np.uint8), tri=np.zeros(max(n, 1), np.uint8), pts=np.zeros((max(n, 1), 3)))
    k = np.ascontiguousarray(K); a = np.ascontiguousarray(kp_ref, np.float32); b = np.ascontiguousarray(kp_cur, np.float32)
    m = np.ascontiguousarray(matches, np.int32)
    ok = lib.lpslam_two_view_initialize(_p(k), _p(a), _p(b), _p(m), n, sigma, iters, seed, _p(out["R"]), _p(out["t"]), _p(out["H"]), _p(out["F"]),
                                        _p(out["scores"]), C.addressof(out["par"]), C.addressof(out["model"]), _p(out["inl"]), _p(out["tri"]), _p(out["pts"]))
    out["ok"] = bool(ok); out["model"] = out["model"].value; out["par"] = out["par"].value
    out["R"] = out["R"].reshape(3, 3)
    return out


def rot_angle(a, b):
    return np.arccos(np.clip((np.trace(a.T @ b) - 1) / 2, -1, 1))


def test_symmetric_eigen_solver(lib):
    rng = np.random.default_rng(1)
    for n in (3, 4, 9):
        a = rng.normal(size=(n, n)); a = a @ a.T
        ev = np.zeros(n); vec = np.zeros((n, n))
        lib.lpslam_sym_eigen(_p(np.ascontiguousarray(a)), n, _p(ev), _p(vec))
        w, _ = np.linalg.eigh(a)
        assert np.allclose(ev, w, rtol=1e-12, atol=1e-12 * w.max())
        assert np.allclose(a @ vec, vec * ev, atol=1e-10 * w.max()) and np.allclose(vec.T @ vec, np.eye(n), atol=1e-12)


@pytest.mark.parametrize("planar,model", [(False, 1), (True, 0)])
def test_general_and_planar_scene(lib, planar, model):
    from oracle import two_view as O
    if planar:      # a strongly tilted plane and a motion for which only one of Faugeras' solutions keeps the points in front
        kp_ref, kp_cur, matches, r, t, x, _ = scene(planar=True, seed=21, r=_rot([0.1, 1.0, 0.2], 0.14), t=[-0.6, -0.34, 0.14], slope=(1.16, -0.82))
    else:
        kp_ref, kp_cur, matches, r, t, x, _ = scene(seed=2)
    g = run(lib, kp_ref, kp_cur, matches)
    o = O.initialize(K, kp_ref, kp_cur, matches, seed=12345)
    assert g["ok"] and o["ok"] and g["model"] == o["model"] == model
    # the two implementations agree ...
    assert np.allclose(g["scores"], [o["score_h"], o["score_f"]], rtol=1e-6)
    assert rot_angle(g["R"], o["R"]) < 1e-6 and np.abs(g["t"] - o["t"]).max() < 1e-6
    assert np.array_equal(g["inl"].astype(bool), o["inlier"]) and np.array_equal(g["tri"].astype(bool), o["triangulated"])
    ok = g["tri"].astype(bool)
    assert np.allclose(g["pts"][ok], o["points"][ok], rtol=1e-6, atol=1e-6)
    # ... and with the truth: rotation within 0.3 degrees, translation direction within 3 degrees, structure up to the scale |t|
    assert rot_angle(g["R"], r) < np.radians(0.3)
    assert np.arccos(np.clip(g["t"] @ t, -1, 1)) < np.radians(3.0)
    assert ok.sum() > 0.9 * len(matches)
    scale = np.median(x[ok][:, 2] / g["pts"][ok][:, 2])
    err = np.linalg.norm(g["pts"][ok] * scale - x[ok], axis=1)          # 0.3 px noise over a 0.4 / 6 baseline-to-depth ratio
    assert np.median(err) < 0.1 and err.max() < 1.5
    assert g["par"] > 1.0


def test_outliers_are_left_out(lib):
    from oracle import two_view as O
    kp_ref, kp_cur, matches, r, t, x, bad = scene(n=400, seed=5, outliers=0.25)
    g = run(lib, kp_ref, kp_cur, matches, iters=200)
    o = O.initialize(K, kp_ref, kp_cur, matches, ransac_iters=200, seed=12345)
    assert g["ok"] and o["ok"] and np.array_equal(g["inl"].astype(bool), o["inlier"])
    inl = g["inl"].astype(bool)
    assert inl[~bad].mean() > 0.9 and inl[bad].mean() < 0.1
    assert rot_angle(g["R"], r) < np.radians(0.5)


def test_rejections(lib):
    from oracle import two_view as O
    # a plane seen under a motion that leaves both of Faugeras' solutions plausible: no clear winner, both sides refuse
    kp_ref, kp_cur, matches, *_ = scene(planar=True, seed=3)
    g = run(lib, kp_ref, kp_cur, matches); o = O.initialize(K, kp_ref, kp_cur, matches, seed=12345)
    assert g["model"] == o["model"] == 0 and not g["ok"] and not o["ok"] and g["par"] > 1.0
    kp_ref, kp_cur, matches, *_ = scene(seed=7)
    assert not run(lib, kp_ref, kp_cur, matches[:7])["ok"]                                # fewer than eight matches
    # (nearly) no baseline: low parallax, no reconstruction
    kp_ref, kp_cur, matches, *_ = scene(seed=8, t=[1e-4, 0, 0])
    assert not run(lib, kp_ref, kp_cur, matches)["ok"]
    # few points: below Initializer.num_min_triangulated_pts
    kp_ref, kp_cur, matches, *_ = scene(n=30, seed=9)
    assert not run(lib, kp_ref, kp_cur, matches)["ok"]


def test_sim3_solver_recovers_a_similarity_and_follows_the_oracle():
    """[UPSTREAM] solve::sim3_solver (Horn's absolute orientation + RANSAC over the reprojection error in both images): the host
    mirror (lpslam_amd/host/two_view.cpp through its C shim) recovers a known Sim3 between two keyframes from matched landmarks
    with 30 % wrong matches, with and without a free scale, and agrees with the numpy restatement (same sampler)."""
    from lpslam_amd import _build
    from oracle import two_view as TV
    l = C.CDLL(_build.host_library())
    l.lpslam_sim3_solve_ransac.restype = C.c_int
    l.lpslam_sim3_solve_ransac.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    cam = np.array([525.0, 525.0, 320.0, 240.0])
    for fix_scale, s_true in ((True, 1.0), (False, 1.37)):
        n = 120
        x2 = np.column_stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)])          # landmarks in camera 2
        r = _rot([0.2, 1.0, -0.1], 0.4); t = np.array([0.8, -0.1, 0.3])
        x1 = s_true * (x2 @ r.T) + t                                                                           # ... and in camera 1
        obs1 = np.column_stack([cam[0] * x1[:, 0] / x1[:, 2] + cam[2], cam[1] * x1[:, 1] / x1[:, 2] + cam[3]]) + rng.normal(0, 0.3, (n, 2))
        obs2 = np.column_stack([cam[0] * x2[:, 0] / x2[:, 2] + cam[2], cam[1] * x2[:, 1] / x2[:, 2] + cam[3]]) + rng.normal(0, 0.3, (n, 2))
        bad = rng.random(n) < 0.3
        x1n = x1 + rng.normal(0, 0.01, x1.shape); x1n[bad] += rng.normal(0, 1.5, (int(bad.sum()), 3))          # wrong matches
        is1 = np.ones(n); is2 = np.ones(n)
        s12 = np.zeros(8); inl = np.zeros(n, np.uint8)
        got = l.lpslam_sim3_solve_ransac(_p(np.ascontiguousarray(x1n)), _p(np.ascontiguousarray(x2)), _p(np.ascontiguousarray(obs1)), _p(np.ascontiguousarray(obs2)),
                                         _p(is1), _p(is2), n, _p(cam), _p(cam), int(fix_scale), 200, 0x9E3779B9, _p(s12), _p(inl))
        want, os12, oinl = TV.sim3_solve_ransac(x1n, x2, obs1, obs2, is1, is2, cam, cam, fix_scale, 200, 0x9E3779B9)
        assert got == want and np.array_equal(inl.astype(bool), oinl) and np.allclose(s12, os12, atol=1e-9)
        assert got >= 0.55 * n and not np.any(inl.astype(bool) & bad & (np.linalg.norm(x1n - x1, axis=1) > 0.5))
        q = s12[:4]; w, x, y, z = q
        rr = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        assert np.abs(rr - r).max() < 0.02 and np.abs(s12[4:7] - t).max() < 0.1 and abs(s12[7] - s_true) < 0.02


def test_pnp_solver_finds_a_pose_without_a_prior_and_follows_the_oracle():
    """[UPSTREAM] solve::pnp_solver, as restated (EPnP on 4-match samples + RANSAC + a refit on the inliers, host/two_view.cpp): a pose far
    from the identity is recovered from landmark / keypoint matches of which 35 % are wrong, and the host mirror agrees with the numpy
    restatement (same sampler, the same Jacobi sweeps: EPnP on four matches has a four-dimensional null space, both sides must pick
    the same basis to test the same hypotheses)."""
    from lpslam_amd import _build
    from oracle import two_view as TV
    l = C.CDLL(_build.host_library())
    l.lpslam_pnp_solve_ransac.restype = C.c_int
    l.lpslam_pnp_solve_ransac.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    cam = np.array([525.0, 525.0, 320.0, 240.0])
    for trial in range(3):
        n = 150
        r = _rot(rng.normal(size=3), 0.3 + 0.5 * trial); t = rng.normal(0, 1.0, 3)
        xc = np.column_stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(3, 12, n)])       # in the camera
        pw = (xc - t) @ r                                                                                  # world: xc = R pw + t
        obs = np.column_stack([cam[0] * xc[:, 0] / xc[:, 2] + cam[2], cam[1] * xc[:, 1] / xc[:, 2] + cam[3]]) + rng.normal(0, 0.4, (n, 2))
        bad = rng.random(n) < 0.35
        obs[bad] = np.column_stack([rng.uniform(0, 640, int(bad.sum())), rng.uniform(0, 480, int(bad.sum()))])
        w = np.ones(n)
        pose = np.zeros(7); inl = np.zeros(n, np.uint8)
        got = l.lpslam_pnp_solve_ransac(_p(np.ascontiguousarray(pw)), _p(np.ascontiguousarray(obs)), _p(w), n, _p(cam), 100, 0x9E3779B9, _p(pose), _p(inl))
        want, opose, oinl = TV.pnp_solve_ransac(pw, obs, w, cam, 100, 0x9E3779B9)
        assert got == want and np.array_equal(inl.astype(bool), oinl) and np.allclose(pose, opose, atol=1e-8)
        assert got >= 0.5 * n and int((inl.astype(bool) & bad).sum()) <= 3
        w_, x, y, z = pose[:4]
        rr = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w_ * z), 2 * (x * z + w_ * y)], [2 * (x * y + w_ * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w_ * x)],
                       [2 * (x * z - w_ * y), 2 * (y * z + w_ * x), 1 - 2 * (x * x + y * y)]])
        assert np.abs(rr - r).max() < 0.02 and np.abs(pose[4:] - t).max() < 0.15


def test_epnp_is_exact_without_noise_and_near_the_least_squares_pose_with_it():
    """EPnP itself (Lepetit, Moreno-Noguer, Fua 2009), independently of the restatement's arithmetic: from five or more exact matches it
    returns the exact pose (the method is algebraically exact there), from four -- a four-dimensional null space, no relinearisation --
    a pose that explains the four matches about half of the time, and from noisy matches a pose within a small multiple of what scipy's reprojection least squares
    (started at the truth) moves -- so the oracle and the product, which share their arithmetic, are both anchored to the published
    algorithm's defining properties."""
    from scipy.optimize import least_squares
    from lpslam_amd import _build
    from oracle import two_view as TV
    l = C.CDLL(_build.host_library())
    l.lpslam_epnp_solve.restype = C.c_int
    rng = np.random.default_rng(9)
    cam = np.array([525.0, 525.0, 320.0, 240.0])

    def product(pw, uv):
        R = np.zeros(9); t = np.zeros(3)
        ok = l.lpslam_epnp_solve(_p(np.ascontiguousarray(pw)), _p(np.ascontiguousarray(uv)), len(pw), _p(cam), _p(R), _p(t))
        return (R.reshape(3, 3), t) if ok else None

    def scene(n, noise):
        r = _rot(rng.normal(size=3), rng.uniform(0.2, 1.2)); t = rng.normal(0, 1.0, 3)
        xc = np.column_stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(3, 12, n)])
        pw = (xc - t) @ r
        uv = np.column_stack([cam[0] * xc[:, 0] / xc[:, 2] + cam[2], cam[1] * xc[:, 1] / xc[:, 2] + cam[3]]) + rng.normal(0, noise, (n, 2))
        return r, t, pw, uv
    for n in (5, 6, 12, 80):
        r, t, pw, uv = scene(n, 0.0)
        got, want = product(pw, uv), TV.epnp_solve(pw, uv, cam)
        assert got is not None and want is not None
        assert np.abs(got[0] - r).max() < 1e-8 and np.abs(got[1] - t).max() < 1e-7, n
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    hits = 0
    for _ in range(60):                                      # four matches: the linearised guesses reach a pose that explains them about half of the time
        r, t, pw, uv = scene(4, 0.0)
        got = product(pw, uv)
        assert got is not None
        pc = pw @ got[0].T + got[1]
        rep = np.column_stack([cam[0] * pc[:, 0] / pc[:, 2] + cam[2], cam[1] * pc[:, 1] / pc[:, 2] + cam[3]])
        hits += int(np.abs(rep - uv).max() < 2.0)
    assert hits >= 15, hits                                  # (measured 0.51 over 400 samples; the RANSAC around it and the refit on >= 5 inliers are what cope with the rest)

    def rvec_pose(x):
        return _rot(x[:3], np.linalg.norm(x[:3])) if np.linalg.norm(x[:3]) > 0 else np.eye(3), x[3:]
    for n in (20, 100):
        r, t, pw, uv = scene(n, 0.5)
        got = product(pw, uv)

        def res(x):
            dr, dt = rvec_pose(x)
            pc = pw @ (dr @ r).T + (t + dt)
            return np.concatenate([cam[0] * pc[:, 0] / pc[:, 2] + cam[2] - uv[:, 0], cam[1] * pc[:, 1] / pc[:, 2] + cam[3] - uv[:, 1]])
        sol = least_squares(res, np.full(6, 1e-9))
        dr, dt = rvec_pose(sol.x)
        r_ml, t_ml = dr @ r, t + dt
        assert np.abs(got[0] - r_ml).max() < 0.01 and np.abs(got[1] - t_ml).max() < 0.08, (n, np.abs(got[0] - r_ml).max(), np.abs(got[1] - t_ml).max())
