"""GPU parity tests of the bundle adjustment (FP64).  Tolerances: chi2 trajectory relative 1e-9, poses within
1e-4 rad / 1e-3 m of the CPU oracle (north-star tolerance), identical lambda-control decisions."""
import numpy as np
import pytest

from conftest import golden
from lpslam_amd import synth

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL, CHI_RTOL = 1e-4, 1e-3, 1e-9


def rot_err(q1, q2):
    return 2 * np.arccos(np.clip(np.abs(np.sum(q1 * q2, axis=1)), 0, 1))


@pytest.fixture(scope="module")
def ctx(hiplib):
    return hiplib.Context(320, 240, 400, 1.2, 4, max_images=1)


def _compare(hiplib, oracle, ctx, prob, robust, iters, active=None):
    obs = oracle.ba_obs(prob)
    op, ox, olog = oracle.ba_optimize(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"], robust, iters, active)
    ba = hiplib.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hiplib.ba_obs_array(prob), prob["cam"])
    if active is not None:
        ba.set_active(active)
    glog = ba.optimize(robust, iters)
    gp, gx = ba.state()
    assert len(glog) == len(olog)
    assert np.allclose(glog["chi2_before"], olog["chi2_before"], rtol=CHI_RTOL)
    assert np.allclose(glog["chi2_after"], olog["chi2_after"], rtol=CHI_RTOL)
    assert np.array_equal(glog["trials"], olog["trials"]) and np.array_equal(glog["status"], olog["status"])
    assert np.allclose(glog["lambda"], olog["lambda"], rtol=1e-6)
    assert rot_err(gp[:, :4], op[:, :4]).max() < ROT_TOL and np.abs(gp[:, 4:] - op[:, 4:]).max() < TRANS_TOL
    assert np.abs(gx - ox).max() < TRANS_TOL
    if ba.solver()[0] == "band":
        # a window whose landmarks are seen by neighbouring keyframes takes the band path (ba_band.inl); the pair lists and the dense
        # panel chain must give the same answer for it
        ba.set_solver("dense"); ba.reset()
        if active is not None:
            ba.set_active(active)           # a reset re-activates every observation
        assert ba.solver()[0] == "dense"
        dlog = ba.optimize(robust, iters)
        dp, dx = ba.state()
        assert len(dlog) == len(olog) and np.allclose(dlog["chi2_after"], olog["chi2_after"], rtol=CHI_RTOL)
        assert np.array_equal(dlog["trials"], olog["trials"]) and np.allclose(dlog["lambda"], olog["lambda"], rtol=1e-6)
        assert np.abs(dp - gp).max() < 1e-7 and np.abs(dx - gx).max() < 1e-7
        ba.set_solver("band"); ba.reset()
        if active is not None:
            ba.set_active(active)
        blog = ba.optimize(robust, iters)
        bp, bx = ba.state()
        assert blog.tobytes() == glog.tobytes() and np.array_equal(bp, gp) and np.array_equal(bx, gx)      # back on the band path: the same bytes
    return ba, gp, gx, glog


def test_golden_toy_problem(hiplib, oracle, ctx):
    g = golden("g5_ba.npz")
    cam = dict(zip(("fx", "fy", "cx", "cy", "fxb"), g["cam"]))
    prob = dict(poses=g["poses0"], points=g["points0"], fixed=g["fixed"], obs_pose=g["obs_pose"], obs_point=g["obs_point"],
                obs_uvr=g["obs_uvr"], obs_inv_sigma2=g["obs_inv_sigma2"], cam=cam)
    ba, gp, gx, glog = _compare(hiplib, oracle, ctx, prob, True, 10)
    assert np.allclose(glog["chi2_after"], g["chi2_after"], rtol=CHI_RTOL)
    assert rot_err(gp[:, :4], g["poses"][:, :4]).max() < ROT_TOL and np.abs(gp[:, 4:] - g["poses"][:, 4:]).max() < TRANS_TOL


@pytest.mark.parametrize("n_kf,n_pts,n_obs,robust", [(2, 20, 40, True), (5, 80, 320, False), (12, 600, 4000, True), (33, 900, 6000, True)])
def test_sizes(hiplib, oracle, ctx, n_kf, n_pts, n_obs, robust):
    prob = synth.ba_problem(n_kf, n_pts, n_obs, 640, 480, seq_id=n_kf)
    _compare(hiplib, oracle, ctx, prob, robust, 8)


@pytest.mark.parametrize("n_kf,n_pts,n_obs", [(10, 800, 8000), (6, 1200, 7200)])
def test_long_pair_lists_take_several_workgroups(hiplib, oracle, ctx, n_kf, n_pts, n_obs):
    """Keyframes one frame apart see the same landmarks: every pose-block pair's term list is ~n_pts long, so off-diagonal lists are
    cut into parts too and the table of further parts (k_bs_blkscan) holds more items than the host foresaw from the keyframes'
    observation counts -- the surplus takes the workgroups behind the pairs' part 0 (k_ba_schur).  Dense chain and, where the window
    is banded, the band path against the oracle; twice the same bytes."""
    prob = synth.ba_problem(n_kf, n_pts, n_obs, 640, 480, seq_id=31, kf_stride=1)
    per_kf = np.bincount(prob["obs_pose"], minlength=n_kf)
    assert per_kf.min() > 3 * 256 - 64                     # the diagonal lists AND the off-diagonal ones are far beyond one part
    ba, gp, gx, glog = _compare(hiplib, oracle, ctx, prob, True, 6)
    ba.set_solver("dense"); ba.reset()
    again = ba.optimize(True, 6)
    ba.reset()
    assert ba.optimize(True, 6).tobytes() == again.tobytes()


def test_mono_and_inactive_observations(hiplib, oracle, ctx):
    prob = synth.ba_problem(7, 200, 1100, 640, 480, seq_id=3)
    prob["obs_uvr"][::3, 2] = -1.0                                  # every third edge monocular (2 rows)
    active = np.ones(len(prob["obs_pose"]), np.uint8); active[::5] = 0
    _compare(hiplib, oracle, ctx, prob, True, 6, active)


def test_rejected_steps_follow_g2o_lambda_control(hiplib, oracle, ctx):
    """Large perturbation: trials with rho < 0 must be rejected and lambda raised by nu, identically on both sides."""
    prob = synth.ba_problem(6, 150, 800, 640, 480, seq_id=46, pose_noise=(0.5, 3.0), point_noise=3.0)
    ba, gp, gx, glog = _compare(hiplib, oracle, ctx, prob, True, 10)
    assert glog["trials"].max() > 1                                 # the fifth iteration needs four trials


def test_all_poses_fixed_and_no_observations(hiplib, oracle, ctx):
    prob = synth.ba_problem(4, 60, 200, 640, 480, seq_id=8)
    prob["fixed"][:] = 1                                            # structure-only: landmarks move, poses do not
    ba, gp, gx, glog = _compare(hiplib, oracle, ctx, prob, True, 5)
    assert np.array_equal(gp, prob["poses"])
    empty = dict(prob); empty["obs_pose"] = prob["obs_pose"][:0]; empty["obs_point"] = prob["obs_point"][:0]
    empty["obs_uvr"] = prob["obs_uvr"][:0]; empty["obs_inv_sigma2"] = prob["obs_inv_sigma2"][:0]; empty["fixed"] = np.zeros(4, np.uint8)
    ba = hiplib.BundleAdjuster(ctx, empty["poses"], empty["fixed"], empty["points"], hiplib.ba_obs_array(empty), empty["cam"])
    log = ba.optimize(True, 3)
    assert len(log) >= 1 and log["chi2_after"][-1] == 0.0


def test_local_ba_flow_with_outliers(hiplib, oracle, ctx):
    prob = synth.ba_problem(8, 300, 1800, 640, 480, seq_id=9)
    bad = np.arange(0, len(prob["obs_pose"]), 29)
    prob["obs_uvr"][bad, 0] += 35.0
    obs = oracle.ba_obs(prob)
    op, ox, oout = oracle.ba_local(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"], 5, 10)
    ba = hiplib.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hiplib.ba_obs_array(prob), prob["cam"])
    gout = ba.local(5, 10)
    gp, gx = ba.state()
    assert np.array_equal(gout, oout) and gout[bad].mean() > 0.9
    assert rot_err(gp[:, :4], op[:, :4]).max() < ROT_TOL and np.abs(gp[:, 4:] - op[:, 4:]).max() < TRANS_TOL
    gchi, gpos = ba.chi2()
    ochi, opos = oracle.ba_chi2(op, ox, obs, prob["cam"])
    assert np.allclose(gchi, ochi, rtol=1e-6, atol=1e-9) and np.array_equal(gpos, opos)


def test_baseline_config3_full_size(hiplib, oracle):
    """BASELINE configs[2]: 50 keyframes / 5000 landmarks / ~40k observations, 10 LM iterations."""
    c = hiplib.Context(1280, 720, 2000, 1.2, 8, max_images=1)
    prob = synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0, top_up=True)
    assert abs(len(prob["obs_pose"]) - 40000) <= 0.02 * 40000            # 40 000 +- 2 %
    ba, gp, gx, glog = _compare(hiplib, oracle, c, prob, True, 10)
    assert ba.solver() == ("dense", -1)
    assert glog["chi2_after"][-1] < 0.2 * glog["chi2_before"][0]
    err0 = np.abs(prob["poses"][:, 4:] - prob["poses_gt"][:, 4:]).max()
    assert np.abs(gp[:, 4:] - prob["poses_gt"][:, 4:]).max() < 0.5 * err0     # converging to the truth
    # determinism: fixed-order reductions give the same bytes on a second run
    ba.reset(); ba.optimize(True, 10)
    gp2, gx2 = ba.state()
    assert np.array_equal(gp, gp2) and np.array_equal(gx, gx2)


def test_contiguous_tracks_take_the_band_path(hiplib, oracle, ctx):
    """Windows as a tracker produces them (synth tracks="contiguous": a landmark is seen by a run of neighbouring keyframes): the
    reduced system is block-banded -- the reference's local bundle adjuster solves it with g2o's LinearSolverCSparse --
    and the problem takes the landmark-group Schur complement on the matrix cores + the band Cholesky (ba_band.inl).  chi2 trajectory,
    trial counts and lambda follow the oracle; _compare also solves the same problem through the pair lists / dense chain."""
    shapes = [(4, 60, 240, 3), (9, 300, 1500, 5), (14, 500, 2600, 6), (23, 900, 5400, 7), (37, 2000, 12000, 8)]
    for i, (kf, pts, n_obs, seed) in enumerate(shapes):
        prob = synth.ba_problem(kf, pts, n_obs, 640, 480, seq_id=seed, tracks="contiguous", top_up=bool(i & 1))
        ba, gp, gx, glog = _compare(hiplib, oracle, ctx, prob, True, 8)
        name, hbw = ba.solver()
        assert name == "band" and 0 <= hbw <= 9, (kf, name, hbw)
    # rejected trials on the band path: lambda control identical to the oracle's
    prob = synth.ba_problem(8, 200, 1000, 640, 480, seq_id=41, pose_noise=(0.5, 3.0), point_noise=3.0, tracks="contiguous")
    ba, _, _, glog = _compare(hiplib, oracle, ctx, prob, True, 10)
    assert ba.solver()[0] == "band" and glog["trials"].max() > 1
    # a random-track window of the same size does not qualify and says so
    prob = synth.ba_problem(23, 900, 5400, 640, 480, seq_id=7)
    ba = hiplib.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hiplib.ba_obs_array(prob), prob["cam"])
    assert ba.solver() == ("dense", -1)
    with pytest.raises(hiplib.LpslamHipError):
        ba.set_solver("band")


def test_band_path_edge_cases(hiplib, oracle, ctx):
    """Fixed keyframes inside the window (their observations shape H_ll but own no rows of S), a landmark seen twice by one keyframe,
    monocular and inactive observations, landmarks seen by fixed keyframes only, caller order shuffled."""
    prob = synth.ba_problem(16, 500, 2800, 640, 480, seq_id=21, tracks="contiguous")
    prob["fixed"][[0, 5, 6, 11]] = 1
    prob["obs_uvr"][::4, 2] = -1.0
    rng = np.random.default_rng(11)
    dup = rng.choice(len(prob["obs_pose"]), 30, replace=False)
    for key in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma2"):
        prob[key] = np.concatenate([prob[key], prob[key][dup]])
    prob["obs_uvr"][-30:, :2] += rng.normal(0, 0.3, (30, 2))
    perm = rng.permutation(len(prob["obs_pose"]))
    for key in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma2"):
        prob[key] = prob[key][perm]
    active = np.ones(len(perm), np.uint8); active[::6] = 0
    ba, _, _, _ = _compare(hiplib, oracle, ctx, prob, True, 7, active)
    assert ba.solver()[0] == "band"
    # the local flow (5 + 10 iterations with outlier re-classification in between)
    prob = synth.ba_problem(12, 400, 2400, 640, 480, seq_id=9, tracks="contiguous")
    bad = np.arange(0, len(prob["obs_pose"]), 29)
    prob["obs_uvr"][bad, 0] += 35.0
    obs = oracle.ba_obs(prob)
    op, ox, oout = oracle.ba_local(prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"], 5, 10)
    ba = hiplib.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], hiplib.ba_obs_array(prob), prob["cam"])
    assert ba.solver()[0] == "band"
    gout = ba.local(5, 10)
    gp, gx = ba.state()
    assert np.array_equal(gout, oout) and gout[bad].mean() > 0.9
    assert rot_err(gp[:, :4], op[:, :4]).max() < ROT_TOL and np.abs(gp[:, 4:] - op[:, 4:]).max() < TRANS_TOL


def test_contiguous_config3_full_size_and_batch(hiplib, oracle):
    """BASELINE configs[2] with contiguous tracks: 50 keyframes / 5000 landmarks / 40 000 +- 2 % observations (block half-bandwidth 8),
    10 iterations against the oracle; then sixteen such windows (and one random-track window riding along on the dense chain) as one
    batch: every problem bit-equal to its single solve, twice the same bytes."""
    c = hiplib.Context(1280, 720, 2000, 1.2, 8, max_images=1)
    prob = synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=0, tracks="contiguous", top_up=True)
    assert abs(len(prob["obs_pose"]) - 40000) <= 0.02 * 40000
    ba, gp, gx, glog = _compare(hiplib, oracle, c, prob, True, 10)
    assert ba.solver() == ("band", 8)
    assert glog["chi2_after"][-1] < 0.2 * glog["chi2_before"][0]
    make = lambda pr: hiplib.BundleAdjuster(c, pr["poses"], pr["fixed"], pr["points"], hiplib.ba_obs_array(pr), pr["cam"])
    probs = [synth.ba_problem(50 - (i % 3), 5000 - 100 * i, 40000 - 800 * i, 1280, 720, seq_id=i, tracks="contiguous", top_up=True) for i in range(1, 16)]
    probs.append(synth.ba_problem(30, 1500, 9000, 1280, 720, seq_id=40))
    batch = [make(pr) for pr in probs]
    assert [b.solver()[0] for b in batch] == ["band"] * 15 + ["dense"]
    logs = hiplib.ba_optimize_batch(batch, True, 6)
    for i, (pr, b, lg) in enumerate(zip(probs, batch, logs)):
        one = make(pr)
        wl = one.optimize(True, 6)
        wp, wx = one.state()
        gp2, gx2 = b.state()
        assert wl.tobytes() == lg.tobytes() and np.array_equal(wp, gp2) and np.array_equal(wx, gx2), i
        one.close()
    hiplib.ba_reset_batch(batch)
    logs2 = hiplib.ba_optimize_batch(batch, True, 6)
    assert all(a.tobytes() == b2.tobytes() for a, b2 in zip(logs, logs2))
    # no hand-over between workgroups (the two chains of the band factorisation, the keyframe blocks of the one-launch update) timed out
    assert ba.timeouts() == (0, 0) and all(b.timeouts() == (0, 0) for b in batch)
    c.close()


def test_pose_optimizer_parity(hiplib, oracle, ctx):
    """Motion-only optimisation (optimize::pose_optimizer): 4 rounds x 10 iterations on ONE pose with outlier re-classification."""
    prob = synth.ba_problem(6, 250, 1500, 640, 480, seq_id=12)
    kf = 4
    sel = prob["obs_pose"] == kf
    obs = oracle.ba_obs(prob)[sel].copy()
    obs["pose"] = 0
    bad = np.arange(0, len(obs), 9)
    obs["v"][bad] += 25.0                                           # gross outliers that the rounds must reject
    pts = prob["points_gt"] + np.random.default_rng(5).normal(0, 0.01, prob["points_gt"].shape)
    start = prob["poses"][kf]
    opose, oout, oin = oracle.pose_optimize(start, pts, obs, prob["cam"])
    hobs = np.zeros(len(obs), hiplib.BA_OBS_DTYPE)
    for f in hobs.dtype.names:
        hobs[f] = obs[f]
    ba = hiplib.BundleAdjuster(ctx, start[None, :], np.zeros(1, np.uint8), pts, hobs, prob["cam"])
    gout, gin = ba.pose_optimize()
    gpose, gpts = ba.state()
    assert gin == oin and np.array_equal(gout, oout) and gout[bad].mean() > 0.9
    assert rot_err(gpose[:, :4], opose[None, :4]).max() < ROT_TOL and np.abs(gpose[0, 4:] - opose[4:]).max() < TRANS_TOL
    assert np.array_equal(gpts, pts)                                # landmarks are constants in this mode
    assert np.abs(gpose[0, 4:] - prob["poses_gt"][kf, 4:]).max() < 0.05
    # the tracker's entry point: the same flow as ONE launch (one workgroup, 6x6 system in LDS)
    kpose, kout, kin = hiplib.pose_optimize(ctx, start, pts, hobs, prob["cam"])
    assert kin == oin and np.array_equal(kout, oout.astype(bool))
    assert rot_err(kpose[None, :4], opose[None, :4]).max() < ROT_TOL and np.abs(kpose[4:] - opose[4:]).max() < TRANS_TOL
    # few observations: the flow stops as soon as fewer than 5 inliers remain; none: the pose is returned unchanged
    few = hobs[:6].copy(); few["v"][:3] += 60.0
    fp, fo, fi = hiplib.pose_optimize(ctx, start, pts, few, prob["cam"])
    op2, oo2, oi2 = oracle.pose_optimize(start, pts, _as_oracle_obs(oracle, few), prob["cam"])
    assert fi == oi2 and np.array_equal(fo, oo2.astype(bool))
    ep, eo, ei = hiplib.pose_optimize(ctx, start, pts, hobs[:0], prob["cam"])
    assert ei == 0 and np.array_equal(ep, start)


def _as_oracle_obs(oracle, hobs):
    o = np.zeros(len(hobs), oracle.OBS_DTYPE)
    for f in o.dtype.names:
        o[f] = hobs[f]
    return o


def test_global_ba_size_runs(hiplib, oracle):
    """BASELINE configs[4] problem size on one GPU: 200 keyframes / 30 000 landmarks / ~240 k observations (dim 1194)."""
    c = hiplib.Context(640, 480, 500, 1.2, 4, max_images=1)
    prob = synth.ba_problem(200, 30000, 240000, 1920, 1080, seq_id=2, kf_stride=2)
    assert abs(len(prob["obs_pose"]) - 240000) <= 0.05 * 240000
    ba = hiplib.BundleAdjuster(c, prob["poses"], prob["fixed"], prob["points"], hiplib.ba_obs_array(prob), prob["cam"])
    log = ba.optimize(True, 10)                                   # the 10 LM iterations the configuration states
    op, ox, olog = oracle.ba_optimize(prob["poses"], prob["fixed"], prob["points"], oracle.ba_obs(prob), prob["cam"], True, 10)
    assert len(log) == 10 and np.allclose(log["chi2_before"], olog["chi2_before"], rtol=CHI_RTOL) and np.allclose(log["chi2_after"], olog["chi2_after"], rtol=CHI_RTOL) and np.array_equal(log["trials"], olog["trials"])
    gp, gx = ba.state()
    assert rot_err(gp[:, :4], op[:, :4]).max() < ROT_TOL and np.abs(gp[:, 4:] - op[:, 4:]).max() < TRANS_TOL


def test_optimize_in_two_halves_beside_front_end_work(hiplib, oracle, ctx):
    """optimize_begin / optimize_end give what optimize gives (also with rejected trials, which need extra units after the
    wait), while front-end work is enqueued on the context in between; misuse is refused."""
    prob = synth.ba_problem(6, 150, 800, 640, 480, seq_id=46, pose_noise=(0.5, 3.0), point_noise=3.0)
    obs = hiplib.ba_obs_array(prob)
    one = hiplib.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"])
    want = one.optimize(True, 10); wp, wx = one.state()
    two = hiplib.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"])
    img = synth.StereoSequence(320, 240, 1).frame(0)[0]
    ctx.upload(0, img)
    for _ in range(3):                      # the third round replays the captured graph
        two.reset()
        two.optimize_begin(True, 10)
        with pytest.raises(hiplib.LpslamHipError):
            two.optimize_begin(True, 10)
        ctx.extract(1)
        got = two.optimize_end()
        gp, gx = two.state()
        assert got.tobytes() == want.tobytes() and np.array_equal(gp, wp) and np.array_equal(gx, wx)
    with pytest.raises(hiplib.LpslamHipError):
        two.optimize_end()


def test_launch_graph_is_shared_by_different_problems_on_a_recycled_stream(hiplib, oracle):
    """The captured launch chain belongs to the STREAM (its kernels read the view from a per-stream slot, launch extents are
    rounded up): a mapping thread's NEW window replays the graph the previous window captured.  Several distinct problems whose
    extents round to one signature -- different observation counts, different landmarks, one group with rejected trials -- are
    created, solved and destroyed in turn (the stream is recycled), and must give, bit for bit, what direct launches give: the
    reference context keeps every problem alive, so each sits on a stream of its own and runs direct (a signature is captured the
    second time a stream sees it)."""
    groups = [[synth.ba_problem(12, 600, n, 640, 480, seq_id=sid) for n, sid in ((4000, 51), (3960, 52), (4060, 53), (4000, 54))],
              [synth.ba_problem(6, 150, n, 640, 480, seq_id=sid, pose_noise=(0.5, 3.0), point_noise=3.0) for n, sid in ((800, 46), (790, 47), (810, 46), (800, 48))]]
    iters = 8
    ref_ctx = hiplib.Context(320, 240, 400, 1.2, 4, max_images=1)
    g_ctx = hiplib.Context(320, 240, 400, 1.2, 4, max_images=1)
    rejected = 0
    for probs in groups:
        alive, want = [], []
        for prob in probs:
            ba = hiplib.BundleAdjuster(ref_ctx, prob["poses"], prob["fixed"], prob["points"], hiplib.ba_obs_array(prob), prob["cam"])
            alive.append(ba)
            log = ba.optimize(True, iters)
            want.append((log, ) + ba.state())
            rejected += int((log["trials"] > 1).sum())
        assert ref_ctx.ba_graph_replays() == 0
        before = g_ctx.ba_graph_replays()
        for prob, (wlog, wp, wx) in zip(probs, want):
            ba = hiplib.BundleAdjuster(g_ctx, prob["poses"], prob["fixed"], prob["points"], hiplib.ba_obs_array(prob), prob["cam"])
            log = ba.optimize(True, iters)
            gp, gx = ba.state()
            ba.close()                                      # hands the stream back: the next problem runs on it
            assert log.tobytes() == wlog.tobytes() and np.array_equal(gp, wp) and np.array_equal(gx, wx)
        # the first problem runs direct, the second captures (and replays), the others replay
        assert g_ctx.ba_graph_replays() - before == len(probs) - 1, "the problems of a group were meant to share one launch signature"
        for ba in alive:
            ba.close()
    assert rejected > 0, "one group is meant to contain rejected trials"
    # one of them against the oracle as well
    prob = groups[1][0]
    op, ox, olog = oracle.ba_optimize(prob["poses"], prob["fixed"], prob["points"], oracle.ba_obs(prob), prob["cam"], True, iters)
    assert np.allclose(want[0][0]["chi2_after"], olog["chi2_after"], rtol=CHI_RTOL) and np.array_equal(want[0][0]["trials"], olog["trials"])
    ref_ctx.close(); g_ctx.close()


def test_set_state_hands_a_prebuilt_window_its_values(hiplib, oracle, ctx):
    """The mapping pipeline: the next window's structure is built (asynchronously) from placeholder values while the previous
    window is being solved, then lpslam_hip_ba_set_state hands it the real poses / landmarks -- bit for bit what creating the
    problem with those values gives.  Refused while a solve is in flight; None keeps a part."""
    prob = synth.ba_problem(6, 150, 800, 640, 480, seq_id=46, pose_noise=(0.5, 3.0), point_noise=3.0)
    obs = hiplib.ba_obs_array(prob)
    direct = hiplib.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"])
    prev = hiplib.BundleAdjuster(ctx, prob["poses"], prob["fixed"], prob["points"], obs, prob["cam"])
    prev.optimize_begin(True, 10)           # "the previous window" in flight while the next one is created
    junk_poses = prob["poses"].copy(); junk_poses[:, 4:] += 0.3
    late = hiplib.BundleAdjuster(ctx, junk_poses, prob["fixed"], prob["points"] + 0.5, obs, prob["cam"])
    with pytest.raises(hiplib.LpslamHipError):
        prev.set_state(prob["poses"], prob["points"])
    prev.optimize_end()
    late.set_state(prob["poses"], prob["points"])
    want = direct.optimize(True, 10); wp, wx = direct.state()
    got = late.optimize(True, 10); gp, gx = late.state()
    assert got.tobytes() == want.tobytes() and np.array_equal(gp, wp) and np.array_equal(gx, wx)
    late.set_state(None, prob["points"])    # poses kept (the creation-time ones set above), LM state cleared
    again = late.optimize(True, 10); ap, ax = late.state()
    assert again.tobytes() == want.tobytes() and np.array_equal(ap, wp) and np.array_equal(ax, wx)


def test_batched_solve_equals_single_solves(hiplib, oracle, ctx):
    """lpslam_hip_ba_optimize_batch: problems of different sizes (different panel counts, one with rejected trials, one with all
    poses fixed) advanced by one launch chain give the bytes the single-problem calls give, and follow the oracle."""
    specs = [dict(n_kf=12, n_pts=600, n_obs=4000, seq=21), dict(n_kf=5, n_pts=80, n_obs=320, seq=22), dict(n_kf=33, n_pts=900, n_obs=6000, seq=23),
             dict(n_kf=6, n_pts=150, n_obs=800, seq=46, pose_noise=(0.5, 3.0), point_noise=3.0), dict(n_kf=2, n_pts=20, n_obs=40, seq=25),
             dict(n_kf=4, n_pts=60, n_obs=200, seq=8, all_fixed=True)]
    probs = []
    for sp in specs:
        kw = {k: sp[k] for k in ("pose_noise", "point_noise") if k in sp}
        pr = synth.ba_problem(sp["n_kf"], sp["n_pts"], sp["n_obs"], 640, 480, seq_id=sp["seq"], **kw)
        if sp.get("all_fixed"):
            pr["fixed"][:] = 1
        probs.append(pr)
    iters = 8
    make = lambda pr: hiplib.BundleAdjuster(ctx, pr["poses"], pr["fixed"], pr["points"], hiplib.ba_obs_array(pr), pr["cam"])
    singles = [make(pr) for pr in probs]
    want = [(b.optimize(True, iters), b.state()) for b in singles]
    batch = [make(pr) for pr in probs]
    logs = hiplib.ba_optimize_batch(batch, True, iters)
    for pr, b, lg, (wl, (wp, wx)) in zip(probs, batch, logs, want):
        gp, gx = b.state()
        assert lg.tobytes() == wl.tobytes() and np.array_equal(gp, wp) and np.array_equal(gx, wx)
        op, ox, olog = oracle.ba_optimize(pr["poses"], pr["fixed"], pr["points"], oracle.ba_obs(pr), pr["cam"], True, iters)
        assert len(lg) == len(olog) and np.allclose(lg["chi2_after"], olog["chi2_after"], rtol=CHI_RTOL) and np.array_equal(lg["trials"], olog["trials"])
        assert rot_err(gp[:, :4], op[:, :4]).max() < ROT_TOL and np.abs(gp[:, 4:] - op[:, 4:]).max() < TRANS_TOL
    assert max(l["trials"].max() for l in logs) > 1                   # the batch did contain rejected trials
    # a second round on the same objects after a batched reset: same bytes again
    hiplib.ba_reset_batch(batch)
    logs2 = hiplib.ba_optimize_batch(batch, True, iters)
    for lg, lg2 in zip(logs, logs2):
        assert lg.tobytes() == lg2.tobytes()
    # misuse: the same problem twice, an empty batch
    with pytest.raises(hiplib.LpslamHipError):
        hiplib.ba_optimize_batch([batch[0], batch[0]], True, 2)


def test_structure_built_on_the_device_handles_any_observation_order(hiplib, oracle, ctx):
    """The device-side structure phase (storage order, CSR, pair lists) must not depend on the order the caller lists the
    observations in, and must cope with a landmark seen twice by one keyframe (duplicate (keyframe, landmark) pairs)."""
    prob = synth.ba_problem(9, 300, 2000, 640, 480, seq_id=31)
    rng = np.random.default_rng(3)
    # duplicates: a second observation of the same landmark in the same keyframe (slightly different measurement)
    dup = rng.choice(len(prob["obs_pose"]), 25, replace=False)
    for key in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma2"):
        prob[key] = np.concatenate([prob[key], prob[key][dup]])
    prob["obs_uvr"][-25:, :2] += rng.normal(0, 0.3, (25, 2))
    perm = rng.permutation(len(prob["obs_pose"]))
    shuf = dict(prob)
    for key in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma2"):
        shuf[key] = prob[key][perm]
    ba, gp, gx, glog = _compare(hiplib, oracle, ctx, shuf, True, 8)
    # per-observation outputs come back in the caller's order
    gchi, gpos = ba.chi2()
    ochi, opos = oracle.ba_chi2(gp, gx, oracle.ba_obs(shuf), shuf["cam"])
    assert np.allclose(gchi, ochi, rtol=1e-9, atol=1e-12) and np.array_equal(gpos, opos)
    # the activity mask is given in the caller's order too
    active = np.ones(len(perm), np.uint8); active[::7] = 0
    _compare(hiplib, oracle, ctx, shuf, True, 5, active)


def test_large_batch_through_the_single_workgroup_factorisation(hiplib, oracle):
    """From 40 problems on (CW_MIN_BATCH, ba.hip), the reduced systems that fit one compute unit (dim + 1 <= 304) are factored and
    solved by one workgroup each (k_chol_wg: lower triangle in registers, panels in LDS); `ba_wg_factorisations` proves that this
    batch went that way and that a batch below the threshold does not.  Sizes cover one panel (dim 6), an odd tile-row count,
    several panels and the full local window (dim 294); a system too large for it (dim 324) rides in the same batch through
    the panel-pair chain.  Every problem follows the oracle; against its single-problem solve it agrees within rounding."""
    c = hiplib.Context(640, 480, 500, 1.2, 4, max_images=1)
    shapes = [(2, 20, 40), (3, 40, 100), (5, 80, 320), (8, 300, 1800), (12, 600, 4000), (20, 800, 5000), (33, 900, 6000), (50, 2500, 16000), (55, 1500, 9000)]
    probs = []
    for i in range(44):
        kf, pts, obs = shapes[i % len(shapes)]
        kw = dict(pose_noise=(0.5, 3.0), point_noise=3.0) if i == 3 else {}
        probs.append(synth.ba_problem(kf, pts, obs, 640, 480, seq_id=100 + i, **kw))
    iters = 6
    def make(pr):
        b = hiplib.BundleAdjuster(c, pr["poses"], pr["fixed"], pr["points"], hiplib.ba_obs_array(pr), pr["cam"])
        b.set_solver("dense")           # the small windows of this list are banded by their size alone: this test is about the dense factorisations
        return b
    batch = [make(pr) for pr in probs]
    assert c.ba_wg_factorisations() == 0
    logs = hiplib.ba_optimize_batch(batch, True, iters)
    wg_launches = c.ba_wg_factorisations()
    assert wg_launches >= iters, "a batch of %d problems was meant to go through k_chol_wg" % len(batch)
    for i, (pr, b, lg) in enumerate(zip(probs, batch, logs)):
        gp, gx = b.state()
        if i < 12:              # the oracle on one problem of every shape (and the one with rejected trials)
            op, ox, olog = oracle.ba_optimize(pr["poses"], pr["fixed"], pr["points"], oracle.ba_obs(pr), pr["cam"], True, iters)
            assert len(lg) == len(olog) and np.allclose(lg["chi2_after"], olog["chi2_after"], rtol=CHI_RTOL), i
            assert np.array_equal(lg["trials"], olog["trials"]) and np.allclose(lg["lambda"], olog["lambda"], rtol=1e-6)
            assert rot_err(gp[:, :4], op[:, :4]).max() < ROT_TOL and np.abs(gp[:, 4:] - op[:, 4:]).max() < TRANS_TOL and np.abs(gx - ox).max() < TRANS_TOL
        one = make(pr)
        wl = one.optimize(True, iters)
        wp, wx = one.state()
        assert len(wl) == len(lg) and np.allclose(wl["chi2_after"], lg["chi2_after"], rtol=1e-10) and np.array_equal(wl["trials"], lg["trials"])
        assert np.abs(wp - gp).max() < 1e-8 and np.abs(wx - gx).max() < 1e-8
        one.close()
    assert c.ba_wg_factorisations() == wg_launches, "single problems go through the panel-pair chain"
    # determinism of the batched path: a second round gives the same bytes
    hiplib.ba_reset_batch(batch)
    logs2 = hiplib.ba_optimize_batch(batch, True, iters)
    assert all(a.tobytes() == b2.tobytes() for a, b2 in zip(logs, logs2))
    assert c.ba_wg_factorisations() >= 2 * iters
    # a batch below the threshold takes the chain: same results as the single solves, bit for bit
    small = batch[:20]
    hiplib.ba_reset_batch(small)
    before = c.ba_wg_factorisations()
    hiplib.ba_optimize_batch(small, True, iters)
    assert c.ba_wg_factorisations() == before
    c.close()


def test_state_of_a_batch_in_one_call(hiplib, ctx):
    """lpslam_hip_ba_set_state_batch / lpslam_hip_ba_get_batch = the single calls problem by problem"""
    probs = [synth.ba_problem(6 + i, 120, 700, 640, 480, seq_id=20 + i) for i in range(3)]
    bas = [hiplib.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], hiplib.ba_obs_array(p), p["cam"]) for p in probs]
    moved = [(p["poses"] + 1e-3, p["points"] - 2e-3) for p in probs]
    hiplib.ba_set_state_batch(bas, [m[0] for m in moved], [None, moved[1][1], moved[2][1]])
    got = hiplib.ba_state_batch(bas)
    for i, (b, (po, pt)) in enumerate(zip(bas, got)):
        spo, spt = b.state()
        assert np.array_equal(po, spo) and np.array_equal(pt, spt)
        assert np.array_equal(po, moved[i][0]) and np.array_equal(pt, probs[i]["points"] if i == 0 else moved[i][1])
    hiplib.ba_optimize_batch(bas, True, 3)
    after = hiplib.ba_state_batch(bas)
    for b, (po, pt) in zip(bas, after):
        spo, spt = b.state()
        assert np.array_equal(po, spo) and np.array_equal(pt, spt)
        b.close()


@pytest.mark.parametrize("n", [5, 40, 100, 300, 600, 3000])
def test_pose_optimizer_lane_layouts_and_sizes(hiplib, oracle, ctx, n):
    """lpslam_hip_pose_optimize over the kernel's layouts: one observation per quad of lanes (n <= 64), per pair (<= 128), per lane,
    two per lane, and more observations than it keeps in LDS (~2700: they are read from device memory); mono / stereo mix, outliers."""
    rng = np.random.default_rng(n)
    cam = dict(synth.intrinsics(640, 480))
    pts = np.stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(2, 20, n)], axis=1)
    t = np.array([0.03, -0.02, 0.05])
    pc = pts + t
    obs = np.zeros(n, hiplib.BA_OBS_DTYPE)
    obs["point"] = np.arange(n)
    obs["u"] = cam["fx"] * pc[:, 0] / pc[:, 2] + cam["cx"] + rng.normal(0, 0.5, n)
    obs["v"] = cam["fy"] * pc[:, 1] / pc[:, 2] + cam["cy"] + rng.normal(0, 0.5, n)
    stereo = rng.random(n) < 0.6
    obs["ur"] = np.where(stereo, obs["u"] - cam["fxb"] / pc[:, 2] + rng.normal(0, 0.5, n), -1.0)
    obs["inv_sigma2"] = 1.0 / (1.2 ** (2 * rng.integers(0, 4, n)))
    gross = np.arange(0, n, 7)
    obs["v"][gross] += 30.0
    start = np.array([1.0, 0, 0, 0, 0, 0, 0])
    opose, oout, oin = oracle.pose_optimize(start, pts, _as_oracle_obs(oracle, obs), cam)
    kpose, kout, kin = hiplib.pose_optimize(ctx, start, pts, obs, cam)
    assert kin == oin and np.array_equal(kout, oout.astype(bool))
    assert np.abs(kpose[:4] - opose[:4]).max() < 1e-7 and np.abs(kpose[4:] - opose[4:]).max() < 1e-6
    assert 5 <= ctx.pose_optimize_passes() <= 4 * 101           # one pass per round + one per Levenberg trial
    if n >= 40:
        assert kout[gross].mean() > 0.9 and np.abs(kpose[4:] - t).max() < 0.02


def test_batched_build_equals_single_builds(hiplib, oracle, ctx):
    """Creation in two halves: the host half of every window on threads of its own (lpslam_hip_ba_prepare enqueues nothing), the
    device half of all of them as ONE launch chain (lpslam_hip_ba_build_batch, blockIdx.y = problem) -- a server's sessions set their
    windows up that way.  Every problem then solves to the same bytes as a problem created alone, whatever shares the build with it:
    banded and dense windows, a tiny one, one with duplicates and a shuffled caller order, one with fixed keyframes inside.  A
    prepared problem refuses every other call until it has been built."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(11)
    probs = [synth.ba_problem(50, 5000, 40000, 1280, 720, seq_id=1, tracks="contiguous", top_up=True),
             synth.ba_problem(30, 1500, 9000, 1280, 720, seq_id=2),
             synth.ba_problem(3, 40, 100, 640, 480, seq_id=3),
             synth.ba_problem(37, 2500, 17000, 1280, 720, seq_id=4, tracks="contiguous"),
             synth.ba_problem(12, 600, 4000, 640, 480, seq_id=5),
             synth.ba_problem(48, 4400, 36000, 1280, 720, seq_id=6, tracks="contiguous", top_up=True)]
    pr = probs[4]
    m = len(pr["obs_pose"])
    dup = rng.choice(m, 40, replace=False)
    for key in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma2"):
        pr[key] = np.concatenate([pr[key], pr[key][dup]])
    perm = rng.permutation(len(pr["obs_pose"]))
    for key in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma2"):
        pr[key] = pr[key][perm]
    probs[3]["fixed"][[5, 9]] = 1
    make = lambda p, build: hiplib.BundleAdjuster(ctx, p["poses"], p["fixed"], p["points"], hiplib.ba_obs_array(p), p["cam"], build=build)
    with ThreadPoolExecutor(4) as pool:
        batch = list(pool.map(lambda p: make(p, False), probs))
    with pytest.raises(hiplib.LpslamHipError):
        batch[0].optimize(True, 1)
    with pytest.raises(hiplib.LpslamHipError):
        hiplib.ba_optimize_batch(batch, True, 1)
    hiplib.ba_build_batch(batch)
    with pytest.raises(hiplib.LpslamHipError):
        hiplib.ba_build_batch(batch[:1])                     # built already
    assert [b.solver()[0] for b in batch] == ["band", "dense", "band", "band", "dense", "band"]
    logs = hiplib.ba_optimize_batch(batch, True, 6)
    for i, (p, b, lg) in enumerate(zip(probs, batch, logs)):
        one = make(p, True)
        wl = one.optimize(True, 6)
        wp, wx = one.state()
        gp, gx = b.state()
        assert wl.tobytes() == lg.tobytes() and np.array_equal(wp, gp) and np.array_equal(wx, gx), i
        if i == 2:                                           # and the oracle, on the problem whose build shares a launch with much larger ones
            op, ox, olog = oracle.ba_optimize(p["poses"], p["fixed"], p["points"], oracle.ba_obs(p), p["cam"], True, 6)
            assert np.allclose(lg["chi2_after"], olog["chi2_after"], rtol=1e-9) and np.array_equal(lg["trials"], olog["trials"])
        one.close()
    # a batch-built problem is an ordinary problem: solved alone afterwards, from a reset, it gives the batch's bytes again
    batch[0].reset()
    again = batch[0].optimize(True, 6)
    assert again.tobytes() == logs[0].tobytes()
    assert all(b.timeouts() == (0, 0) for b in batch)
    for b in batch:
        b.close()
