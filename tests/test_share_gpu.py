"""Launches shared by the sessions of a process (lpslam_amd/csrc/share.hip): the front end of several sessions' pending frames as
one launch chain over an image list of the session pool, the window matchers' first scans and the pose optimisations of several
sessions as one launch each -- against the same calls made by a lone context: integer results (keypoints, descriptors, stereo
columns, match lists) are equal, the pose optimiser runs the same device code per request and returns the same bits."""
import threading

import numpy as np
import pytest

from lpslam_amd import synth

pytestmark = pytest.mark.gpu


def _frame_of(ctx, slot):
    kp, desc, xr, depth = ctx.frame_view(slot, True)
    return kp.copy(), desc.copy(), xr.copy(), depth.copy()


@pytest.mark.parametrize("with_upload", [False, True])
@pytest.mark.parametrize("size", [(640, 480, 1000, 4), (1280, 720, 2000, 8)])
def test_shared_front_end_equals_the_unshared_one(hiplib, size, with_upload):
    """Four sessions of one pool submit lpslam_hip_front_end from four threads (sharing forced on, so a lone request is a shared launch of
    one too); every session's frame -- keypoints, descriptors, right-image columns, depths -- is what a private context extracts from the
    same images with extract_range + match_stereo + prefetch_frame; three rounds over the rotating slot pairs."""
    w, h, kpts, levels = size
    k = synth.intrinsics(w, h)
    n_sessions, rounds = 4, 3
    seqs = [synth.StereoSequence(w, h, 4 + s) for s in range(n_sessions)]
    images = [[seqs[s].frame(r) for r in range(rounds)] for s in range(n_sessions)]
    ref_ctx = hiplib.Context(w, h, kpts, 1.2, levels, max_images=2)
    ref = {}
    for s in range(n_sessions):
        for r in range(rounds):
            l, rt = images[s][r]
            ref_ctx.upload(0, l); ref_ctx.upload(1, rt)
            ref_ctx.extract_range(0, 2); ref_ctx.match_stereo(0, 1, k["fxb"], k["baseline"]); ref_ctx.prefetch_frame(0, True)
            ref[(s, r)] = _frame_of(ref_ctx, 0)
    ref_ctx.close()
    got, errors = {}, []
    try:
        hiplib.set_shared_launches(1)
        b0, r0 = hiplib.shared_front_end_counters(0)
        sessions = [hiplib.Context(w, h, kpts, 1.2, levels, max_images=6, session=True) for _ in range(n_sessions)]
        barrier = threading.Barrier(n_sessions)

        def run(s):
            try:
                c = sessions[s]
                for r in range(rounds):
                    slot = 2 * (r % 3)
                    l, rt = images[s][r]
                    barrier.wait()
                    if with_upload:                  # lpslam_hip_front_end_images: the chain uploads the frames itself
                        c.front_end_images(slot, l, rt, k["fxb"], k["baseline"])
                    else:
                        c.upload(slot, l); c.upload(slot + 1, rt)
                        c.front_end(slot, True, k["fxb"], k["baseline"])
                    got[(s, r)] = _frame_of(c, slot)
            except BaseException as e:
                errors.append(e)
                barrier.abort()

        th = [threading.Thread(target=run, args=(s,)) for s in range(n_sessions)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        b1, r1 = hiplib.shared_front_end_counters(0)
        for c in sessions:
            c.close()
    finally:
        hiplib.set_shared_launches(None)
    assert not errors, errors
    assert r1 - r0 == n_sessions * rounds and b1 - b0 <= r1 - r0
    for key, (kp, desc, xr, depth) in ref.items():
        gk, gd, gx, gz = got[key]
        assert len(kp) == len(gk) and len(kp) > 100, (key, len(kp), len(gk))
        for f in kp.dtype.names:
            assert np.array_equal(kp[f], gk[f]), (key, f)
        assert np.array_equal(desc, gd) and np.array_equal(xr, gx) and np.array_equal(depth, gz), key
    print("shared front end %dx%d: %d requests in %d launch chains" % (w, h, r1 - r0, b1 - b0))


def test_shared_monocular_front_end_equals_the_unshared_one(hiplib):
    """The monocular form of the shared front end (one slot per frame, no stereo kernels in the chain): three sessions' frames against a
    private context's extract_range + prefetch_frame."""
    w, h, kpts, levels = 640, 480, 1000, 4
    n_sessions, rounds = 3, 3
    seqs = [synth.StereoSequence(w, h, 9 + s) for s in range(n_sessions)]
    images = [[seqs[s].frame(r)[0] for r in range(rounds)] for s in range(n_sessions)]
    ref_ctx = hiplib.Context(w, h, kpts, 1.2, levels, max_images=2)
    ref = {}
    for s in range(n_sessions):
        for r in range(rounds):
            ref_ctx.upload(0, images[s][r]); ref_ctx.extract_range(0, 1); ref_ctx.prefetch_frame(0, False)
            kp, desc, _, _ = ref_ctx.frame_view(0, False)
            ref[(s, r)] = (kp.copy(), desc.copy())
    ref_ctx.close()
    got, errors = {}, []
    try:
        hiplib.set_shared_launches(1)
        sessions = [hiplib.Context(w, h, kpts, 1.2, levels, max_images=6, session=True) for _ in range(n_sessions)]
        barrier = threading.Barrier(n_sessions)

        def run(s):
            try:
                for r in range(rounds):
                    slot = 2 * (r % 3)
                    barrier.wait()
                    sessions[s].front_end_images(slot, images[s][r])
                    kp, desc, _, _ = sessions[s].frame_view(slot, False)
                    got[(s, r)] = (kp.copy(), desc.copy())
            except BaseException as e:
                errors.append(e); barrier.abort()

        th = [threading.Thread(target=run, args=(s,)) for s in range(n_sessions)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for c in sessions:
            c.close()
    finally:
        hiplib.set_shared_launches(None)
    assert not errors, errors
    for key, (kp, desc) in ref.items():
        gk, gd = got[key]
        assert len(kp) == len(gk) and len(kp) > 100
        for f in kp.dtype.names:
            assert np.array_equal(kp[f], gk[f]), (key, f)
        assert np.array_equal(desc, gd), key


def test_shared_pose_optimiser_matchers_and_windows_return_the_unshared_bits(hiplib):
    """Four sessions, four threads, sharing forced on: lpslam_hip_pose_optimize on different observation sets (one, two and four
    wavefronts), lpslam_hip_match_projection on different query sets and lpslam_hip_ba_local_window on different windows, several times
    each, against the same calls with sharing off -- poses, outlier masks, inlier counts, match lists, solved windows: the same bits.
    The counters show that shared launches carried the requests."""
    w, h, kpts, levels = 640, 480, 1000, 4
    n_sessions, reps = 4, 6
    k = synth.intrinsics(w, h)
    rng = np.random.default_rng(7)
    # pose problems of the three kernel classes
    pose_cases = []
    for s in range(n_sessions):
        n_obs = (40, 100, 200, 64)[s]
        prob = synth.ba_problem(2, n_obs, 2 * n_obs, w, h, seq_id=30 + s)
        sel = prob["obs_pose"] == 1
        pose_cases.append((prob["poses"][1].copy(), prob["points"], hiplib.ba_obs_array(prob)[sel], prob["cam"]))
    windows = [synth.ba_problem(6 + s, 150 + 20 * s, 700 + 100 * s, w, h, seq_id=50 + s, tracks="contiguous", top_up=True) for s in range(n_sessions)]
    seq = synth.StereoSequence(w, h, 4, n_points=6000)
    frames = [seq.frame(i) for i in range(n_sessions)]

    def work(c, s, out):
        l, r = frames[s]
        c.upload(0, l); c.upload(1, r)
        c.extract_range(0, 2); c.match_stereo(0, 1, k["fxb"], k["baseline"])
        kp, desc = c.keypoints(0)
        nq = min(300, len(kp))
        q = np.zeros(nq, hiplib.PROJ_QUERY_DTYPE)
        q["x"] = kp["x"][:nq] + 1.5; q["y"] = kp["y"][:nq] - 1.0; q["x_right"] = -1.0; q["radius"] = 12.0; q["min_level"] = -1; q["max_level"] = -1
        for rep in range(reps):
            out.append(("pose", s, rep) + tuple(np.asarray(x).tobytes() if hasattr(x, "tobytes") else x for x in hiplib.pose_optimize(c, *pose_cases[s])))
            idx, dist, nm = c.match_projection(0, q, desc[:nq], 100, 0.9)
            out.append(("match", s, rep, idx.tobytes(), dist.tobytes(), nm))
        p = windows[s]
        po, pt, outl = hiplib.ba_local_window(c, p["poses"], p["fixed"], p["points"], hiplib.ba_obs_array(p), p["cam"])
        out.append(("window", s, po.tobytes(), pt.tobytes(), outl.tobytes()))

    def run_all(mode):
        hiplib.set_shared_launches(mode)
        ctxs = [hiplib.Context(w, h, kpts, 1.2, levels, max_images=6, session=True) for _ in range(n_sessions)]
        outs, errors = [[] for _ in range(n_sessions)], []

        def run(s):
            try:
                work(ctxs[s], s, outs[s])
            except BaseException as e:
                errors.append(e)
        th = [threading.Thread(target=run, args=(s,)) for s in range(n_sessions)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for c in ctxs:
            c.close()
        assert not errors, errors
        return outs

    try:
        ref = run_all(0)
        b0, r0 = hiplib.shared_launch_counters(0); sb0, sr0 = hiplib.shared_solve_counters(0)
        got = run_all(1)
        b1, r1 = hiplib.shared_launch_counters(0); sb1, sr1 = hiplib.shared_solve_counters(0)
    finally:
        hiplib.set_shared_launches(None)
    assert r1 - r0 >= 2 * n_sessions * reps                      # every pose optimisation and matcher scan was a request of a shared launch
    assert sr1 - sr0 == n_sessions - 1 or sr1 - sr0 == n_sessions  # the windows of the sessions on the role streams (the pool's first session keeps its own stream)
    for s in range(n_sessions):
        assert len(ref[s]) == len(got[s])
        for a, b in zip(ref[s], got[s]):
            assert a == b, (a[:3], "differs between the unshared and the shared run")


@pytest.mark.parametrize("env_extra", [{"GPU_MAX_HW_QUEUES": "2"}, {"LPSLAM_HIP_SHARE_PRIO": "1"}, {"LPSLAM_HIP_SHARE_NO_PROBE": "1"}],
                         ids=["two_hardware_queues", "two_priority_levels", "no_probe"])
def test_role_streams_when_queues_are_short_or_prioritised(env_extra):
    """share_init's other ways of choosing the role streams -- fewer independent hardware queues than roles (roles share a queue), the
    latency-bound roles on high-priority streams with an auxiliary role beside them, no probing at all -- are process-wide choices made
    once, so each runs in a process of its own: the shared pose optimiser / matcher / window test above must hold there too."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from lpslam_amd import _build, hip\n_build.hip_library(); hip.load()\n"
            "import test_share_gpu as t\n"
            "t.test_shared_pose_optimiser_matchers_and_windows_return_the_unshared_bits(hip)\n"
            "print('VARIANT-OK')\n") % (root, os.path.join(root, "tests"))
    env = dict(os.environ); env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "VARIANT-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
