"""Closed-loop parity of the stereo tracker: the product path (LpSlamManager -> VSLAMStereo tracker -> HIP kernels) against the
closed-loop oracle (oracle/tracker.py, golden tests/golden/g10_track.npz made by tools/make_golden_track.py) on a 24-frame
640x480 sequence, pose by pose.  Tolerance: the north star's 1e-4 rad / 1e-3 m, per tracked frame."""
import hashlib
import math
import time

import numpy as np
import pytest

from conftest import golden
from lpslam_amd import synth

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-4, 1e-3
W, H = 640, 480
TRACKER = '{"cameraSetup": "stereo", "slamKeypoints": 1000, "numLevels": 4, "keyframeInterval": 4, "localWindow": 10, "asyncMapping": false, "loopClosure": false}'


def _to_result(pose7):
    """createTrackerResult (src/Trackers/OpenVSLAMTrackerBase.cpp:307-329) of a world -> camera pose: position and orientation in
    lpslam axes, as the manager's callback delivers them"""
    from oracle import tracker as T
    R = T.quat_to_rot(pose7[:4])
    t = pose7[4:]
    C = -(R.T @ t)
    q = T.rot_to_quat(R)
    return np.array([-C[1], C[0], C[2]]), np.array([q[0], -q[2], q[1], q[3]])


def _run_product(frames, tmp_path):
    from lpslam_amd import _build, manager
    _build.host_library()
    k = synth.intrinsics(W, H)
    m = manager.Manager()
    for num in (0, 1):
        c = manager.default_camera()
        c.camera_number = num; c.f_x = k["fx"]; c.f_y = k["fy"]; c.c_x = k["cx"]; c.c_y = k["cy"]
        c.resolution_x = W; c.resolution_y = H; c.focal_x_baseline = k["fxb"]
        m.set_camera(c)
    assert m.add_tracker("VSLAMStereo", TRACKER)
    m.collect_results(); m.provide_odometry()
    log = tmp_path / "slam.log"
    m.log_to_file(log)
    m.start()
    for i, (l, r) in enumerate(frames):
        assert m.add_stereo((i + 1) * 40_000_000, l, r)
    t0 = time.time()
    while len(m.results) < len(frames) and time.time() - t0 < 60:
        time.sleep(0.01)
    m.stop()
    return m.results, manager.Manager.statistics(log)


def test_tracked_poses_follow_the_closed_loop_oracle(hiplib, tmp_path):
    g = golden("g10_track.npz")
    n = int(g["frames"])
    seq = synth.StereoSequence(W, H, 4, n_points=6000)
    frames = [seq.frame(i) for i in range(n)]
    sha = hashlib.sha256()
    for l, r in frames:
        sha.update(l.tobytes()); sha.update(r.tobytes())
    assert sha.hexdigest() == str(g["sha"])                               # the committed generator still makes the golden's images
    results, stats = _run_product(frames, tmp_path)
    assert len(results) == n and all(r["valid"] for r in results)
    worst_rot, worst_pos = 0.0, 0.0
    for i, r in enumerate(results):
        p, q = _to_result(g["poses"][i])
        dq = abs(float(np.dot(q / np.linalg.norm(q), np.array(r["q"]) / np.linalg.norm(r["q"]))))
        ang = 2 * math.acos(min(1.0, dq))
        dp = float(np.abs(p - np.array(r["p"])).max())
        worst_rot, worst_pos = max(worst_rot, ang), max(worst_pos, dp)
        assert ang < ROT_TOL and dp < TRANS_TOL, (i, ang, dp)
    # the same discrete history: keyframes, motion-model frames, local BA runs, fused duplicates
    for key in ("keyframes", "motion_tracked", "bf_tracked", "local_map_joined", "fused_added", "fused_merged", "local_ba"):
        assert stats[key] == int(g["stat_" + key]), (key, stats[key], int(g["stat_" + key]))
    print("closed loop: worst rotation %.2e rad, worst position %.2e m over %d frames" % (worst_rot, worst_pos, n))
