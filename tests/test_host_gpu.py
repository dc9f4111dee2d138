"""GPU test of the drop-in boundary end to end: stereo frames in through the manager, poses out through the callback."""
import time

import numpy as np
import pytest

from lpslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("async_mapping", [True, False])
def test_stereo_sequence_through_the_manager(hiplib, async_mapping):
    """Local BA beside tracking (the default, as the reference's mapping thread) and inline: both track the sequence."""
    from lpslam_amd import _build, manager
    _build.host_library()
    w, h, n_frames = 640, 480, 24
    k = synth.intrinsics(w, h)
    seq = synth.StereoSequence(w, h, 4, n_points=6000)
    m = manager.Manager()
    for num in (0, 1):
        c = manager.default_camera()
        c.camera_number = num; c.f_x = k["fx"]; c.f_y = k["fy"]; c.c_x = k["cx"]; c.c_y = k["cy"]
        c.resolution_x = w; c.resolution_y = h; c.focal_x_baseline = k["fxb"]
        m.set_camera(c)
    assert m.add_tracker("VSLAMStereo", '{"cameraSetup": "stereo", "slamKeypoints": 1000, "numLevels": 4, "keyframeInterval": 4, "asyncMapping": %s}'
                         % ("true" if async_mapping else "false"))
    m.collect_results(); m.provide_odometry()
    import ctypes
    counter = ctypes.CDLL(_build.host_library()).lpslam_debug_motion_tracked
    counter.restype = ctypes.c_long
    tracked0 = counter()
    m.start()
    frames = [seq.frame(i) for i in range(n_frames)]
    for i, (l, r) in enumerate(frames):
        assert m.add_stereo((i + 1) * 40_000_000, l, r)
    t0 = time.time()
    while len(m.results) < n_frames and time.time() - t0 < 60:
        time.sleep(0.01)
    st = m.status()
    feats = m.features()
    m.stop()
    assert len(m.results) == n_frames
    assert counter() - tracked0 >= n_frames - 6          # constant-velocity prediction + projection matching carried most frames
    valid = [r for r in m.results if r["valid"]]
    assert len(valid) >= n_frames - 2 and st.localization == 2 and st.key_frames >= 3 and st.feature_points > 100
    assert len(feats) == st.feature_points
    # ground truth: camera centre = (0, 0, 0.05 k) in optical axes -> lpslam axes (-y, x, z); first frame is the origin
    last = valid[-1]
    kf = n_frames - 1
    # (the synthetic patches sit on integer pixels, so disparities are quantised: allow 10 % scale error)
    assert abs(last["p"][2] - 0.05 * kf) < 0.1 * 0.05 * kf and abs(last["p"][0]) < 0.05 and abs(last["p"][1]) < 0.05
    zs = [r["p"][2] for r in valid]
    assert all(b > a - 0.01 for a, b in zip(zs, zs[1:]))           # moving forward
    assert abs(last["q"][0]) > 0.999                                # yaw stays below 0.2 degrees


def test_replay_file_through_the_manager(hiplib, tmp_path):
    """A recording in the reference's stream format (src/Serialize/ProtoStream.h, SlamSerialize.proto; images as PGM) is read
    by LpSlamManager::readReplayItems and tracked; replayed frames carry no ROS stamp, hence "require_odometry": false."""
    import json
    import replay_format as rf
    from lpslam_amd import _build, manager
    _build.host_library()
    w, h, n_frames = 640, 480, 10
    k = synth.intrinsics(w, h)
    seq = synth.StereoSequence(w, h, 6, n_points=6000)
    stream = b"".join(rf.record(rf.CAMERA_IMAGE, rf.camera_image((i + 1) * 40_000_000, *seq.frame(i), cam=0, data_number=i)) for i in range(n_frames))
    rec = tmp_path / "drive.pb"; rec.write_bytes(stream)
    cfg = {"manager": {"require_odometry": False}}
    cfg_path = tmp_path / "replay.json"; cfg_path.write_text(json.dumps(cfg))
    m = manager.Manager()
    assert m.read_configuration_file(str(cfg_path))
    for num in (0, 1):
        c = manager.default_camera()
        c.camera_number = num; c.f_x = k["fx"]; c.f_y = k["fy"]; c.c_x = k["cx"]; c.c_y = k["cy"]
        c.resolution_x = w; c.resolution_y = h; c.focal_x_baseline = k["fxb"]
        m.set_camera(c)
    assert m.add_tracker("VSLAMStereo", '{"cameraSetup": "stereo", "slamKeypoints": 1000, "numLevels": 4, "keyframeInterval": 4}')
    m.collect_results()
    m.start()
    assert m.read_replay_items(rec) and not m.read_replay_items(tmp_path / "nope.pb")
    t0 = time.time()
    while len(m.results) < n_frames and time.time() - t0 < 60:
        time.sleep(0.01)
    st = m.status()
    m.stop()
    assert len(m.results) == n_frames
    valid = [r for r in m.results if r["valid"]]
    assert len(valid) >= n_frames - 2 and st.key_frames >= 2
    assert [r["timestamp"] for r in m.results] == [(i + 1) * 40_000_000 for i in range(n_frames)]


def test_local_map_tracking_brings_landmarks_back(hiplib):
    """One frame with its left half blanked: the frame after it cannot get those landmarks from the motion model (the previous
    frame does not hold them), local-map tracking projects them from the keyframes and matches them again
    ([UPSTREAM] tracking_module::optimize_current_frame_with_local_map)."""
    import ctypes
    from lpslam_amd import _build, manager
    lib = ctypes.CDLL(_build.host_library())
    lib.lpslam_debug_local_map_joined.restype = ctypes.c_long
    w, h = 640, 480
    k = synth.intrinsics(w, h)
    seq = synth.StereoSequence(w, h, 4, n_points=6000)
    m = manager.Manager()
    for num in (0, 1):
        c = manager.default_camera()
        c.camera_number = num; c.f_x = k["fx"]; c.f_y = k["fy"]; c.c_x = k["cx"]; c.c_y = k["cy"]
        c.resolution_x = w; c.resolution_y = h; c.focal_x_baseline = k["fxb"]
        m.set_camera(c)
    assert m.add_tracker("VSLAMStereo", '{"cameraSetup": "stereo", "slamKeypoints": 1000, "numLevels": 4, "keyframeInterval": 8}')
    m.collect_results(); m.provide_odometry()
    joined0 = lib.lpslam_debug_local_map_joined()
    m.start()
    frames = [list(seq.frame(i)) for i in range(6)]
    for eye in (0, 1):
        frames[3][eye] = frames[3][eye].copy(); frames[3][eye][:, : w // 2] = 0
    for i, (l, r) in enumerate(frames):
        assert m.add_stereo((i + 1) * 40_000_000, l, r)
    t0 = time.time()
    while len(m.results) < len(frames) and time.time() - t0 < 60:
        time.sleep(0.01)
    m.stop()
    assert len(m.results) == len(frames) and all(r["valid"] for r in m.results)
    assert lib.lpslam_debug_local_map_joined() - joined0 >= 20
    assert abs(m.results[-1]["p"][2] - 0.05 * 5) < 0.05


def test_monocular_sequence_initialises_and_tracks(hiplib):
    """VSLAMMono through the manager: two-view initialisation ([UPSTREAM] initialize::perspective) on a sideways-moving camera,
    then motion-model / local-map tracking and keyframes whose new landmarks are triangulated against the previous keyframe.
    The map's scale is arbitrary (median depth 1), so the trajectory is checked in direction and proportion."""
    from lpslam_amd import _build, manager
    _build.host_library()
    w, h, n_frames = 640, 480, 30
    k = synth.intrinsics(w, h)
    seq = synth.WallSequence(w, h, 11)                   # three textured walls at 14 / 9 / 6 m, camera moving sideways
    centres = [seq.centre(i) for i in range(n_frames)]
    frames = [seq.frame(i) for i in range(n_frames)]
    m = manager.Manager()
    c = manager.default_camera()
    c.camera_number = 0; c.f_x = k["fx"]; c.f_y = k["fy"]; c.c_x = k["cx"]; c.c_y = k["cy"]; c.resolution_x = w; c.resolution_y = h
    m.set_camera(c)
    assert m.add_tracker("VSLAMMono", '{"cameraSetup": "monocular", "slamKeypoints": 2000, "numLevels": 3, "keyframeInterval": 4}')
    m.collect_results(); m.provide_odometry()
    m.start()
    for i, img in enumerate(frames):
        assert m.add_image((i + 1) * 40_000_000, img)
    t0 = time.time()
    while len(m.results) < n_frames and time.time() - t0 < 60:
        time.sleep(0.01)
    st = m.status()
    m.stop()
    valid = [(i, r) for i, r in enumerate(m.results) if r["valid"]]
    assert len(m.results) == n_frames and len(valid) >= n_frames - 12          # a few frames pass before the parallax suffices
    assert st.localization == 2 and st.key_frames >= 4 and st.feature_points > 150
    # lpslam axes: p_lp = (-y, x, z) of the optical-frame camera centre; the first valid pose defines origin and scale
    i0, r0 = valid[0]; i1, r1 = valid[-1]
    d = np.array(r1["p"]) - np.array(r0["p"])
    truth = centres[i1] - centres[i0]
    truth_lp = np.array([-truth[1], truth[0], truth[2]])
    cosang = d @ truth_lp / (np.linalg.norm(d) * np.linalg.norm(truth_lp))
    assert cosang > 0.995                                                         # direction of travel within ~6 degrees
    mid_i, mid_r = valid[len(valid) // 2]
    frac = np.linalg.norm(np.array(mid_r["p"]) - np.array(r0["p"])) / np.linalg.norm(d)
    frac_true = np.linalg.norm(centres[mid_i] - centres[i0]) / np.linalg.norm(truth)
    assert abs(frac - frac_true) < 0.1                                            # constant speed: no scale jump along the way


def test_loop_is_detected_and_closed(hiplib):
    """Out and back along the same line: when the camera returns, archived keyframes near the current position are matched by
    descriptor on the device, verified with the Sim3 optimiser of the loop detector and the loop is closed by the Sim3 pose graph
    ([UPSTREAM] module::loop_detector / optimize::transform_optimizer / optimize::graph_optimizer; the reference switches the
    detector with the tracker's loopClosure key, src/Trackers/OpenVSLAMTrackerBase.cpp:250-255).  The tracker barely drifts on
    this sequence, so closing the loop must leave the trajectory where it was."""
    import ctypes
    from lpslam_amd import _build, manager
    lib = ctypes.CDLL(_build.host_library())
    lib.lpslam_debug_loops_closed.restype = ctypes.c_long
    w, h = 640, 480
    k = synth.intrinsics(w, h)
    seq = synth.StereoSequence(w, h, 4, n_points=6000)
    zs = [0.05 * i for i in range(25)] + [0.05 * (24 - i) for i in range(1, 25)]
    frames = []
    for i, z in enumerate(zs):
        rng = np.random.Generator(np.random.PCG64([5, i]))
        t = -np.array([0.0, 0.0, z])
        frames.append((seq._render(np.eye(3), t, rng), seq._render(np.eye(3), t - np.array([k["baseline"], 0.0, 0.0]), rng)))
    m = manager.Manager()
    for num in (0, 1):
        c = manager.default_camera()
        c.camera_number = num; c.f_x = k["fx"]; c.f_y = k["fy"]; c.c_x = k["cx"]; c.c_y = k["cy"]
        c.resolution_x = w; c.resolution_y = h; c.focal_x_baseline = k["fxb"]
        m.set_camera(c)
    assert m.add_tracker("VSLAMStereo", '{"cameraSetup": "stereo", "slamKeypoints": 1000, "numLevels": 4, "keyframeInterval": 3, "localWindow": 4, "loopClosure": true}')
    m.collect_results(); m.provide_odometry()
    closed0 = lib.lpslam_debug_loops_closed()
    m.start()
    for i, (l, r) in enumerate(frames):
        assert m.add_stereo((i + 1) * 40_000_000, l, r)
    t0 = time.time()
    while len(m.results) < len(frames) and time.time() - t0 < 60:
        time.sleep(0.01)
    m.stop()
    assert len(m.results) == len(frames) and sum(r["valid"] for r in m.results) >= len(frames) - 2
    assert lib.lpslam_debug_loops_closed() - closed0 >= 1
    valid = [(i, r) for i, r in enumerate(m.results) if r["valid"]]
    for i, r in valid[5:]:
        assert abs(r["p"][2] - zs[i]) < 0.12 * max(zs[i], 0.25) + 0.02 and abs(r["p"][0]) < 0.06 and abs(r["p"][1]) < 0.06, (i, r["p"], zs[i])
