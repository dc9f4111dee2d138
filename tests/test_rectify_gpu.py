"""GPU tests of the on-device undistort / rectify (k_remap behind lpslam_hip_set_rectify_map / lpslam_hip_upload_raw_image):
bit-exact against the cv::remap restatement, and the stereo tracker fed with raw (distorted) frames."""
import time

import numpy as np
import pytest

from lpslam_amd import synth

pytestmark = pytest.mark.gpu

W, H = 320, 240


@pytest.fixture(scope="module")
def rect():
    from oracle import rectify
    return rectify


@pytest.fixture(scope="module")
def ctx(hiplib):
    return hiplib.Context(W, H, 300, 1.2, 4, max_images=2)


def test_remap_bit_exact_random_maps(hiplib, rect, ctx):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (H, W)).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    cases = {
        "identity": (xx, yy),
        "jitter": (xx + rng.uniform(-4, 4, xx.shape).astype(np.float32), yy + rng.uniform(-4, 4, xx.shape).astype(np.float32)),
        "border": (xx * 1.1 - 20, yy * 1.15 - 25),                 # leaves the image on every side (zero border, partial taps)
        "far": (xx + 40000, yy - 40000),                            # beyond short range: saturated, all zero
        "halves": (xx + 0.015625, yy + 0.484375),                   # x * 32 exactly at .5: round half to even
    }
    for eye, (name, (mx, my)) in enumerate(cases.items()):
        ctx.set_rectify_map(eye & 1, mx, my)
        ctx.upload_raw(eye & 1, eye & 1, img)
        got = ctx.pyramid_level(eye & 1, 0)
        assert np.array_equal(got, rect.remap_linear_u8(img, mx, my)), name
    with pytest.raises(RuntimeError):
        hiplib.Context(W, H, 300, 1.2, 4, max_images=1).upload_raw(0, 0, img)       # no map set


def test_rectified_pair_feeds_the_front_end(hiplib, rect, oracle, ctx):
    K1 = np.array([[262.0, 0, 161.0], [0, 263.0, 120.5], [0, 0, 1]]); K2 = np.array([[261.0, 0, 159.0], [0, 262.5, 119.0], [0, 0, 1]])
    D1 = np.array([-0.17, 0.025, 0.0007, -0.0004, 0.0]); D2 = np.array([-0.168, 0.024, -0.0005, 0.0003, 0.0])
    R = rect.rodrigues_to_matrix([0.004, -0.007, 0.002]); T = np.array([-0.12, 0.001, -0.0008])
    R1, R2, P1, P2 = rect.stereo_rectify(K1, D1, K2, D2, (W, H), R, T)
    l, r = synth.StereoSequence(W, H, 2, n_points=2500).frame(0)
    for eye, (img, K, D, Rk, P) in enumerate(((l, K1, D1, R1, P1), (r, K2, D2, R2, P2))):
        mx, my = rect.init_undistort_rectify_map(K, D, Rk, P, (W, H))
        ctx.set_rectify_map(eye, mx, my)
        ctx.upload_raw(eye, eye, img)
        want = rect.remap_linear_u8(img, mx, my)
        assert np.array_equal(ctx.pyramid_level(eye, 0), want)
    ctx.extract(2)
    p = oracle.params(300, 1.2, 4)
    for eye, img in enumerate((l, r)):
        kp, desc = ctx.keypoints(eye)
        okp, odesc, _, _ = oracle.extract(ctx.pyramid_level(eye, 0), p)
        assert len(kp) == len(okp) > 100 and np.array_equal(desc, odesc)                 # the front end runs on the remapped level 0


def test_stereo_tracker_on_raw_frames(hiplib, rect):
    """Raw frames = ideal rectified frames pushed through the inverse of the rectification, so that the device-side remap
    restores a consistent stereo pair; cameras configured as pinhole with distortion -> the tracker rectifies on the device."""
    from lpslam_amd import _build, manager
    _build.host_library()
    w, h, n_frames = 640, 480, 12
    k = synth.intrinsics(w, h)
    K = np.array([[k["fx"], 0, k["cx"]], [0, k["fy"], k["cy"]], [0, 0, 1.0]])
    D = np.array([-0.05, 0.01, 0.0003, -0.0002, 0.0])
    R = np.eye(3); T = np.array([-k["baseline"], 0.0, 0.0])
    R1, R2, P1, P2 = rect.stereo_rectify(K, D, K, D, (w, h), R, T)

    def raw_from_ideal(img, Rk, P):      # raw pixel -> where it lies in the rectified image (vectorised cvUndistortPoints)
        v, u = np.mgrid[0:h, 0:w].astype(np.float64)
        x = (u - K[0, 2]) / K[0, 0]; y = (v - K[1, 2]) / K[1, 1]
        x0, y0 = x.copy(), y.copy()
        for _ in range(5):
            r2 = x * x + y * y
            ic = 1.0 / (1 + ((D[4] * r2 + D[1]) * r2 + D[0]) * r2)
            dx = 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x); dy = D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
            x = (x0 - dx) * ic; y = (y0 - dy) * ic
        RR = P[:, :3] @ Rk
        ww = RR[2, 0] * x + RR[2, 1] * y + RR[2, 2]
        mx = ((RR[0, 0] * x + RR[0, 1] * y + RR[0, 2]) / ww).astype(np.float32)
        my = ((RR[1, 0] * x + RR[1, 1] * y + RR[1, 2]) / ww).astype(np.float32)
        return rect.remap_linear_u8(img, mx, my)

    seq = synth.StereoSequence(w, h, 4, n_points=6000)
    m = manager.Manager()
    for num in (0, 1):
        c = manager.default_camera()
        c.camera_number = num; c.distortion_function = manager.PINHOLE
        c.f_x = k["fx"]; c.f_y = k["fy"]; c.c_x = k["cx"]; c.c_y = k["cy"]
        for i, v in enumerate(D):
            c.dist[i] = v
        for i, v in enumerate(T):
            c.translation[i] = v
        c.resolution_x = w; c.resolution_y = h; c.focal_x_baseline = k["fxb"]
        m.set_camera(c)
    assert m.add_tracker("VSLAMStereo", '{"cameraSetup": "stereo", "slamKeypoints": 1000, "numLevels": 4, "keyframeInterval": 4}')
    m.collect_results(); m.provide_odometry()
    m.start()
    for i in range(n_frames):
        l, r = seq.frame(i)
        assert m.add_stereo((i + 1) * 40_000_000, raw_from_ideal(l, R1, P1), raw_from_ideal(r, R2, P2))
    t0 = time.time()
    while len(m.results) < n_frames and time.time() - t0 < 60:
        time.sleep(0.01)
    st = m.status()
    m.stop()
    assert len(m.results) == n_frames
    valid = [r_ for r_ in m.results if r_["valid"]]
    assert len(valid) >= n_frames - 3 and st.key_frames >= 2 and st.feature_points > 50
