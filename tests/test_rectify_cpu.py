"""CPU tests of the undistort / rectify restatement (oracle/rectify.py) and of the host library's map generation
(lpslam_amd/host/rectify.cpp, the product side of ImageProcessing::Undistort) against it."""
import ctypes as C

import numpy as np
import pytest

from lpslam_amd import synth

W, H = 320, 240
K1 = np.array([[262.0, 0, 161.0], [0, 263.0, 120.5], [0, 0, 1]])
K2 = np.array([[261.0, 0, 159.0], [0, 262.5, 119.0], [0, 0, 1]])
D1 = np.array([-0.17, 0.025, 0.0007, -0.0004, 0.0])
D2 = np.array([-0.168, 0.024, -0.0005, 0.0003, 0.0])
T = np.array([-0.12, 0.001, -0.0008])


@pytest.fixture(scope="module")
def rect():
    from oracle import rectify
    return rectify


@pytest.fixture(scope="module")
def hostlib():
    from lpslam_amd import _build
    return C.CDLL(_build.host_library())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_remap_definition(rect):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (H, W)).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    assert np.array_equal(rect.remap_linear_u8(img, xx, yy), img)                          # identity map
    assert np.array_equal(rect.remap_linear_u8(img, xx + 1, yy)[:, :-1], img[:, 1:])       # integer shift, zero border
    assert (rect.remap_linear_u8(img, xx + 1, yy)[:, -1] == 0).all()
    mx = xx + rng.uniform(-3, 3, xx.shape).astype(np.float32); my = yy + rng.uniform(-3, 3, xx.shape).astype(np.float32)
    out = rect.remap_linear_u8(img, mx, my)
    # float bilinear interpolation of the same 1/32-quantised coordinates, zero outside: equal within 1 grey level
    qx = np.rint(mx.astype(np.float64) * 32) / 32; qy = np.rint(my.astype(np.float64) * 32) / 32
    x0 = np.floor(qx).astype(int); y0 = np.floor(qy).astype(int); ax = qx - x0; ay = qy - y0
    pad = np.zeros((H + 8, W + 8)); pad[4:-4, 4:-4] = img
    g = lambda yy_, xx_: pad[np.clip(yy_ + 4, 0, H + 7), np.clip(xx_ + 4, 0, W + 7)]
    ref = g(y0, x0) * (1 - ax) * (1 - ay) + g(y0, x0 + 1) * ax * (1 - ay) + g(y0 + 1, x0) * (1 - ax) * ay + g(y0 + 1, x0 + 1) * ax * ay
    assert np.abs(out.astype(float) - ref).max() <= 1.0
    far = rect.remap_linear_u8(img, xx + 10000, yy - 10000)
    assert (far == 0).all()


def test_rodrigues_roundtrip(rect):
    rng = np.random.default_rng(1)
    for _ in range(10):
        r = rng.normal(0, 0.8, 3)
        R = rect.rodrigues_to_matrix(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and np.allclose(rect.rodrigues_to_vector(R), r, atol=1e-12)
    assert np.allclose(rect.rodrigues_to_vector(np.eye(3)), 0)


def test_stereo_rectify_geometry(rect):
    R = rect.rodrigues_to_matrix([0.004, -0.007, 0.002])
    R1, R2, P1, P2 = rect.stereo_rectify(K1, D1, K2, D2, (W, H), R, T)
    assert np.allclose(R1 @ R1.T, np.eye(3), atol=1e-13) and np.allclose(R2 @ R @ R1.T, np.eye(3), atol=1e-12)   # R2 R = R1: common frame
    assert P1[0, 0] == P1[1, 1] == P2[0, 0] == P2[1, 1] and np.array_equal(P1[:, :3], P2[:, :3])                 # zero disparity at infinity
    assert np.isclose(P2[0, 3], -P1[0, 0] * np.linalg.norm(T), rtol=1e-6) and P2[1, 3] == 0                      # -f * baseline
    # epipolar lines are rows: a world point projects to the same y in both rectified views
    X1 = np.array([0.3, -0.2, 4.0]); X2 = R @ X1 + T
    y1 = (P1[:, :3] @ (R1 @ X1)); y2 = (P2[:, :3] @ (R2 @ X2))
    assert abs(y1[1] / y1[2] - y2[1] / y2[2]) < 1e-9
    d = y1[0] / y1[2] - y2[0] / y2[2]
    assert np.isclose(d, -P2[0, 3] / (R1 @ X1)[2], rtol=1e-9)                                                    # disparity = f b / Z
    # alpha = 0: every rectified pixel maps inside the source image
    for K, D, Rk, P in ((K1, D1, R1, P1), (K2, D2, R2, P2)):
        mx, my = rect.init_undistort_rectify_map(K, D, Rk, P, (W, H))
        assert mx.min() >= -0.5 and mx.max() <= W - 0.5 and my.min() >= -0.5 and my.max() <= H - 0.5


def test_maps_invert_the_distortion_model(rect):
    R = rect.rodrigues_to_matrix([0.004, -0.007, 0.002])
    R1, R2, P1, P2 = rect.stereo_rectify(K1, D1, K2, D2, (W, H), R, T)
    mx, my = rect.init_undistort_rectify_map(K1, D1, R1, P1, (W, H))
    # a rectified pixel's map entry is where that ray falls in the raw image: undistorting that raw point with (R1, P1)
    # must give the pixel back
    pts = np.array([[mx[v, u], my[v, u]] for v, u in ((10, 12), (120, 160), (200, 300), (239, 0))], np.float32)
    back = rect.undistort_points(pts, K1, D1, R1, P1)
    assert np.abs(back - np.array([[12, 10], [160, 120], [300, 200], [0, 239]])).max() < 2e-2
    fx, fy = rect.fisheye_init_undistort_rectify_map(K1, D1[:4] * 0.1, R1, P1, (W, H))
    # fisheye model by hand at one pixel
    u, v = 200, 90
    ray = np.linalg.inv(P1[:, :3] @ R1) @ np.array([u, v, 1.0]); x, y = ray[0] / ray[2], ray[1] / ray[2]
    r = np.hypot(x, y); th = np.arctan(r); k = D1[:4] * 0.1
    thd = th * (1 + k[0] * th**2 + k[1] * th**4 + k[2] * th**6 + k[3] * th**8)
    assert abs(fx[v, u] - (K1[0, 0] * x * thd / r + K1[0, 2])) < 1e-3 and abs(fy[v, u] - (K1[1, 1] * y * thd / r + K1[1, 2])) < 1e-3


def _config(mgr, num, K, D, fn, R, T_):
    c = mgr.default_camera()
    c.camera_number = num; c.distortion_function = fn
    c.f_x, c.f_y, c.c_x, c.c_y = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    for i, v in enumerate(D):
        c.dist[i] = v
    c.resolution_x, c.resolution_y = W, H
    c.focal_x_baseline = 30.0
    for i, v in enumerate(R.reshape(-1)):
        c.rotation[i] = v
    for i, v in enumerate(T_):
        c.translation[i] = v
    return c


@pytest.mark.parametrize("model", ["pinhole", "fisheye"])
def test_host_map_generation_matches_the_oracle(rect, hostlib, model):
    from lpslam_amd import manager
    R = rect.rodrigues_to_matrix([0.004, -0.007, 0.002])
    fisheye = model == "fisheye"
    fn = manager.FISHEYE if fisheye else manager.PINHOLE
    d1 = D1[:4] * 0.1 if fisheye else D1; d2 = D2[:4] * 0.1 if fisheye else D2
    left = _config(manager, 0, K1, d1, fn, R, T); right = _config(manager, 1, K2, d2, fn, np.eye(3), np.zeros(3))
    # the reference hands the fisheye coefficients to the (pinhole) cv::stereoRectify as they are: 4 values = k1 k2 p1 p2
    R1, R2, P1, P2 = rect.stereo_rectify(K1, d1, K2, d2, (W, H), R, T)
    o = [np.zeros(9), np.zeros(9), np.zeros(12), np.zeros(12)]
    hostlib.lpslam_rectify_stereo(_p(K1), _p(np.ascontiguousarray(d1)), len(d1), _p(K2), _p(np.ascontiguousarray(d2)), len(d2), W, H,
                                  _p(np.ascontiguousarray(R)), _p(T), *[_p(x) for x in o])
    assert np.allclose(o[0].reshape(3, 3), R1, atol=1e-12) and np.allclose(o[1].reshape(3, 3), R2, atol=1e-12)
    assert np.allclose(o[2].reshape(3, 4), P1, atol=1e-9) and np.allclose(o[3].reshape(3, 4), P2, atol=1e-9)
    for is_left, (K, D, Rk, P) in ((1, (K1, d1, R1, P1)), (0, (K2, d2, R2, P2))):
        mx = np.zeros((H, W), np.float32); my = np.zeros((H, W), np.float32)
        assert hostlib.lpslam_rectify_maps(C.byref(left), C.byref(right), is_left, _p(mx), _p(my)) == 1
        ox, oy = (rect.fisheye_init_undistort_rectify_map if fisheye else rect.init_undistort_rectify_map)(K, D, Rk, P, (W, H))
        assert np.abs(mx - ox).max() < 1e-3 and np.abs(my - oy).max() < 1e-3              # float maps: a few ulp at most
        assert (mx == ox).mean() > 0.99 and (my == oy).mean() > 0.99
    nd = _config(manager, 0, K1, D1, manager.NO_DISTORTION, R, T)
    assert hostlib.lpslam_rectify_maps(C.byref(nd), C.byref(right), 1, _p(mx), _p(my)) == 0
