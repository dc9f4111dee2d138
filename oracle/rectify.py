"""oracle/rectify.py -- numpy restatement of the undistort / rectify step.  TEST INFRASTRUCTURE ONLY: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  PARITY UNPINNED: the arithmetic lives in
OpenCV (>= 4.2, /root/reference/CMakeLists.txt:53-58), which is not in the image; the reference has no fixture for it.

Restates, from the published OpenCV 4.x sources, what /root/reference/src/Utils/ImageProcessing.h:134-250 calls:
  cv::stereoRectify(K1, D1, K2, D2, size, R, T, ..., CALIB_ZERO_DISPARITY, alpha = 0, newImageSize = size)   (:182-183)
  cv::initUndistortRectifyMap(K, D, R, P, size, CV_32FC1)            (pinhole, :203-206)
  cv::fisheye::initUndistortRectifyMap(K, D, R, P, size, CV_32FC1)   (fisheye, :198-201)
  cv::remap(src, dst, map1, map2, cv::INTER_LINEAR)                  (per frame, :248; BORDER_CONSTANT 0)
with their helpers cvRodrigues2, cvUndistortPoints (5 fixed-point iterations), cvProjectPoints2 (no distortion),
icvGetRectangles.  Points that OpenCV keeps as CV_32F are rounded to float32 at the same places.
"""
import numpy as np


# ---- cv::remap, CV_32FC1 maps, INTER_LINEAR, BORDER_CONSTANT(0), 8UC1 ------------------------------------------------------
def convert_maps(map_x, map_y):
    """remap's own float -> fixed-point conversion: sx = cvRound(x * 32) in float (round half to even);
    returns integer coordinates (saturated to short) and the 5-bit fractions."""
    fx = (np.asarray(map_x, np.float32) * np.float32(32.0)).astype(np.float32)
    fy = (np.asarray(map_y, np.float32) * np.float32(32.0)).astype(np.float32)
    sx = np.rint(fx.astype(np.float64)).astype(np.int64)
    sy = np.rint(fy.astype(np.float64)).astype(np.int64)
    ix = np.clip(sx >> 5, -32768, 32767); iy = np.clip(sy >> 5, -32768, 32767)
    return ix, iy, sx & 31, sy & 31


def remap_linear_u8(src, map_x, map_y):
    src = np.asarray(src, np.uint8)
    h, w = src.shape
    ix, iy, fx, fy = convert_maps(map_x, map_y)
    pad = np.zeros((h + 2, w + 2), np.int64)          # border value 0 around the image
    pad[1:-1, 1:-1] = src
    outside = (ix >= w) | (ix + 1 < 0) | (iy >= h) | (iy + 1 < 0)
    cx = np.clip(ix, -1, w - 1) + 1; cy = np.clip(iy, -1, h - 1) + 1
    t00 = pad[cy, cx]; t01 = pad[cy, cx + 1]; t10 = pad[cy + 1, cx]; t11 = pad[cy + 1, cx + 1]
    w00 = (32 - fx) * (32 - fy) * 32; w01 = fx * (32 - fy) * 32; w10 = (32 - fx) * fy * 32; w11 = fx * fy * 32
    val = (t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11 + (1 << 14)) >> 15
    val[outside] = 0
    return val.astype(np.uint8)


# ---- cvRodrigues2 -------------------------------------------------------------------------------------------------------
def rodrigues_to_matrix(r):
    r = np.asarray(r, np.float64).reshape(3)
    theta = np.linalg.norm(r)
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    c, s = np.cos(theta), np.sin(theta)
    k = r / theta
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * K


def rodrigues_to_vector(R):
    U, _, Vt = np.linalg.svd(np.asarray(R, np.float64))
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt(np.dot(r, r) * 0.25)
    c = np.clip((np.trace(R) - 1) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5; x = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5; y = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5; z = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and (R[1, 2] > 0) != (y * z > 0):
            z = -z
        v = np.array([x, y, z])
        return v * (theta / np.linalg.norm(v))
    return r * (theta / (2 * s))


# ---- cvUndistortPoints (CV_32FC2 in / out), criteria = 5 iterations -------------------------------------------------------
def _dist14(D):
    k = np.zeros(14)
    D = np.asarray(D, np.float64).reshape(-1)
    k[:len(D)] = D                      # k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 tx ty
    return k


def undistort_points(pts, K, D, R=None, P=None):
    pts = np.asarray(pts, np.float32).astype(np.float64)
    k = _dist14(D)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    RR = np.eye(3)
    if R is not None:
        RR = np.asarray(R, np.float64)
    if P is not None:
        RR = np.asarray(P, np.float64)[:3, :3] @ RR
    out = np.zeros_like(pts)
    for i, (u, v) in enumerate(pts):
        x = (u - cx) / fx; y = (v - cy) / fy
        x0, y0 = x, y
        for _ in range(5):
            r2 = x * x + y * y
            icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            if icdist < 0:
                x, y = (u - cx) / fx, (v - cy) / fy
                break
            dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2
            dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2
            x = (x0 - dx) * icdist; y = (y0 - dy) * icdist
        xx = RR[0, 0] * x + RR[0, 1] * y + RR[0, 2]
        yy = RR[1, 0] * x + RR[1, 1] * y + RR[1, 2]
        ww = 1.0 / (RR[2, 0] * x + RR[2, 1] * y + RR[2, 2])
        out[i] = (xx * ww, yy * ww)
    return out.astype(np.float32)


def _get_rectangles(K, D, R, P, size):
    """icvGetRectangles: inner (inscribed) and outer rectangles of the undistorted 9 x 9 grid, as (x, y, w, h) float32."""
    w, h = size
    N = 9
    pts = np.array([[np.float32(x) * np.float32(w) / np.float32(N - 1), np.float32(y) * np.float32(h) / np.float32(N - 1)]
                    for y in range(N) for x in range(N)], np.float32)
    p = undistort_points(pts, K, D, R, P).reshape(N, N, 2)
    ix0 = p[:, 0, 0].max(); ix1 = p[:, N - 1, 0].min(); iy0 = p[0, :, 1].max(); iy1 = p[N - 1, :, 1].min()
    ox0 = p[..., 0].min(); ox1 = p[..., 0].max(); oy0 = p[..., 1].min(); oy1 = p[..., 1].max()
    f = np.float32
    return (f(ix0), f(iy0), f(ix1 - ix0), f(iy1 - iy0)), (f(ox0), f(oy0), f(ox1 - ox0), f(oy1 - oy0))


# ---- cv::stereoRectify (flags = CALIB_ZERO_DISPARITY, newImageSize = imageSize) ---------------------------------------------
def stereo_rectify(K1, D1, K2, D2, size, R, T, alpha=0.0):
    K1 = np.asarray(K1, np.float64); K2 = np.asarray(K2, np.float64)
    nx, ny = size
    om = rodrigues_to_vector(R) * -0.5                 # average rotation
    r_r = rodrigues_to_matrix(om)
    t = r_r @ np.asarray(T, np.float64).reshape(3)
    idx = 0 if abs(t[0]) > abs(t[1]) else 1
    c, nt = t[idx], np.linalg.norm(t)
    uu = np.zeros(3); uu[idx] = 1.0 if c > 0 else -1.0
    ww = np.cross(t, uu)
    nw = np.linalg.norm(ww)
    if nw > 0.0:
        ww = ww * (np.arccos(abs(c) / nt) / nw)
    wR = rodrigues_to_matrix(ww)
    R1 = wR @ r_r.T
    R2 = wR @ r_r
    t = R2 @ np.asarray(T, np.float64).reshape(3)
    ratio = 0.5                                         # newImgSize == imageSize
    fc_new = (K1[idx ^ 1, idx ^ 1] + K2[idx ^ 1, idx ^ 1]) * ratio
    cc = np.zeros((2, 2))
    for k, (A, Dk, Rk) in enumerate(((K1, D1, R1), (K2, D2, R2))):
        pts = np.array([[(i % 2) * (nx - 1), (0 if i < 2 else 1) * (ny - 1)] for i in range(4)], np.float32)
        und = undistort_points(pts, A, Dk).astype(np.float64)                       # CV_32FC2
        p3 = np.concatenate([und, np.ones((4, 1))], axis=1).astype(np.float32).astype(np.float64)   # cvConvertPointsHomogeneous, CV_32FC3
        q = p3 @ Rk.T                                                                # cvProjectPoints2: rotation, zero translation,
        proj = np.stack([fc_new * q[:, 0] / q[:, 2], fc_new * q[:, 1] / q[:, 2]], axis=1).astype(np.float32)   # f = fc_new, c = 0
        avg = proj.astype(np.float64).mean(axis=0)
        cc[k] = ((nx - 1) // 2 - avg[0], (ny - 1) // 2 - avg[1])
    cc[0, 0] = cc[1, 0] = (cc[0, 0] + cc[1, 0]) * 0.5      # CALIB_ZERO_DISPARITY
    cc[0, 1] = cc[1, 1] = (cc[0, 1] + cc[1, 1]) * 0.5
    P1 = np.zeros((3, 4)); P2 = np.zeros((3, 4))
    P1[0, 0] = P1[1, 1] = fc_new; P1[0, 2], P1[1, 2] = cc[0]; P1[2, 2] = 1
    P2[0, 0] = P2[1, 1] = fc_new; P2[0, 2], P2[1, 2] = cc[1]; P2[2, 2] = 1
    P2[idx, 3] = t[idx] * fc_new                         # baseline * focal length
    alpha = min(alpha, 1.0)
    inner1, outer1 = _get_rectangles(K1, D1, R1, P1, size)
    inner2, outer2 = _get_rectangles(K2, D2, R2, P2, size)
    cx1_0, cy1_0 = cc[0]; cx2_0, cy2_0 = cc[1]
    cx1, cy1, cx2, cy2 = cx1_0, cy1_0, cx2_0, cy2_0      # nx * c / nx
    s = 1.0
    if alpha >= 0:
        def terms(cx, cy, cx0, cy0, r):      # Rect_<float>: x + width and y + height are float sums
            right = float(np.float32(r[0]) + np.float32(r[2])); bottom = float(np.float32(r[1]) + np.float32(r[3]))
            return (cx / (cx0 - float(r[0])), cy / (cy0 - float(r[1])), (nx - cx) / (right - cx0), (ny - cy) / (bottom - cy0))
        s0 = max(max(terms(cx1, cy1, cx1_0, cy1_0, inner1)), max(terms(cx2, cy2, cx2_0, cy2_0, inner2)))
        s1 = min(min(terms(cx1, cy1, cx1_0, cy1_0, outer1)), min(terms(cx2, cy2, cx2_0, cy2_0, outer2)))
        s = s0 * (1 - alpha) + s1 * alpha
    fc_new *= s
    P1[0, 0] = P1[1, 1] = fc_new; P1[0, 2], P1[1, 2] = cx1, cy1
    P2[0, 0] = P2[1, 1] = fc_new; P2[0, 2], P2[1, 2] = cx2, cy2
    P2[idx, 3] = s * P2[idx, 3]
    return R1, R2, P1, P2


# ---- cv::initUndistortRectifyMap (CV_32FC1) -------------------------------------------------------------------------------
def init_undistort_rectify_map(K, D, R, P, size):
    w, h = size
    K = np.asarray(K, np.float64)
    k = _dist14(D)
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = k[:12]
    fx, fy, u0, v0 = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ir = np.linalg.inv(np.asarray(P, np.float64)[:3, :3] @ np.asarray(R, np.float64)).reshape(-1)       # DECOMP_LU
    map_x = np.zeros((h, w), np.float32); map_y = np.zeros((h, w), np.float32)
    for i in range(h):
        # the column loop advances _x, _y, _w by repeated addition
        steps = np.arange(w, dtype=np.float64)
        _x = np.concatenate([[i * ir[1] + ir[2]], np.full(w - 1, ir[0])]).cumsum() if w > 1 else np.array([i * ir[1] + ir[2]])
        _y = np.concatenate([[i * ir[4] + ir[5]], np.full(w - 1, ir[3])]).cumsum() if w > 1 else np.array([i * ir[4] + ir[5]])
        _w = np.concatenate([[i * ir[7] + ir[8]], np.full(w - 1, ir[6])]).cumsum() if w > 1 else np.array([i * ir[7] + ir[8]])
        del steps
        iw = 1.0 / _w
        x = _x * iw; y = _y * iw
        x2 = x * x; y2 = y * y; r2 = x2 + y2; _2xy = 2 * x * y
        kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
        xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2
        yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2
        map_x[i] = (fx * xd + u0).astype(np.float32)
        map_y[i] = (fy * yd + v0).astype(np.float32)
    return map_x, map_y


def fisheye_init_undistort_rectify_map(K, D, R, P, size):
    w, h = size
    K = np.asarray(K, np.float64)
    k = np.asarray(D, np.float64).reshape(-1)[:4]
    f = (K[0, 0], K[1, 1]); c = (K[0, 2], K[1, 2])
    iR = np.linalg.inv(np.asarray(P, np.float64)[:3, :3] @ np.asarray(R, np.float64))                     # DECOMP_SVD upstream
    map_x = np.zeros((h, w), np.float32); map_y = np.zeros((h, w), np.float32)
    for i in range(h):
        _x = np.concatenate([[i * iR[0, 1] + iR[0, 2]], np.full(w - 1, iR[0, 0])]).cumsum()
        _y = np.concatenate([[i * iR[1, 1] + iR[1, 2]], np.full(w - 1, iR[1, 0])]).cumsum()
        _w = np.concatenate([[i * iR[2, 1] + iR[2, 2]], np.full(w - 1, iR[2, 0])]).cumsum()
        x = _x / _w; y = _y / _w
        r = np.sqrt(x * x + y * y)
        theta = np.arctan(r)
        t2 = theta * theta; t4 = t2 * t2; t6 = t4 * t2; t8 = t4 * t4
        theta_d = theta * (1 + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8)
        scale = np.where(r == 0, 1.0, theta_d / np.where(r == 0, 1.0, r))
        map_x[i] = (f[0] * x * scale + c[0]).astype(np.float32)
        map_y[i] = (f[1] * y * scale + c[1]).astype(np.float32)
    return map_x, map_y
