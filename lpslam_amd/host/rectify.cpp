// rectify.cpp -- see rectify.h.  OpenCV routines restated: calib3d cvStereoRectify / icvGetRectangles / cvUndistortPoints /
// cvRodrigues2, imgproc initUndistortRectifyMap, calib3d fisheye::initUndistortRectifyMap.
#include "rectify.h"
#include <fstream>
#include <iterator>
#include <cmath>
#include <cfloat>
#include <algorithm>

namespace LpSlam {
namespace {

void mat3_mul(const double* A, const double* B, double* C)
{
    double t[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    std::copy(t, t + 9, C);
}
void mat3_mul_bt(const double* A, const double* B, double* C)       // A * B^T
{
    double t[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
    std::copy(t, t + 9, C);
}
void mat3_vec(const double* A, const double* v, double* o)
{
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
    std::copy(t, t + 3, o);
}
void mat3_inv(const double* A, double* Ai)
{
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02, id = 1.0 / det;
    double t[9] = {c00 * id, (A[2] * A[7] - A[1] * A[8]) * id, (A[1] * A[5] - A[2] * A[4]) * id,
                   c01 * id, (A[0] * A[8] - A[2] * A[6]) * id, (A[2] * A[3] - A[0] * A[5]) * id,
                   c02 * id, (A[1] * A[6] - A[0] * A[7]) * id, (A[0] * A[4] - A[1] * A[3]) * id};
    std::copy(t, t + 9, Ai);
}
double norm3(const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// cvRodrigues2, vector -> matrix
void rodrigues_to_matrix(const double* r, double* R)
{
    const double theta = norm3(r);
    if (theta < DBL_EPSILON) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1 - c, it = 1.0 / theta;
    const double k[3] = {r[0] * it, r[1] * it, r[2] * it};
    const double K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i * 3 + j] = c * (i == j ? 1.0 : 0.0) + c1 * k[i] * k[j] + s * K[i * 3 + j];
}
// cvRodrigues2, matrix -> vector (the matrix is taken as orthonormal)
void rodrigues_to_vector(const double* R, double* r)
{
    r[0] = R[7] - R[5]; r[1] = R[2] - R[6]; r[2] = R[3] - R[1];
    const double s = std::sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    const double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t = (R[0] + 1) * 0.5; double x = std::sqrt(std::max(t, 0.0));
        t = (R[4] + 1) * 0.5; double y = std::sqrt(std::max(t, 0.0)) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; double z = std::sqrt(std::max(t, 0.0)) * (R[2] < 0 ? -1. : 1.);
        if (std::fabs(x) < std::fabs(y) && std::fabs(x) < std::fabs(z) && (R[5] > 0) != (y * z > 0)) z = -z;
        const double v[3] = {x, y, z}, f = theta / norm3(v);
        r[0] = x * f; r[1] = y * f; r[2] = z * f;
        return;
    }
    const double vth = theta / (2 * s);
    r[0] *= vth; r[1] *= vth; r[2] *= vth;
}

struct Dist { double k[14]; };
Dist dist14(const double* D, int n) { Dist d{}; for (int i = 0; i < n && i < 14; ++i) d.k[i] = D[i]; return d; }

// cvUndistortPoints on CV_32FC2 points, 5 iterations, optional RR = P[:3,:3] * R
void undistort_points(float* pts, int n, const double* K, const Dist& d, const double* R, const double* P)
{
    const double* k = d.k;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double RR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (R) std::copy(R, R + 9, RR);
    if (P) { const double P3[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}; mat3_mul(P3, RR, RR); }
    for (int i = 0; i < n; ++i) {
        const double u = pts[2 * i], v = pts[2 * i + 1];
        double x = (u - cx) / fx, y = (v - cy) / fy;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) { x = (u - cx) / fx; y = (v - cy) / fy; break; }
            const double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - dx) * icdist; y = (y0 - dy) * icdist;
        }
        const double xx = RR[0] * x + RR[1] * y + RR[2], yy = RR[3] * x + RR[4] * y + RR[5], ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
        pts[2 * i] = (float)(xx * ww); pts[2 * i + 1] = (float)(yy * ww);
    }
}

struct RectF { float x, y, w, h; };
// icvGetRectangles
void get_rectangles(const double* K, const Dist& d, const double* R, const double* P, int width, int height, RectF& inner, RectF& outer)
{
    const int N = 9;
    float pts[2 * N * N];
    for (int y = 0, k = 0; y < N; ++y) for (int x = 0; x < N; ++x, ++k) { pts[2 * k] = (float)x * width / (N - 1); pts[2 * k + 1] = (float)y * height / (N - 1); }
    undistort_points(pts, N * N, K, d, R, P);
    float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX, oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
    for (int y = 0, k = 0; y < N; ++y) for (int x = 0; x < N; ++x, ++k) {
        const float px = pts[2 * k], py = pts[2 * k + 1];
        oX0 = std::min(oX0, px); oX1 = std::max(oX1, px); oY0 = std::min(oY0, py); oY1 = std::max(oY1, py);
        if (x == 0) iX0 = std::max(iX0, px);
        if (x == N - 1) iX1 = std::min(iX1, px);
        if (y == 0) iY0 = std::max(iY0, py);
        if (y == N - 1) iY1 = std::min(iY1, py);
    }
    inner = RectF{iX0, iY0, iX1 - iX0, iY1 - iY0};
    outer = RectF{oX0, oY0, oX1 - oX0, oY1 - oY0};
}

}  // namespace

void stereo_rectify(const double* K1, const double* D1, int n1, const double* K2, const double* D2, int n2, int nx, int ny,
                    const double* R, const double* T, double* R1, double* R2, double* P1, double* P2)
{
    double om[3], r_r[9], t[3], ww[3], wR[9];
    rodrigues_to_vector(R, om);
    for (int i = 0; i < 3; ++i) om[i] *= -0.5;                 // average rotation
    rodrigues_to_matrix(om, r_r);
    mat3_vec(r_r, T, t);
    const int idx = std::fabs(t[0]) > std::fabs(t[1]) ? 0 : 1;
    const double c = t[idx], nt = norm3(t);
    double uu[3] = {0, 0, 0};
    uu[idx] = c > 0 ? 1 : -1;
    ww[0] = t[1] * uu[2] - t[2] * uu[1]; ww[1] = t[2] * uu[0] - t[0] * uu[2]; ww[2] = t[0] * uu[1] - t[1] * uu[0];
    const double nw = norm3(ww);
    if (nw > 0.0) { const double f = std::acos(std::fabs(c) / nt) / nw; for (int i = 0; i < 3; ++i) ww[i] *= f; }
    rodrigues_to_matrix(ww, wR);
    mat3_mul_bt(wR, r_r, R1);
    mat3_mul(wR, r_r, R2);
    mat3_vec(R2, T, t);
    double fc_new = (K1[(idx ^ 1) * 4] + K2[(idx ^ 1) * 4]) * 0.5;       // newImageSize == imageSize: ratio 1/2
    const Dist d1 = dist14(D1, n1), d2 = dist14(D2, n2);
    double cc[2][2];
    for (int k = 0; k < 2; ++k) {
        const double* A = k == 0 ? K1 : K2;
        const double* Rk = k == 0 ? R1 : R2;
        float pts[8];
        for (int i = 0; i < 4; ++i) { pts[2 * i] = (float)((i % 2) * (nx - 1)); pts[2 * i + 1] = (float)((i < 2 ? 0 : 1) * (ny - 1)); }
        undistort_points(pts, 4, A, k == 0 ? d1 : d2, nullptr, nullptr);
        double ax = 0, ay = 0;
        for (int i = 0; i < 4; ++i) {
            const double p3[3] = {(double)pts[2 * i], (double)pts[2 * i + 1], 1.0};      // cvConvertPointsHomogeneous (CV_32FC3)
            double q[3];
            mat3_vec(Rk, p3, q);                                                          // cvProjectPoints2: f = fc_new, c = 0, no distortion
            ax += (double)(float)(fc_new * q[0] / q[2]); ay += (double)(float)(fc_new * q[1] / q[2]);
        }
        cc[k][0] = (nx - 1) / 2 - ax * 0.25; cc[k][1] = (ny - 1) / 2 - ay * 0.25;       // integer division, as upstream
    }
    cc[0][0] = cc[1][0] = (cc[0][0] + cc[1][0]) * 0.5;                                  // CALIB_ZERO_DISPARITY
    cc[0][1] = cc[1][1] = (cc[0][1] + cc[1][1]) * 0.5;
    std::fill(P1, P1 + 12, 0.0); std::fill(P2, P2 + 12, 0.0);
    P1[0] = P1[5] = fc_new; P1[2] = cc[0][0]; P1[6] = cc[0][1]; P1[10] = 1;
    P2[0] = P2[5] = fc_new; P2[2] = cc[1][0]; P2[6] = cc[1][1]; P2[10] = 1;
    P2[idx * 4 + 3] = t[idx] * fc_new;                                                  // baseline * focal length
    RectF inner1, outer1, inner2, outer2;
    get_rectangles(K1, d1, R1, P1, nx, ny, inner1, outer1);
    get_rectangles(K2, d2, R2, P2, nx, ny, inner2, outer2);
    const double cx1 = cc[0][0], cy1 = cc[0][1], cx2 = cc[1][0], cy2 = cc[1][1];
    auto smax = [&](double cx, double cy, const RectF& r) {
        return std::max(std::max(std::max(cx / (cx - r.x), cy / (cy - r.y)), (nx - cx) / (r.x + r.w - cx)), (ny - cy) / (r.y + r.h - cy));
    };
    const double s = std::max(smax(cx1, cy1, inner1), smax(cx2, cy2, inner2));          // alpha = 0: all pixels valid
    fc_new *= s;
    P1[0] = P1[5] = fc_new; P2[0] = P2[5] = fc_new;
    P2[idx * 4 + 3] *= s;
    (void)outer1; (void)outer2;
}

void init_undistort_rectify_map(const double* K, const double* D, int nd, const double* R, const double* P, int width, int height,
                                float* map_x, float* map_y)
{
    const Dist d = dist14(D, nd);
    const double k1 = d.k[0], k2 = d.k[1], p1 = d.k[2], p2 = d.k[3], k3 = d.k[4], k4 = d.k[5], k5 = d.k[6], k6 = d.k[7];
    const double s1 = d.k[8], s2 = d.k[9], s3 = d.k[10], s4 = d.k[11];
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    const double P3[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    double PR[9], ir[9];
    mat3_mul(P3, R, PR);
    mat3_inv(PR, ir);
    for (int i = 0; i < height; ++i) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < width; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2;
            const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2;
            map_x[(size_t)i * width + j] = (float)(fx * xd + u0);
            map_y[(size_t)i * width + j] = (float)(fy * yd + v0);
        }
    }
}

void fisheye_init_undistort_rectify_map(const double* K, const double* k, const double* R, const double* P, int width, int height,
                                        float* map_x, float* map_y)
{
    const double f0 = K[0], f1 = K[4], c0 = K[2], c1 = K[5];
    const double P3[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    double PR[9], iR[9];
    mat3_mul(P3, R, PR);
    mat3_inv(PR, iR);
    for (int i = 0; i < height; ++i) {
        double _x = i * iR[1] + iR[2], _y = i * iR[4] + iR[5], _w = i * iR[7] + iR[8];
        for (int j = 0; j < width; ++j) {
            const double x = _x / _w, y = _y / _w;
            const double r = std::sqrt(x * x + y * y);
            const double theta = std::atan(r);
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
            const double theta_d = theta * (1 + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8);
            const double scale = (r == 0) ? 1.0 : theta_d / r;
            map_x[(size_t)i * width + j] = (float)(f0 * x * scale + c0);
            map_y[(size_t)i * width + j] = (float)(f1 * y * scale + c1);
            _x += iR[0]; _y += iR[3]; _w += iR[6];
        }
    }
}

bool build_rectify_maps(const LpSlamCameraConfiguration& left, const LpSlamCameraConfiguration& right, bool is_left,
                        RectifyMaps& out, std::string* err)
{
    auto fail = [&](const char* m) { if (err) *err = m; return false; };
    const auto fn = left.distortion_function;
    if (fn == LpSlamCameraDistortionFunction_NoDistortion) return fail("no_distortion: frames pass through");
    if (fn == LpSlamCameraDistortionFunction_Omni) return fail("omni rectification is not available (the reference's branch is commented out as well)");
    if (fn != LpSlamCameraDistortionFunction_Pinhole && fn != LpSlamCameraDistortionFunction_Fisheye) return fail("distortion function not supported");
    if (left.resolution_x < 2 || left.resolution_y < 2) return fail("camera resolution not set");
    auto cam_matrix = [](const LpSlamCameraConfiguration& c, double* K) {              // getCameraMatrix, ImageProcessing.h:104-110
        K[0] = c.f_x; K[1] = 0; K[2] = c.c_x; K[3] = 0; K[4] = c.f_y; K[5] = c.c_y; K[6] = 0; K[7] = 0; K[8] = 1; };
    auto n_dist = [](const LpSlamCameraConfiguration& c) {                              // getDistortionCoeffs, ImageProcessing.h:62-102
        if (c.distortion_function == LpSlamCameraDistortionFunction_Fisheye) return 4;
        if (c.distortion_function == LpSlamCameraDistortionFunction_Pinhole) return c.dist[5] == 0 ? 5 : 8;
        return 5; };
    auto dist = [&](const LpSlamCameraConfiguration& c, double* D) {
        const int n = n_dist(c);
        for (int i = 0; i < 8; ++i) D[i] = (c.distortion_function == LpSlamCameraDistortionFunction_NoDistortion || i >= n) ? 0.0 : c.dist[i]; };
    double K1[9], K2[9], D1[8], D2[8], R1[9], R2[9], P1[12], P2[12];
    cam_matrix(left, K1); cam_matrix(right, K2);
    dist(left, D1); dist(right, D2);
    const int w = left.resolution_x, h = left.resolution_y;
    // stereo extrinsics are read from the LEFT camera's rotation / translation (ImageProcessing.h:160-161)
    stereo_rectify(K1, D1, n_dist(left), K2, D2, n_dist(right), w, h, left.rotation, left.translation, R1, R2, P1, P2);
    out.width = w; out.height = h;
    out.map_x.assign((size_t)w * h, 0.f); out.map_y.assign((size_t)w * h, 0.f);
    const double* K = is_left ? K1 : K2; const double* D = is_left ? D1 : D2;
    const double* R = is_left ? R1 : R2; const double* P = is_left ? P1 : P2;
    if (fn == LpSlamCameraDistortionFunction_Fisheye) fisheye_init_undistort_rectify_map(K, D, R, P, w, h, out.map_x.data(), out.map_y.data());
    else init_undistort_rectify_map(K, D, n_dist(is_left ? left : right), R, P, w, h, out.map_x.data(), out.map_y.data());
    return true;
}

bool build_camera_mask(const LpSlamCameraConfiguration& cam, bool is_left, std::vector<uint8_t>& mask, std::string* err)
{
    const int w = cam.resolution_x, h = cam.resolution_y;
    if (w <= 0 || h <= 0) { if (err) *err = "camera resolution missing"; return false; }
    if (cam.mask_type == LpSlamCameraMaskType_Radial) {
        // cv::circle(mask, (w/2, h/2), int(radius), 255, FILLED): every pixel whose centre lies within the radius
        const int cx = w / 2, cy = h / 2, r = (int)cam.mask_parameter;
        mask.assign((size_t)w * h, 0);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                if ((long)(x - cx) * (x - cx) + (long)(y - cy) * (y - cy) <= (long)r * r) mask[(size_t)y * w + x] = 255;
        return true;
    }
    if (cam.mask_type == LpSlamCameraMaskType_Image) {
        const std::string name = std::string("camera_mask_") + (is_left ? "left" : "right") + ".bmp";
        std::ifstream f(name, std::ios::binary);
        if (!f) { if (err) *err = "cannot open " + name; return false; }
        std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        auto u16 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8); };
        auto u32 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24); };
        if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') { if (err) *err = name + " is not a BMP file"; return false; }
        const uint32_t off = u32(10), hdr = u32(14), bpp = u16(28), comp = u32(30);
        const int bw = (int)u32(18); int bh = (int)u32(22);
        const bool top_down = bh < 0;
        if (top_down) bh = -bh;
        if (hdr < 40 || comp != 0 || (bpp != 8 && bpp != 24 && bpp != 32)) { if (err) *err = name + ": only uncompressed 8 / 24 / 32-bit BMP is read"; return false; }
        if (bw != w || bh != h) { if (err) *err = name + " does not have the camera's resolution"; return false; }
        const size_t row = ((size_t)bw * bpp / 8 + 3) & ~(size_t)3;
        if (d.size() < off + row * (size_t)bh) { if (err) *err = name + " is truncated"; return false; }
        const size_t pal = 14 + hdr;                     // 8-bit: palette of BGRA quads
        mask.assign((size_t)w * h, 0);
        for (int y = 0; y < h; ++y) {
            const uint8_t* src = d.data() + off + row * (size_t)(top_down ? y : h - 1 - y);
            for (int x = 0; x < w; ++x) {
                int b, g, r;
                if (bpp == 8) { const size_t e = pal + 4 * (size_t)src[x]; if (e + 2 >= d.size()) { b = g = r = src[x]; } else { b = d[e]; g = d[e + 1]; r = d[e + 2]; } }
                else { const uint8_t* px = src + (size_t)x * (bpp / 8); b = px[0]; g = px[1]; r = px[2]; }
                // cv::cvtColor BGR2GRAY fixed point: (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14
                mask[(size_t)y * w + x] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
            }
        }
        return true;
    }
    if (err) *err = "no mask configured";
    return false;
}

}  // namespace LpSlam

// C shim for the tests (ctypes)
extern "C" {
__attribute__((visibility("default"))) void lpslam_rectify_stereo(const double* K1, const double* D1, int n1, const double* K2, const double* D2, int n2,
                                                                  int w, int h, const double* R, const double* T, double* R1, double* R2, double* P1, double* P2)
{ LpSlam::stereo_rectify(K1, D1, n1, K2, D2, n2, w, h, R, T, R1, R2, P1, P2); }
__attribute__((visibility("default"))) int lpslam_rectify_maps(const LpSlamCameraConfiguration* left, const LpSlamCameraConfiguration* right, int is_left,
                                                               float* map_x, float* map_y)
{
    LpSlam::RectifyMaps m; std::string err;
    if (!LpSlam::build_rectify_maps(*left, *right, is_left != 0, m, &err)) return 0;
    std::copy(m.map_x.begin(), m.map_x.end(), map_x); std::copy(m.map_y.begin(), m.map_y.end(), map_y);
    return 1;
}
}
