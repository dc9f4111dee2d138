// two_view.cpp -- see two_view.h
#include "two_view.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <limits>

namespace LpSlam {

namespace {

struct M3 { double m[9]; };

inline M3 mul(const M3& a, const M3& b)
{
    M3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
    return r;
}
inline M3 transpose(const M3& a) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = a.m[j * 3 + i]; return r; }
inline double det(const M3& a)
{
    return a.m[0] * (a.m[4] * a.m[8] - a.m[5] * a.m[7]) - a.m[1] * (a.m[3] * a.m[8] - a.m[5] * a.m[6]) + a.m[2] * (a.m[3] * a.m[7] - a.m[4] * a.m[6]);
}
inline M3 inverse(const M3& a)
{
    const double d = det(a), id = 1.0 / d;
    M3 r;
    r.m[0] = (a.m[4] * a.m[8] - a.m[5] * a.m[7]) * id; r.m[1] = (a.m[2] * a.m[7] - a.m[1] * a.m[8]) * id; r.m[2] = (a.m[1] * a.m[5] - a.m[2] * a.m[4]) * id;
    r.m[3] = (a.m[5] * a.m[6] - a.m[3] * a.m[8]) * id; r.m[4] = (a.m[0] * a.m[8] - a.m[2] * a.m[6]) * id; r.m[5] = (a.m[2] * a.m[3] - a.m[0] * a.m[5]) * id;
    r.m[6] = (a.m[3] * a.m[7] - a.m[4] * a.m[6]) * id; r.m[7] = (a.m[1] * a.m[6] - a.m[0] * a.m[7]) * id; r.m[8] = (a.m[0] * a.m[4] - a.m[1] * a.m[3]) * id;
    return r;
}

// M = U diag(s) V^T, s descending; U, V proper or improper as they come (callers fix signs)
void svd3(const M3& M, M3& U, double* s, M3& V)
{
    const M3 MtM = mul(transpose(M), M);
    double ev[3], evec[9];
    sym_eigen_jacobi(MtM.m, 3, ev, evec);                       // ascending
    for (int c = 0; c < 3; ++c) {
        const int src = 2 - c;
        s[c] = std::sqrt(std::max(ev[src], 0.0));
        for (int r = 0; r < 3; ++r) V.m[r * 3 + c] = evec[r * 3 + src];
    }
    for (int c = 0; c < 2; ++c) {
        double u[3] = {0, 0, 0};
        for (int r = 0; r < 3; ++r) u[r] = M.m[r * 3] * V.m[c] + M.m[r * 3 + 1] * V.m[3 + c] + M.m[r * 3 + 2] * V.m[6 + c];
        const double n = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        for (int r = 0; r < 3; ++r) U.m[r * 3 + c] = n > 0 ? u[r] / n : (r == c ? 1.0 : 0.0);
    }
    // second column re-orthogonalised against the first, third = first x second (valid for rank 2 and rank 3 alike up to sign)
    double d01 = U.m[0] * U.m[1] + U.m[3] * U.m[4] + U.m[6] * U.m[7];
    for (int r = 0; r < 3; ++r) U.m[r * 3 + 1] -= d01 * U.m[r * 3];
    double n1 = std::sqrt(U.m[1] * U.m[1] + U.m[4] * U.m[4] + U.m[7] * U.m[7]);
    for (int r = 0; r < 3; ++r) U.m[r * 3 + 1] /= n1;
    U.m[2] = U.m[3] * U.m[7] - U.m[6] * U.m[4];
    U.m[5] = U.m[6] * U.m[1] - U.m[0] * U.m[7];
    U.m[8] = U.m[0] * U.m[4] - U.m[3] * U.m[1];
    // sign of the third column such that M v3 = s3 u3 where s3 is not negligible
    double mv[3];
    for (int r = 0; r < 3; ++r) mv[r] = M.m[r * 3] * V.m[2] + M.m[r * 3 + 1] * V.m[5] + M.m[r * 3 + 2] * V.m[8];
    if (mv[0] * U.m[2] + mv[1] * U.m[5] + mv[2] * U.m[8] < 0) for (int r = 0; r < 3; ++r) U.m[r * 3 + 2] = -U.m[r * 3 + 2];
}

// Hartley normalisation as ORB-SLAM / OpenVSLAM do it: centroid to the origin, mean absolute deviation 1 per axis
void normalize_points(const std::vector<double>& in, std::vector<double>& out, M3& T)
{
    const size_t n = in.size() / 2;
    double mx = 0, my = 0;
    for (size_t i = 0; i < n; ++i) { mx += in[2 * i]; my += in[2 * i + 1]; }
    mx /= (double)n; my /= (double)n;
    double dx = 0, dy = 0;
    out.resize(in.size());
    for (size_t i = 0; i < n; ++i) { out[2 * i] = in[2 * i] - mx; out[2 * i + 1] = in[2 * i + 1] - my; dx += std::fabs(out[2 * i]); dy += std::fabs(out[2 * i + 1]); }
    dx /= (double)n; dy /= (double)n;
    const double sx = 1.0 / dx, sy = 1.0 / dy;
    for (size_t i = 0; i < n; ++i) { out[2 * i] *= sx; out[2 * i + 1] *= sy; }
    T = M3{{sx, 0, -mx * sx, 0, sy, -my * sy, 0, 0, 1}};
}

struct Rng {
    uint32_t s;
    uint32_t next() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
};

double check_homography(const M3& H21, const M3& H12, const std::vector<double>& p1, const std::vector<double>& p2, double sigma, std::vector<uint8_t>& inl)
{
    const size_t n = p1.size() / 2;
    const double th = 5.991, inv_s2 = 1.0 / (sigma * sigma);
    double score = 0;
    inl.assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
        const double u1 = p1[2 * i], v1 = p1[2 * i + 1], u2 = p2[2 * i], v2 = p2[2 * i + 1];
        bool in = true;
        {   // second image point into the first
            const double w = 1.0 / (H12.m[6] * u2 + H12.m[7] * v2 + H12.m[8]);
            const double x = (H12.m[0] * u2 + H12.m[1] * v2 + H12.m[2]) * w, y = (H12.m[3] * u2 + H12.m[4] * v2 + H12.m[5]) * w;
            const double c = ((u1 - x) * (u1 - x) + (v1 - y) * (v1 - y)) * inv_s2;
            if (c > th) in = false; else score += th - c;
        }
        {   // first image point into the second
            const double w = 1.0 / (H21.m[6] * u1 + H21.m[7] * v1 + H21.m[8]);
            const double x = (H21.m[0] * u1 + H21.m[1] * v1 + H21.m[2]) * w, y = (H21.m[3] * u1 + H21.m[4] * v1 + H21.m[5]) * w;
            const double c = ((u2 - x) * (u2 - x) + (v2 - y) * (v2 - y)) * inv_s2;
            if (c > th) in = false; else score += th - c;
        }
        inl[i] = in ? 1 : 0;
    }
    return score;
}

double check_fundamental(const M3& F21, const std::vector<double>& p1, const std::vector<double>& p2, double sigma, std::vector<uint8_t>& inl)
{
    const size_t n = p1.size() / 2;
    const double th = 3.841, th_score = 5.991, inv_s2 = 1.0 / (sigma * sigma);
    double score = 0;
    inl.assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
        const double u1 = p1[2 * i], v1 = p1[2 * i + 1], u2 = p2[2 * i], v2 = p2[2 * i + 1];
        bool in = true;
        {   // epipolar line of x1 in the second image: l2 = F21 x1
            const double a = F21.m[0] * u1 + F21.m[1] * v1 + F21.m[2], b = F21.m[3] * u1 + F21.m[4] * v1 + F21.m[5], c = F21.m[6] * u1 + F21.m[7] * v1 + F21.m[8];
            const double num = a * u2 + b * v2 + c;
            const double chi = num * num / (a * a + b * b) * inv_s2;
            if (chi > th) in = false; else score += th_score - chi;
        }
        {   // epipolar line of x2 in the first image: l1 = F21^T x2
            const double a = F21.m[0] * u2 + F21.m[3] * v2 + F21.m[6], b = F21.m[1] * u2 + F21.m[4] * v2 + F21.m[7], c = F21.m[2] * u2 + F21.m[5] * v2 + F21.m[8];
            const double num = a * u1 + b * v1 + c;
            const double chi = num * num / (a * a + b * b) * inv_s2;
            if (chi > th) in = false; else score += th_score - chi;
        }
        inl[i] = in ? 1 : 0;
    }
    return score;
}

struct Hypothesis { M3 R; double t[3]; };

// triangulates the inlier matches under (R, t), counts the plausible ones (ORB-SLAM CheckRT / OpenVSLAM check_triangulated_pts)
int check_pose(const Hypothesis& h, const double* K, const std::vector<double>& p1, const std::vector<double>& p2, const std::vector<uint8_t>& inl,
               double th2, std::vector<double>& pts, std::vector<uint8_t>& good, double& parallax_deg)
{
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const size_t n = p1.size() / 2;
    double P1[12] = {fx, 0, cx, 0, 0, fy, cy, 0, 0, 0, 1, 0}, P2[12];
    for (int c = 0; c < 3; ++c) {
        P2[c] = fx * h.R.m[c] + cx * h.R.m[6 + c]; P2[4 + c] = fy * h.R.m[3 + c] + cy * h.R.m[6 + c]; P2[8 + c] = h.R.m[6 + c];
    }
    P2[3] = fx * h.t[0] + cx * h.t[2]; P2[7] = fy * h.t[1] + cy * h.t[2]; P2[11] = h.t[2];
    const double O2[3] = {-(h.R.m[0] * h.t[0] + h.R.m[3] * h.t[1] + h.R.m[6] * h.t[2]), -(h.R.m[1] * h.t[0] + h.R.m[4] * h.t[1] + h.R.m[7] * h.t[2]),
                          -(h.R.m[2] * h.t[0] + h.R.m[5] * h.t[1] + h.R.m[8] * h.t[2])};
    pts.assign(3 * n, std::numeric_limits<double>::quiet_NaN());
    good.assign(n, 0);
    std::vector<double> cosines;
    int n_good = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!inl[i]) continue;
        double X[3];
        if (!triangulate_point(P1, P2, &p1[2 * i], &p2[2 * i], X)) continue;
        if (!std::isfinite(X[0]) || !std::isfinite(X[1]) || !std::isfinite(X[2])) continue;
        const double n1 = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
        const double d2[3] = {X[0] - O2[0], X[1] - O2[1], X[2] - O2[2]};
        const double n2 = std::sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
        const double cosp = (X[0] * d2[0] + X[1] * d2[1] + X[2] * d2[2]) / (n1 * n2);
        if (X[2] <= 0 && cosp < 0.99998) continue;
        const double X2[3] = {h.R.m[0] * X[0] + h.R.m[1] * X[1] + h.R.m[2] * X[2] + h.t[0], h.R.m[3] * X[0] + h.R.m[4] * X[1] + h.R.m[5] * X[2] + h.t[1],
                              h.R.m[6] * X[0] + h.R.m[7] * X[1] + h.R.m[8] * X[2] + h.t[2]};
        if (X2[2] <= 0 && cosp < 0.99998) continue;
        const double e1x = fx * X[0] / X[2] + cx - p1[2 * i], e1y = fy * X[1] / X[2] + cy - p1[2 * i + 1];
        if (e1x * e1x + e1y * e1y > th2) continue;
        const double e2x = fx * X2[0] / X2[2] + cx - p2[2 * i], e2y = fy * X2[1] / X2[2] + cy - p2[2 * i + 1];
        if (e2x * e2x + e2y * e2y > th2) continue;
        cosines.push_back(cosp);
        pts[3 * i] = X[0]; pts[3 * i + 1] = X[1]; pts[3 * i + 2] = X[2];
        ++n_good;
        if (cosp < 0.99998) good[i] = 1;
    }
    parallax_deg = 0;
    if (!cosines.empty()) {
        std::sort(cosines.begin(), cosines.end());
        const size_t idx = std::min<size_t>(50, cosines.size() - 1);
        parallax_deg = std::acos(std::min(1.0, std::max(-1.0, cosines[idx]))) * 180.0 / M_PI;
    }
    return n_good;
}

void hypotheses_from_fundamental(const M3& F21, const double* K, std::vector<Hypothesis>& out)
{
    const M3 Km{{K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1}};
    const M3 E = mul(mul(transpose(Km), F21), Km);
    M3 U, V; double s[3];
    svd3(E, U, s, V);
    if (det(U) < 0) for (int r = 0; r < 3; ++r) U.m[r * 3 + 2] = -U.m[r * 3 + 2];
    if (det(V) < 0) for (int r = 0; r < 3; ++r) V.m[r * 3 + 2] = -V.m[r * 3 + 2];
    const M3 W{{0, -1, 0, 1, 0, 0, 0, 0, 1}};
    M3 R1 = mul(mul(U, W), transpose(V)), R2 = mul(mul(U, transpose(W)), transpose(V));
    if (det(R1) < 0) for (double& v : R1.m) v = -v;
    if (det(R2) < 0) for (double& v : R2.m) v = -v;
    double t[3] = {U.m[2], U.m[5], U.m[8]};
    const double nt = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    for (double& v : t) v /= nt;
    out.clear();
    out.push_back({R1, {t[0], t[1], t[2]}}); out.push_back({R2, {t[0], t[1], t[2]}});
    out.push_back({R1, {-t[0], -t[1], -t[2]}}); out.push_back({R2, {-t[0], -t[1], -t[2]}});
}

// Faugeras & Lustman, "Motion and structure from motion in a piecewise planar environment" (1988): 8 motion hypotheses
bool hypotheses_from_homography(const M3& H21, const double* K, std::vector<Hypothesis>& out)
{
    const M3 Km{{K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1}};
    const M3 A = mul(mul(inverse(Km), H21), Km);
    M3 U, V; double w[3];
    svd3(A, U, w, V);
    const double s = det(U) * det(V);
    const double d1 = w[0], d2 = w[1], d3 = w[2];
    out.clear();
    if (d1 / d2 < 1.00001 || d2 / d3 < 1.00001) return false;
    const double aux1 = std::sqrt((d1 * d1 - d2 * d2) / (d1 * d1 - d3 * d3)), aux3 = std::sqrt((d2 * d2 - d3 * d3) / (d1 * d1 - d3 * d3));
    const double x1[4] = {aux1, aux1, -aux1, -aux1}, x3[4] = {aux3, -aux3, aux3, -aux3};
    {   // d' = d2
        const double aux_s = std::sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 + d3) * d2);
        const double ct = (d2 * d2 + d1 * d3) / ((d1 + d3) * d2);
        const double st[4] = {aux_s, -aux_s, -aux_s, aux_s};
        for (int i = 0; i < 4; ++i) {
            const M3 Rp{{ct, 0, -st[i], 0, 1, 0, st[i], 0, ct}};
            M3 R = mul(mul(U, Rp), transpose(V));
            for (double& v : R.m) v *= s;
            const double tp[3] = {x1[i] * (d1 - d3), 0, -x3[i] * (d1 - d3)};
            double t[3];
            for (int r = 0; r < 3; ++r) t[r] = U.m[r * 3] * tp[0] + U.m[r * 3 + 1] * tp[1] + U.m[r * 3 + 2] * tp[2];
            const double nt = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
            out.push_back({R, {t[0] / nt, t[1] / nt, t[2] / nt}});
        }
    }
    {   // d' = -d2
        const double aux_s = std::sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 - d3) * d2);
        const double cp = (d1 * d3 - d2 * d2) / ((d1 - d3) * d2);
        const double sp[4] = {aux_s, -aux_s, -aux_s, aux_s};
        for (int i = 0; i < 4; ++i) {
            const M3 Rp{{cp, 0, sp[i], 0, -1, 0, sp[i], 0, -cp}};
            M3 R = mul(mul(U, Rp), transpose(V));
            for (double& v : R.m) v *= s;
            const double tp[3] = {x1[i] * (d1 + d3), 0, x3[i] * (d1 + d3)};
            double t[3];
            for (int r = 0; r < 3; ++r) t[r] = U.m[r * 3] * tp[0] + U.m[r * 3 + 1] * tp[1] + U.m[r * 3 + 2] * tp[2];
            const double nt = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
            out.push_back({R, {t[0] / nt, t[1] / nt, t[2] / nt}});
        }
    }
    return true;
}

}  // namespace

void sym_eigen_jacobi(const double* A, int n, double* eigval, double* eigvec)
{
    std::vector<double> a(A, A + n * n), v((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) (i == j ? diag : off) += a[(size_t)i * n + j] * a[(size_t)i * n + j];
        if (off <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
            const double apq = a[(size_t)p * n + q];
            if (apq == 0.0) continue;
            const double theta = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < n; ++k) {                   // columns p, q
                const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                a[(size_t)k * n + p] = c * akp - s * akq; a[(size_t)k * n + q] = s * akp + c * akq;
            }
            for (int k = 0; k < n; ++k) {                   // rows p, q
                const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                a[(size_t)p * n + k] = c * apk - s * aqk; a[(size_t)q * n + k] = s * apk + c * aqk;
            }
            for (int k = 0; k < n; ++k) {
                const double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
                v[(size_t)k * n + p] = c * vkp - s * vkq; v[(size_t)k * n + q] = s * vkp + c * vkq;
            }
        }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return a[(size_t)x * n + x] < a[(size_t)y * n + y]; });
    for (int c = 0; c < n; ++c) {
        eigval[c] = a[(size_t)order[c] * n + order[c]];
        for (int r = 0; r < n; ++r) eigvec[(size_t)r * n + c] = v[(size_t)r * n + order[c]];
    }
}

void homography_from_matches(const double* x1, const double* x2, int n, double* H)
{
    double AtA[81] = {0};
    for (int i = 0; i < n; ++i) {
        const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        const double r1[9] = {0, 0, 0, -u1, -v1, -1, v2 * u1, v2 * v1, v2};
        const double r2[9] = {u1, v1, 1, 0, 0, 0, -u2 * u1, -u2 * v1, -u2};
        for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) AtA[a * 9 + b] += r1[a] * r1[b] + r2[a] * r2[b];
    }
    double ev[9], evec[81];
    sym_eigen_jacobi(AtA, 9, ev, evec);
    for (int k = 0; k < 9; ++k) H[k] = evec[k * 9];
}

void fundamental_from_matches(const double* x1, const double* x2, int n, double* F)
{
    double AtA[81] = {0};
    for (int i = 0; i < n; ++i) {
        const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        const double r[9] = {u2 * u1, u2 * v1, u2, v2 * u1, v2 * v1, v2, u1, v1, 1};
        for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) AtA[a * 9 + b] += r[a] * r[b];
    }
    double ev[9], evec[81];
    sym_eigen_jacobi(AtA, 9, ev, evec);
    M3 Fp;
    for (int k = 0; k < 9; ++k) Fp.m[k] = evec[k * 9];
    // rank 2: F <- F (I - v3 v3^T), v3 the right singular vector of the smallest singular value
    const M3 FtF = mul(transpose(Fp), Fp);
    double e3[3], v3[9];
    sym_eigen_jacobi(FtF.m, 3, e3, v3);
    const double v[3] = {v3[0], v3[3], v3[6]};
    for (int r = 0; r < 3; ++r) {
        const double fv = Fp.m[r * 3] * v[0] + Fp.m[r * 3 + 1] * v[1] + Fp.m[r * 3 + 2] * v[2];
        for (int c = 0; c < 3; ++c) F[r * 3 + c] = Fp.m[r * 3 + c] - fv * v[c];
    }
}

bool triangulate_point(const double* P1, const double* P2, const double* x1, const double* x2, double* X)
{
    double A[16];
    for (int c = 0; c < 4; ++c) {
        A[c] = x1[0] * P1[8 + c] - P1[c];
        A[4 + c] = x1[1] * P1[8 + c] - P1[4 + c];
        A[8 + c] = x2[0] * P2[8 + c] - P2[c];
        A[12 + c] = x2[1] * P2[8 + c] - P2[4 + c];
    }
    double AtA[16] = {0};
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) AtA[a * 4 + b] += A[r * 4 + a] * A[r * 4 + b];
    double ev[4], evec[16];
    sym_eigen_jacobi(AtA, 4, ev, evec);
    const double w = evec[12];
    if (w == 0.0) return false;
    X[0] = evec[0] / w; X[1] = evec[4] / w; X[2] = evec[8] / w;
    return true;
}

bool two_view_initialize(const double* K, const float* kp_ref, const float* kp_cur, const int32_t* matches, int n_matches,
                         const TwoViewParams& prm, TwoViewResult& out)
{
    out = TwoViewResult();
    out.inlier.assign((size_t)std::max(n_matches, 0), 0);
    out.triangulated.assign((size_t)std::max(n_matches, 0), 0);
    out.points.assign(3 * (size_t)std::max(n_matches, 0), std::numeric_limits<double>::quiet_NaN());
    if (n_matches < 8) return false;
    const size_t n = (size_t)n_matches;
    std::vector<double> p1(2 * n), p2(2 * n);
    for (size_t i = 0; i < n; ++i) {
        p1[2 * i] = kp_ref[2 * matches[2 * i]]; p1[2 * i + 1] = kp_ref[2 * matches[2 * i] + 1];
        p2[2 * i] = kp_cur[2 * matches[2 * i + 1]]; p2[2 * i + 1] = kp_cur[2 * matches[2 * i + 1] + 1];
    }
    std::vector<double> n1, n2;
    M3 T1, T2;
    normalize_points(p1, n1, T1);
    normalize_points(p2, n2, T2);
    const M3 T2inv = inverse(T2), T2t = transpose(T2);

    // the same 8-match samples serve both models (ORB-SLAM draws them once); sampling without replacement
    Rng rng{prm.seed ? prm.seed : 1u};
    std::vector<uint8_t> inl, best_inl_h(n, 0), best_inl_f(n, 0);
    double best_h = -1, best_f = -1;
    M3 best_H{}, best_F{};
    std::vector<int> avail(n);
    for (int it = 0; it < prm.ransac_iters; ++it) {
        for (size_t i = 0; i < n; ++i) avail[i] = (int)i;
        double s1[16], s2[16];
        size_t left = n;
        for (int k = 0; k < 8; ++k) {
            const size_t r = rng.next() % left;
            const int idx = avail[r];
            avail[r] = avail[left - 1]; --left;
            s1[2 * k] = n1[2 * idx]; s1[2 * k + 1] = n1[2 * idx + 1]; s2[2 * k] = n2[2 * idx]; s2[2 * k + 1] = n2[2 * idx + 1];
        }
        M3 Hn, Fn;
        homography_from_matches(s1, s2, 8, Hn.m);
        fundamental_from_matches(s1, s2, 8, Fn.m);
        const M3 H21 = mul(mul(T2inv, Hn), T1);
        const M3 F21 = mul(mul(T2t, Fn), T1);
        if (std::fabs(det(H21)) > 1e-300) {
            const double sh = check_homography(H21, inverse(H21), p1, p2, prm.sigma, inl);
            if (sh > best_h) { best_h = sh; best_H = H21; best_inl_h = inl; }
        }
        const double sf = check_fundamental(F21, p1, p2, prm.sigma, inl);
        if (sf > best_f) { best_f = sf; best_F = F21; best_inl_f = inl; }
    }
    // [UPSTREAM] find_via_ransac(..., recompute = true): the best model of each kind is estimated again from all its inliers
    // (normalised coordinates) and its inliers and score are taken from that estimate
    auto gather = [&](const std::vector<uint8_t>& mask, std::vector<double>& a, std::vector<double>& b) {
        a.clear(); b.clear();
        for (size_t i = 0; i < n; ++i) if (mask[i]) { a.push_back(n1[2 * i]); a.push_back(n1[2 * i + 1]); b.push_back(n2[2 * i]); b.push_back(n2[2 * i + 1]); }
    };
    std::vector<double> ga, gb;
    if (best_h >= 0) {
        gather(best_inl_h, ga, gb);
        if (ga.size() >= 16) {
            M3 Hn; homography_from_matches(ga.data(), gb.data(), (int)(ga.size() / 2), Hn.m);
            const M3 H21 = mul(mul(T2inv, Hn), T1);
            if (std::fabs(det(H21)) > 1e-300) { best_h = check_homography(H21, inverse(H21), p1, p2, prm.sigma, inl); best_H = H21; best_inl_h = inl; }
        }
    }
    if (best_f >= 0) {
        gather(best_inl_f, ga, gb);
        if (ga.size() >= 16) {
            M3 Fn; fundamental_from_matches(ga.data(), gb.data(), (int)(ga.size() / 2), Fn.m);
            const M3 F21 = mul(mul(T2t, Fn), T1);
            best_f = check_fundamental(F21, p1, p2, prm.sigma, inl); best_F = F21; best_inl_f = inl;
        }
    }
    out.score_h = std::max(best_h, 0.0); out.score_f = std::max(best_f, 0.0);
    std::memcpy(out.H, best_H.m, sizeof(out.H)); std::memcpy(out.F, best_F.m, sizeof(out.F));
    if (out.score_h + out.score_f <= 0) return false;
    const bool use_h = out.score_h / (out.score_h + out.score_f) > 0.40;
    out.model = use_h ? 0 : 1;
    const std::vector<uint8_t>& model_inl = use_h ? best_inl_h : best_inl_f;
    out.inlier = model_inl;
    out.n_inliers = (int)std::count(model_inl.begin(), model_inl.end(), (uint8_t)1);

    std::vector<Hypothesis> hyps;
    if (use_h) { if (!hypotheses_from_homography(best_H, K, hyps)) return false; }
    else hypotheses_from_fundamental(best_F, K, hyps);

    const double th2 = prm.reproj_err_thr * prm.sigma * prm.sigma;       // ORB-SLAM: 4 sigma^2
    int best_good = 0, second_good = 0, best_i = -1;
    double best_parallax = 0;
    std::vector<double> pts, best_pts;
    std::vector<uint8_t> good, best_goodmask;
    for (size_t i = 0; i < hyps.size(); ++i) {
        double parallax = 0;
        const int ng = check_pose(hyps[i], K, p1, p2, model_inl, th2, pts, good, parallax);
        if (ng > best_good) { second_good = best_good; best_good = ng; best_i = (int)i; best_parallax = parallax; best_pts = pts; best_goodmask = good; }
        else if (ng > second_good) second_good = ng;
    }
    out.n_valid = best_good; out.parallax_deg = best_parallax;
    // one clear winner with enough plausible points and enough parallax ([UPSTREAM] initialize::base::find_most_plausible_pose)
    const int min_valid = std::max(prm.min_triangulated, (int)(0.9 * out.n_inliers));
    if (best_i < 0 || best_good < min_valid) return false;
    if ((double)second_good > 0.8 * (double)best_good) return false;
    if (best_parallax < prm.parallax_deg_thr) return false;
    std::memcpy(out.R, hyps[(size_t)best_i].R.m, sizeof(out.R));
    std::memcpy(out.t, hyps[(size_t)best_i].t, sizeof(out.t));
    out.points = best_pts; out.triangulated = best_goodmask;
    out.ok = true;
    return true;
}


bool horn_absolute_orientation(const double* x1, const double* x2, int n, bool fix_scale, double* R, double* t, double* s_out)
{
    if (n < 3) return false;
    double o1[3] = {0, 0, 0}, o2[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) { o1[a] += x1[3 * i + a]; o2[a] += x2[3 * i + a]; }
    for (int a = 0; a < 3; ++a) { o1[a] /= n; o2[a] /= n; }
    double M[9] = {0};                                   // M[a][b] = sum (x2 - o2)[a] (x1 - o1)[b]
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[a * 3 + b] += (x2[3 * i + a] - o2[a]) * (x1[3 * i + b] - o1[b]);
    const double N[16] = {M[0] + M[4] + M[8], M[5] - M[7], M[6] - M[2], M[1] - M[3],
                          M[5] - M[7], M[0] - M[4] - M[8], M[1] + M[3], M[6] + M[2],
                          M[6] - M[2], M[1] + M[3], -M[0] + M[4] - M[8], M[5] + M[7],
                          M[1] - M[3], M[6] + M[2], M[5] + M[7], -M[0] - M[4] + M[8]};
    double ev[4], evec[16];
    sym_eigen_jacobi(N, 4, ev, evec);                    // ascending: the last column belongs to the largest eigenvalue
    double q[4] = {evec[3], evec[7], evec[11], evec[15]};
    const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(qn > 0)) return false;
    for (double& v : q) v /= qn;
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double Rm[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                          2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                          2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    double sc = 1.0;
    if (!fix_scale) {
        double nom = 0, den = 0;
        for (int i = 0; i < n; ++i) {
            double p3[3];
            for (int r = 0; r < 3; ++r) p3[r] = Rm[r * 3] * (x2[3 * i] - o2[0]) + Rm[r * 3 + 1] * (x2[3 * i + 1] - o2[1]) + Rm[r * 3 + 2] * (x2[3 * i + 2] - o2[2]);
            for (int r = 0; r < 3; ++r) { nom += (x1[3 * i + r] - o1[r]) * p3[r]; den += p3[r] * p3[r]; }
        }
        if (!(den > 0) || !(nom > 0)) return false;
        sc = nom / den;
    }
    for (int k = 0; k < 9; ++k) R[k] = Rm[k];
    for (int r = 0; r < 3; ++r) t[r] = o1[r] - sc * (Rm[r * 3] * o2[0] + Rm[r * 3 + 1] * o2[1] + Rm[r * 3 + 2] * o2[2]);
    *s_out = sc;
    return true;
}

// rotation matrix (row major) -> quaternion w x y z (Eigen::Quaterniond(R))
static void rot_to_quat(const double* R, double* q)
{
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) { const double s4 = std::sqrt(tr + 1.0) * 2; q[0] = 0.25 * s4; q[1] = (R[7] - R[5]) / s4; q[2] = (R[2] - R[6]) / s4; q[3] = (R[3] - R[1]) / s4; }
    else if (R[0] > R[4] && R[0] > R[8]) { const double s4 = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[0] = (R[7] - R[5]) / s4; q[1] = 0.25 * s4; q[2] = (R[1] + R[3]) / s4; q[3] = (R[2] + R[6]) / s4; }
    else if (R[4] > R[8]) { const double s4 = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[0] = (R[2] - R[6]) / s4; q[1] = (R[1] + R[3]) / s4; q[2] = 0.25 * s4; q[3] = (R[5] + R[7]) / s4; }
    else { const double s4 = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[0] = (R[3] - R[1]) / s4; q[1] = (R[2] + R[6]) / s4; q[2] = (R[5] + R[7]) / s4; q[3] = 0.25 * s4; }
}

int sim3_solve_ransac(const double* p1c, const double* p2c, const double* obs1, const double* obs2, const double* inv_sigma2_1, const double* inv_sigma2_2,
                      int n, const double* cam1, const double* cam2, bool fix_scale, int iterations, uint32_t seed, double* s12, uint8_t* inlier)
{
    if (n < 3) return 0;
    Rng rng{seed ? seed : 1u};
    std::vector<int> avail((size_t)n);
    std::vector<uint8_t> cur((size_t)n);
    int best = 0;
    for (int it = 0; it < iterations; ++it) {
        for (int i = 0; i < n; ++i) avail[(size_t)i] = i;
        int left = n, idx[3];
        for (int k = 0; k < 3; ++k) { const int r = (int)(rng.next() % (uint32_t)left); idx[k] = avail[(size_t)r]; avail[(size_t)r] = avail[(size_t)left - 1]; --left; }
        double a1[9], a2[9], R[9], t[3], sc;
        for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) { a1[3 * k + a] = p1c[3 * idx[k] + a]; a2[3 * k + a] = p2c[3 * idx[k] + a]; }
        if (!horn_absolute_orientation(a1, a2, 3, fix_scale, R, t, &sc)) continue;
        int count = 0;
        for (int i = 0; i < n; ++i) {
            cur[(size_t)i] = 0;
            // 2 -> 1: x1 = s R x2 + t;  1 -> 2: x2 = R^T (x1 - t) / s
            double q1[3], q2[3];
            for (int r = 0; r < 3; ++r) q1[r] = sc * (R[r * 3] * p2c[3 * i] + R[r * 3 + 1] * p2c[3 * i + 1] + R[r * 3 + 2] * p2c[3 * i + 2]) + t[r];
            const double d[3] = {p1c[3 * i] - t[0], p1c[3 * i + 1] - t[1], p1c[3 * i + 2] - t[2]};
            for (int r = 0; r < 3; ++r) q2[r] = (R[r] * d[0] + R[3 + r] * d[1] + R[6 + r] * d[2]) / sc;
            if (!(q1[2] > 0) || !(q2[2] > 0)) continue;
            const double e1x = cam1[0] * q1[0] / q1[2] + cam1[2] - obs1[2 * i], e1y = cam1[1] * q1[1] / q1[2] + cam1[3] - obs1[2 * i + 1];
            const double e2x = cam2[0] * q2[0] / q2[2] + cam2[2] - obs2[2 * i], e2y = cam2[1] * q2[1] / q2[2] + cam2[3] - obs2[2 * i + 1];
            if ((e1x * e1x + e1y * e1y) * inv_sigma2_1[i] < 9.210 && (e2x * e2x + e2y * e2y) * inv_sigma2_2[i] < 9.210) { cur[(size_t)i] = 1; ++count; }
        }
        if (count > best) {
            best = count;
            double q[4];
            rot_to_quat(R, q);
            s12[0] = q[0]; s12[1] = q[1]; s12[2] = q[2]; s12[3] = q[3]; s12[4] = t[0]; s12[5] = t[1]; s12[6] = t[2]; s12[7] = sc;
            if (inlier) std::copy(cur.begin(), cur.end(), inlier);
        }
    }
    return best;
}


// ---- [UPSTREAM] solve::pnp_solver (relocalisation without a pose prior) -------------------------------------------------------------
// EPnP (Lepetit, Moreno-Noguer, Fua: "EPnP: An Accurate O(n) Solution to the PnP Problem", IJCV 2009) inside a RANSAC over 4-match
// samples, then a refit on the inliers of the best sample -- the structure of OpenVSLAM's solver (which follows ORB-SLAM's PnPsolver,
// itself the authors' reference implementation).  The algorithm, as published:
//   1. four control points in the world: the centroid and the centroid plus the principal directions scaled by sqrt(eigenvalue / n);
//   2. every landmark as a barycentric combination of them (alphas, they sum to one);
//   3. the projection equations are linear in the 12 camera-frame coordinates of the control points: M x = 0 (2 n x 12); the solution is
//      a combination x = sum beta_k v_k of the eigenvectors of M^T M with the four smallest eigenvalues;
//   4. the betas from the six pairwise distances of the control points, which the camera frame must preserve (L_6x10 beta_ij = rho):
//      three linearised guesses (N = 1 .. 3 null vectors: approx_1 / _2 / _3), each refined by five Gauss-Newton steps;
//   5. per guess the camera-frame landmarks, the sign that puts them in front of the camera, the rigid motion world -> camera (Horn's
//      absolute orientation here; the reference implementation takes the SVD form), and its mean reprojection error: the best wins.
// Inliers are counted by the reprojection error (chi-square 5.991 at the keypoint's level).  The pose optimiser on the device refines
// the result, as upstream refines EPnP's.  The linear algebra is this file's Jacobi eigen-solver and small Gaussian eliminations,
// written so that the tests' numpy restatement can follow it operation by operation (both sides then test the same hypotheses).
namespace {
// least squares A x = b (m x k, k <= 5) by the normal equations and Gaussian elimination with partial pivoting; false when singular
bool lsq_small(const double* A, const double* b, int m, int k, double* x)
{
    double N[5 * 6];
    for (int i = 0; i < k; ++i) {
        for (int j = 0; j < k; ++j) { double s = 0; for (int r = 0; r < m; ++r) s += A[r * k + i] * A[r * k + j]; N[i * (k + 1) + j] = s; }
        double s = 0;
        for (int r = 0; r < m; ++r) s += A[r * k + i] * b[r];
        N[i * (k + 1) + k] = s;
    }
    for (int c = 0; c < k; ++c) {
        int piv = c;
        for (int r = c + 1; r < k; ++r) if (std::fabs(N[r * (k + 1) + c]) > std::fabs(N[piv * (k + 1) + c])) piv = r;
        if (!(std::fabs(N[piv * (k + 1) + c]) > 1e-300)) return false;
        if (piv != c) for (int j = 0; j <= k; ++j) std::swap(N[c * (k + 1) + j], N[piv * (k + 1) + j]);
        for (int r = c + 1; r < k; ++r) {
            const double f = N[r * (k + 1) + c] / N[c * (k + 1) + c];
            for (int j = c; j <= k; ++j) N[r * (k + 1) + j] -= f * N[c * (k + 1) + j];
        }
    }
    for (int i = k - 1; i >= 0; --i) {
        double s2 = N[i * (k + 1) + k];
        for (int j = i + 1; j < k; ++j) s2 -= N[i * (k + 1) + j] * x[j];
        x[i] = s2 / N[i * (k + 1) + i];
    }
    return true;
}
}  // namespace

bool epnp_solve(const double* pw, const double* uv, int n, const double* cam, double* R, double* t)
{
    if (n < 4) return false;
    const double fu = cam[0], fv = cam[1], uc = cam[2], vc = cam[3];
    // 1. control points
    double cws[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) cws[0][a] += pw[3 * i + a];
    for (int a = 0; a < 3; ++a) cws[0][a] /= n;
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) C[a * 3 + b2] += (pw[3 * i + a] - cws[0][a]) * (pw[3 * i + b2] - cws[0][b2]);
    double ev[3], evec[9];
    sym_eigen_jacobi(C, 3, ev, evec);                       // ascending; the principal direction first, as the published code orders them
    for (int j = 1; j <= 3; ++j) {
        const int col = 3 - j;
        const double k = std::sqrt(std::max(ev[col], 0.0) / n);
        for (int a = 0; a < 3; ++a) cws[j][a] = cws[0][a] + k * evec[a * 3 + col];
    }
    // 2. barycentric coordinates: [c1 - c0, c2 - c0, c3 - c0] a = p - c0
    double CC[9], CI[9];
    for (int a = 0; a < 3; ++a) for (int j = 1; j <= 3; ++j) CC[a * 3 + (j - 1)] = cws[j][a] - cws[0][a];
    const double det = CC[0] * (CC[4] * CC[8] - CC[5] * CC[7]) - CC[1] * (CC[3] * CC[8] - CC[5] * CC[6]) + CC[2] * (CC[3] * CC[7] - CC[4] * CC[6]);
    if (!(std::fabs(det) > 1e-300)) return false;           // landmarks in a plane or on a line: no volume to span
    CI[0] = (CC[4] * CC[8] - CC[5] * CC[7]) / det; CI[1] = (CC[2] * CC[7] - CC[1] * CC[8]) / det; CI[2] = (CC[1] * CC[5] - CC[2] * CC[4]) / det;
    CI[3] = (CC[5] * CC[6] - CC[3] * CC[8]) / det; CI[4] = (CC[0] * CC[8] - CC[2] * CC[6]) / det; CI[5] = (CC[2] * CC[3] - CC[0] * CC[5]) / det;
    CI[6] = (CC[3] * CC[7] - CC[4] * CC[6]) / det; CI[7] = (CC[1] * CC[6] - CC[0] * CC[7]) / det; CI[8] = (CC[0] * CC[4] - CC[1] * CC[3]) / det;
    std::vector<double> al((size_t)4 * n);
    for (int i = 0; i < n; ++i) {
        const double d0 = pw[3 * i] - cws[0][0], d1 = pw[3 * i + 1] - cws[0][1], d2 = pw[3 * i + 2] - cws[0][2];
        for (int j = 0; j < 3; ++j) al[4 * (size_t)i + 1 + j] = CI[j * 3] * d0 + CI[j * 3 + 1] * d1 + CI[j * 3 + 2] * d2;
        al[4 * (size_t)i] = 1.0 - al[4 * (size_t)i + 1] - al[4 * (size_t)i + 2] - al[4 * (size_t)i + 3];
    }
    // 3. M^T M, accumulated row pair by row pair
    double MtM[144];
    for (double& v : MtM) v = 0;
    for (int i = 0; i < n; ++i) {
        double r1[12], r2[12];
        for (int j = 0; j < 4; ++j) {
            const double a = al[4 * (size_t)i + j];
            r1[3 * j] = a * fu; r1[3 * j + 1] = 0.0;    r1[3 * j + 2] = a * (uc - uv[2 * i]);
            r2[3 * j] = 0.0;    r2[3 * j + 1] = a * fv; r2[3 * j + 2] = a * (vc - uv[2 * i + 1]);
        }
        for (int a = 0; a < 12; ++a) for (int b2 = 0; b2 < 12; ++b2) MtM[a * 12 + b2] += r1[a] * r1[b2] + r2[a] * r2[b2];
    }
    double mev[12], mvec[144];
    sym_eigen_jacobi(MtM, 12, mev, mvec);                   // ascending: columns 0 .. 3 span the (approximate) null space
    double v[4][12];
    for (int k = 0; k < 4; ++k) for (int a = 0; a < 12; ++a) v[k][a] = mvec[a * 12 + k];
    // 4. distance constraints
    static const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    double dv[4][6][3], L[60], rho[6];
    for (int k = 0; k < 4; ++k) for (int p = 0; p < 6; ++p) for (int a = 0; a < 3; ++a) dv[k][p][a] = v[k][3 * pa[p] + a] - v[k][3 * pb[p] + a];
    auto dot3 = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
    for (int p = 0; p < 6; ++p) {
        double* row = L + 10 * p;
        row[0] = dot3(dv[0][p], dv[0][p]); row[1] = 2.0 * dot3(dv[0][p], dv[1][p]); row[2] = dot3(dv[1][p], dv[1][p]);
        row[3] = 2.0 * dot3(dv[0][p], dv[2][p]); row[4] = 2.0 * dot3(dv[1][p], dv[2][p]); row[5] = dot3(dv[2][p], dv[2][p]);
        row[6] = 2.0 * dot3(dv[0][p], dv[3][p]); row[7] = 2.0 * dot3(dv[1][p], dv[3][p]); row[8] = 2.0 * dot3(dv[2][p], dv[3][p]);
        row[9] = dot3(dv[3][p], dv[3][p]);
        double d2 = 0;
        for (int a = 0; a < 3; ++a) { const double d = cws[pa[p]][a] - cws[pb[p]][a]; d2 += d * d; }
        rho[p] = d2;
    }
    double best_err = 1e300;
    bool found = false;
    std::vector<double> pc((size_t)3 * n);
    for (int guess = 0; guess < 3; ++guess) {
        double be[4] = {0, 0, 0, 0};
        if (guess == 0) {                                   // betas 11 12 13 14 (columns 0, 1, 3, 6): N = 4 with the cross terms of beta_1 only
            double A[24], x4[4];
            for (int p = 0; p < 6; ++p) { A[4 * p] = L[10 * p]; A[4 * p + 1] = L[10 * p + 1]; A[4 * p + 2] = L[10 * p + 3]; A[4 * p + 3] = L[10 * p + 6]; }
            if (!lsq_small(A, rho, 6, 4, x4)) continue;
            if (x4[0] < 0) { be[0] = std::sqrt(-x4[0]); be[1] = -x4[1] / be[0]; be[2] = -x4[2] / be[0]; be[3] = -x4[3] / be[0]; }
            else { be[0] = std::sqrt(x4[0]); be[1] = x4[1] / be[0]; be[2] = x4[2] / be[0]; be[3] = x4[3] / be[0]; }
        } else if (guess == 1) {                            // betas 11 12 22 (columns 0, 1, 2): N = 2
            double A[18], x3[3];
            for (int p = 0; p < 6; ++p) { A[3 * p] = L[10 * p]; A[3 * p + 1] = L[10 * p + 1]; A[3 * p + 2] = L[10 * p + 2]; }
            if (!lsq_small(A, rho, 6, 3, x3)) continue;
            if (x3[0] < 0) { be[0] = std::sqrt(-x3[0]); be[1] = x3[2] < 0 ? std::sqrt(-x3[2]) : 0.0; }
            else { be[0] = std::sqrt(x3[0]); be[1] = x3[2] > 0 ? std::sqrt(x3[2]) : 0.0; }
            if (x3[1] < 0) be[0] = -be[0];
        } else {                                            // betas 11 12 22 13 23 (columns 0 .. 4): N = 3
            double A[30], x5[5];
            for (int p = 0; p < 6; ++p) for (int c = 0; c < 5; ++c) A[5 * p + c] = L[10 * p + c];
            if (!lsq_small(A, rho, 6, 5, x5)) continue;
            if (x5[0] < 0) { be[0] = std::sqrt(-x5[0]); be[1] = x5[2] < 0 ? std::sqrt(-x5[2]) : 0.0; }
            else { be[0] = std::sqrt(x5[0]); be[1] = x5[2] > 0 ? std::sqrt(x5[2]) : 0.0; }
            if (x5[1] < 0) be[0] = -be[0];
            be[2] = x5[3] / be[0];
        }
        if (!(be[0] == be[0]) || !std::isfinite(be[0]) || !std::isfinite(be[1]) || !std::isfinite(be[2]) || !std::isfinite(be[3])) continue;
        bool gn_ok = true;
        for (int it = 0; it < 5 && gn_ok; ++it) {           // Gauss-Newton on the six distance equations
            double A[24], r6[6], dx[4];
            for (int p = 0; p < 6; ++p) {
                const double* l = L + 10 * p;
                A[4 * p] = 2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3];
                A[4 * p + 1] = l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3];
                A[4 * p + 2] = l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3];
                A[4 * p + 3] = l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3];
                r6[p] = rho[p] - (l[0] * be[0] * be[0] + l[1] * be[0] * be[1] + l[2] * be[1] * be[1] + l[3] * be[0] * be[2] + l[4] * be[1] * be[2] +
                                  l[5] * be[2] * be[2] + l[6] * be[0] * be[3] + l[7] * be[1] * be[3] + l[8] * be[2] * be[3] + l[9] * be[3] * be[3]);
            }
            if (!lsq_small(A, r6, 6, 4, dx)) { gn_ok = false; break; }
            for (int k = 0; k < 4; ++k) be[k] += dx[k];
        }
        if (!gn_ok || !std::isfinite(be[0]) || !std::isfinite(be[1]) || !std::isfinite(be[2]) || !std::isfinite(be[3])) continue;
        // 5. camera-frame control points and landmarks, sign, rigid motion, reprojection error
        double ccs[4][3];
        for (int j = 0; j < 4; ++j) for (int a = 0; a < 3; ++a) ccs[j][a] = be[0] * v[0][3 * j + a] + be[1] * v[1][3 * j + a] + be[2] * v[2][3 * j + a] + be[3] * v[3][3 * j + a];
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) pc[3 * (size_t)i + a] = al[4 * (size_t)i] * ccs[0][a] + al[4 * (size_t)i + 1] * ccs[1][a] + al[4 * (size_t)i + 2] * ccs[2][a] + al[4 * (size_t)i + 3] * ccs[3][a];
        if (pc[2] < 0) for (double& x : pc) x = -x;         // the first landmark in front of the camera
        double Rg[9], tg[3], sdummy = 1.0;
        if (!horn_absolute_orientation(pc.data(), pw, n, true, Rg, tg, &sdummy)) continue;
        double err = 0;
        bool finite = true;
        for (int i = 0; i < n; ++i) {
            const double* X = pw + 3 * (size_t)i;
            const double xc = Rg[0] * X[0] + Rg[1] * X[1] + Rg[2] * X[2] + tg[0], yc = Rg[3] * X[0] + Rg[4] * X[1] + Rg[5] * X[2] + tg[1], zc = Rg[6] * X[0] + Rg[7] * X[1] + Rg[8] * X[2] + tg[2];
            const double du = uc + fu * xc / zc - uv[2 * i], dv2 = vc + fv * yc / zc - uv[2 * i + 1];
            const double e = std::sqrt(du * du + dv2 * dv2);
            if (!std::isfinite(e)) { finite = false; break; }
            err += e;
        }
        if (!finite) continue;
        err /= n;
        if (err < best_err) { best_err = err; found = true; for (int k = 0; k < 9; ++k) R[k] = Rg[k]; for (int k = 0; k < 3; ++k) t[k] = tg[k]; }
    }
    return found;
}

int pnp_solve_ransac(const double* pw, const double* obs, const double* inv_sigma2, int n, const double* cam, int iterations, uint32_t seed, double* pose7, uint8_t* inlier)
{
    for (int i = 0; i < n; ++i) inlier[i] = 0;
    if (n < 4) return 0;
    Rng rng{seed ? seed : 1u};
    std::vector<int> avail((size_t)n);
    std::vector<uint8_t> cur((size_t)n);
    int best = 0;
    double bestR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, bestT[3] = {0, 0, 0};
    auto count_inliers = [&](const double* R, const double* t, uint8_t* flags) {
        int count = 0;
        for (int i = 0; i < n; ++i) {
            const double* X = pw + 3 * (size_t)i;
            const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
            const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
            const double du = cam[0] * xc / z + cam[2] - obs[2 * i], dv = cam[1] * yc / z + cam[3] - obs[2 * i + 1];
            flags[i] = (z > 0 && (du * du + dv * dv) * inv_sigma2[i] < 5.991) ? 1 : 0;
            count += flags[i];
        }
        return count;
    };
    for (int it = 0; it < iterations; ++it) {
        for (int i = 0; i < n; ++i) avail[(size_t)i] = i;
        int left = n, idx[4];
        for (int k = 0; k < 4; ++k) { const int r = (int)(rng.next() % (uint32_t)left); idx[k] = avail[(size_t)r]; avail[(size_t)r] = avail[(size_t)left - 1]; --left; }
        double p4[12], u4[8], R[9], t[3];
        for (int k = 0; k < 4; ++k) { for (int a = 0; a < 3; ++a) p4[3 * k + a] = pw[3 * (size_t)idx[k] + a]; u4[2 * k] = obs[2 * (size_t)idx[k]]; u4[2 * k + 1] = obs[2 * (size_t)idx[k] + 1]; }
        if (!epnp_solve(p4, u4, 4, cam, R, t)) continue;
        const int count = count_inliers(R, t, cur.data());
        if (count > best) {
            best = count;
            for (int i = 0; i < n; ++i) inlier[i] = cur[(size_t)i];
            for (int k = 0; k < 9; ++k) bestR[k] = R[k];
            for (int k = 0; k < 3; ++k) bestT[k] = t[k];
        }
    }
    if (best < 4) { for (int i = 0; i < n; ++i) inlier[i] = 0; return 0; }
    // refit on the inliers of the best sample ([UPSTREAM] refine): EPnP over all of them, kept when it explains at least as many matches
    {
        std::vector<double> pi, ui;
        for (int i = 0; i < n; ++i) if (inlier[i]) { for (int a = 0; a < 3; ++a) pi.push_back(pw[3 * (size_t)i + a]); ui.push_back(obs[2 * (size_t)i]); ui.push_back(obs[2 * (size_t)i + 1]); }
        double R[9], t[3];
        if (epnp_solve(pi.data(), ui.data(), (int)(ui.size() / 2), cam, R, t)) {
            const int count = count_inliers(R, t, cur.data());
            if (count >= best) {
                best = count;
                for (int i = 0; i < n; ++i) inlier[i] = cur[(size_t)i];
                for (int k = 0; k < 9; ++k) bestR[k] = R[k];
                for (int k = 0; k < 3; ++k) bestT[k] = t[k];
            }
        }
    }
    double q[4];
    rot_to_quat(bestR, q);
    for (int k = 0; k < 4; ++k) pose7[k] = q[k];
    for (int k = 0; k < 3; ++k) pose7[4 + k] = bestT[k];
    return best;
}

}  // namespace LpSlam

// ---- C shim for the ctypes tests ------------------------------------------------------------------------------------------
extern "C" {
__attribute__((visibility("default"))) int lpslam_two_view_initialize(const double* K, const float* kp_ref, const float* kp_cur, const int32_t* matches,
                                                                        int n_matches, double sigma, int ransac_iters, uint32_t seed,
                                                                        double* R9, double* t3, double* H9, double* F9, double* scores2, double* parallax_deg,
                                                                        int32_t* model, uint8_t* inlier, uint8_t* triangulated, double* points3)
{
    LpSlam::TwoViewParams prm;
    prm.sigma = sigma; prm.ransac_iters = ransac_iters; prm.seed = seed;
    LpSlam::TwoViewResult res;
    const bool ok = LpSlam::two_view_initialize(K, kp_ref, kp_cur, matches, n_matches, prm, res);
    if (R9) std::memcpy(R9, res.R, sizeof(res.R));
    if (t3) std::memcpy(t3, res.t, sizeof(res.t));
    if (H9) std::memcpy(H9, res.H, sizeof(res.H));
    if (F9) std::memcpy(F9, res.F, sizeof(res.F));
    if (scores2) { scores2[0] = res.score_h; scores2[1] = res.score_f; }
    if (parallax_deg) *parallax_deg = res.parallax_deg;
    if (model) *model = res.model;
    for (int i = 0; i < n_matches; ++i) {
        if (inlier) inlier[i] = res.inlier.size() > (size_t)i ? res.inlier[(size_t)i] : 0;
        if (triangulated) triangulated[i] = res.triangulated.size() > (size_t)i ? res.triangulated[(size_t)i] : 0;
        if (points3) for (int c = 0; c < 3; ++c) points3[3 * i + c] = res.points.size() > (size_t)(3 * i + c) ? res.points[(size_t)(3 * i + c)] : 0.0;
    }
    return ok ? 1 : 0;
}

__attribute__((visibility("default"))) void lpslam_sym_eigen(const double* A, int n, double* eigval, double* eigvec) { LpSlam::sym_eigen_jacobi(A, n, eigval, eigvec); }
}

extern "C" __attribute__((visibility("default"))) int lpslam_sim3_solve_ransac(const double* p1c, const double* p2c, const double* obs1, const double* obs2,
                                                                                const double* is1, const double* is2, int n, const double* cam1, const double* cam2,
                                                                                int fix_scale, int iterations, uint32_t seed, double* s12, uint8_t* inlier)
{
    return LpSlam::sim3_solve_ransac(p1c, p2c, obs1, obs2, is1, is2, n, cam1, cam2, fix_scale != 0, iterations, seed, s12, inlier);
}

extern "C" __attribute__((visibility("default"))) int lpslam_epnp_solve(const double* pw, const double* uv, int n, const double* cam, double* R9, double* t3)
{
    return LpSlam::epnp_solve(pw, uv, n, cam, R9, t3) ? 1 : 0;
}

extern "C" __attribute__((visibility("default"))) int lpslam_pnp_solve_ransac(const double* pw, const double* obs, const double* inv_sigma2, int n, const double* cam,
                                                                             int iterations, uint32_t seed, double* pose7, uint8_t* inlier)
{
    return LpSlam::pnp_solve_ransac(pw, obs, inv_sigma2, n, cam, iterations, seed, pose7, inlier);
}
