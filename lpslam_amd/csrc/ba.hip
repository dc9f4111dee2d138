// ba.hip -- SE3 bundle adjustment on gfx950 (FP64): g2o-style Levenberg-Marquardt with landmark Schur complement.
//
// [UPSTREAM] g2o@691dc51 OptimizationAlgorithmLevenberg + BlockSolver<6,3> (buildSystem / Schur / solve) and the
// OpenVSLAM reprojection edges, which the reference runs on its mapping / global-optimisation threads
// (/root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:239,250-255; pin conan-packages/g2o-conan/conanfile.py:6).
//
// Device pipeline of one LM iteration (all sums are fixed-order segmented reductions: no float atomics, results are
// reproducible run to run):
//   point pass   one thread per landmark   : H_ll (3x3), b_l, W = B^T w A (6x3 per observation)
//   pose pass    one wavefront per keyframe: H_pp (6x6), b_p, robust chi2
//   per trial    point inverse (H_ll + lambda I)^-1, Y = W H_ll^-1;  Schur blocks S_ik = H_pp - sum Y W^T, one
//                wavefront per block pair over a precomputed pair list;  blocked Cholesky (32x32 panels, right-
//                looking, rhs carried as an extra row);  back substitution;  landmark update;  trial chi2.
// The reduced system [S | rhs | b_p | diag H_pp | chi2] is one contiguous buffer so a landmark-partitioned multi-GPU
// solve only needs one sum all-reduce of it per trial (lpslam_hip_ba_step_*).
#include "internal.h"
#include <cmath>
#include <cfloat>
#include <algorithm>

#pragma clang fp contract(off)

using namespace lpslam;

namespace {

constexpr int NB = 32;                 // Cholesky panel width

struct BaCam { double fx, fy, cx, cy, fxb, hub_mono, hub_stereo; };

struct BaView {                        // device pointers handed to kernels by value
    int n_poses, n_points, n_obs, n_free, dim, dim_pad;
    const double* poses; const double* points;      // state being evaluated
    const int* pose_slot; const int* free_pose;
    const int* o_pose; const int* o_point;
    const double* o_u; const double* o_v; const double* o_ur; const double* o_w;
    const uint8_t* o_active;
    const int* pt_start; const int* pt_obs; const int* ps_start; const int* ps_obs;
    double* W; double* Y; double* Ybl; double* Hll; double* bl; double* Hinv; double* Hpp;
    double* S; double* rhs; double* bp; double* hppdiag; double* chi_cur;   // reduced buffer sections
    double* xp; double* xl; double* chi_pose; double* part; double* scal;
    const int* blk_start; const int2* blk_terms;
    BaCam cam;
};

__device__ __forceinline__ void quat_to_rot(const double* q, double* R)
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// residual e = obs - projection, camera-frame point pc; returns 2 (mono) or 3 (stereo)
__device__ __forceinline__ int ba_residual(const BaView& v, int k, const double* R, const double* t, const double* X,
                                           double* e, double* pc)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2] + t[i];
    const double iz = 1.0 / pc[2];
    const double u = v.cam.fx * pc[0] * iz + v.cam.cx;
    const double vv = v.cam.fy * pc[1] * iz + v.cam.cy;
    e[0] = v.o_u[k] - u; e[1] = v.o_v[k] - vv;
    const double ur = v.o_ur[k];
    if (ur < 0) { e[2] = 0; return 2; }
    e[2] = ur - (u - v.cam.fxb * iz);
    return 3;
}

__device__ __forceinline__ void huber(double e2, double delta, double* rho0, double* rho1)
{
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; }
    else { const double sq = sqrt(e2); *rho0 = 2 * sq * delta - dsqr; *rho1 = delta / sq; }
}

// Jacobians of the reprojection error: A (D x 3, landmark), B (D x 6, pose, rotation first)
__device__ __forceinline__ void ba_jacobians(const BaCam& c, const double* R, const double* pc, int D, double A[3][3], double B[3][6])
{
    const double x = pc[0], y = pc[1], z = pc[2], z2 = z * z;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        A[0][k] = -c.fx * R[k] / z + c.fx * x * R[6 + k] / z2;
        A[1][k] = -c.fy * R[3 + k] / z + c.fy * y * R[6 + k] / z2;
        A[2][k] = A[0][k] - c.fxb * R[6 + k] / z2;
    }
    B[0][0] = x * y / z2 * c.fx;          B[0][1] = -(1.0 + (x * x / z2)) * c.fx; B[0][2] = y / z * c.fx;
    B[0][3] = -1.0 / z * c.fx;            B[0][4] = 0.0;                           B[0][5] = x / z2 * c.fx;
    B[1][0] = (1.0 + y * y / z2) * c.fy;  B[1][1] = -x * y / z2 * c.fy;            B[1][2] = -x / z * c.fy;
    B[1][3] = 0.0;                        B[1][4] = -1.0 / z * c.fy;               B[1][5] = y / z2 * c.fy;
    B[2][0] = B[0][0] - c.fxb * y / z2;   B[2][1] = B[0][1] + c.fxb * x / z2;      B[2][2] = B[0][2];
    B[2][3] = B[0][3];                    B[2][4] = 0.0;                           B[2][5] = B[0][5] - c.fxb / z2;
    if (D == 2) {      // monocular edge: the third row does not exist; zero rows keep every sum exact and loops unrolled
#pragma unroll
        for (int k = 0; k < 3; ++k) A[2][k] = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) B[2][k] = 0.0;
    }
}

// weight (rho1 * inv_sigma2) and robustified chi2 of one observation
__device__ __forceinline__ double ba_weight(const BaView& v, int k, int D, const double* e, int robust, double* rho0)
{
    const double om = v.o_w[k];
    const double chi = om * (e[0] * e[0] + e[1] * e[1] + (D == 3 ? e[2] * e[2] : 0.0));
    const double delta = D == 3 ? v.cam.hub_stereo : v.cam.hub_mono;
    double w = om;
    *rho0 = chi;
    if (robust && delta > 0) { double r1; huber(chi, delta, rho0, &r1); w *= r1; }
    return w;
}

// ---- point pass: H_ll, b_l, W per observation ------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_ba_point_pass(BaView v, int robust, int points_fixed)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.n_points) return;
    if (points_fixed) {        // motion-only: landmarks are constants, no landmark blocks (x_l = 0)
        for (int s = v.pt_start[j]; s < v.pt_start[j + 1]; ++s) { double* Wk = v.W + 18 * (size_t)v.pt_obs[s]; for (int i = 0; i < 18; ++i) Wk[i] = 0.0; }
        for (int i = 0; i < 6; ++i) v.Hll[6 * (size_t)j + i] = 0.0;
        for (int i = 0; i < 3; ++i) v.bl[3 * (size_t)j + i] = 0.0;
        return;
    }
    const double X[3] = {v.points[3 * j], v.points[3 * j + 1], v.points[3 * j + 2]};
    double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int s = v.pt_start[j]; s < v.pt_start[j + 1]; ++s) {
        const int k = v.pt_obs[s];
        double* Wk = v.W + 18 * (size_t)k;
        const int p = v.o_pose[k];
        const int slot = v.pose_slot[p];
        if (!v.o_active[k]) {
#pragma unroll
            for (int i = 0; i < 18; ++i) Wk[i] = 0.0;
            continue;
        }
        double R[9], e[3], pc[3], A[3][3], B[3][6], rho0;
        quat_to_rot(v.poses + 7 * p, R);
        const int D = ba_residual(v, k, R, v.poses + 7 * p + 4, X, e, pc);
        ba_jacobians(v.cam, R, pc, D, A, B);
        const double w = ba_weight(v, k, D, e, robust, &rho0);
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int c = a; c < 3; ++c) {
                double s2 = 0;
                _Pragma("unroll") for (int r = 0; r < 3; ++r) s2 += A[r][a] * w * A[r][c];
                h[idx++] += s2;
            }
            double s3 = 0;
            _Pragma("unroll") for (int r = 0; r < 3; ++r) s3 += A[r][a] * (-w * e[r]);
            b[a] += s3;
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double s2 = 0;
                if (slot >= 0) _Pragma("unroll") for (int r = 0; r < 3; ++r) s2 += B[r][a] * w * A[r][c];
                Wk[a * 3 + c] = s2;
            }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) v.Hll[6 * (size_t)j + i] = h[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) v.bl[3 * (size_t)j + i] = b[i];
}

__device__ __forceinline__ double wave_sum(double x)
{
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// ---- pose pass: H_pp, b_p, chi2 (one wavefront per keyframe; fixed keyframes only contribute chi2) ---------------
__global__ __launch_bounds__(256) void k_ba_pose_pass(BaView v, int robust, int chi_only, double* chi_out)
{
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= v.n_poses) return;
    double R[9];
    quat_to_rot(v.poses + 7 * p, R);
    const double* t = v.poses + 7 * p + 4;
    const int slot = v.pose_slot[p];
    double h[21], b[6], chi = 0;
#pragma unroll
    for (int i = 0; i < 21; ++i) h[i] = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = 0;
    for (int s = v.ps_start[p] + lane; s < v.ps_start[p + 1]; s += 64) {
        const int k = v.ps_obs[s];
        if (!v.o_active[k]) continue;
        const int j = v.o_point[k];
        const double X[3] = {v.points[3 * j], v.points[3 * j + 1], v.points[3 * j + 2]};
        double e[3], pc[3], rho0;
        const int D = ba_residual(v, k, R, t, X, e, pc);
        const double w = ba_weight(v, k, D, e, robust, &rho0);
        chi += rho0;
        if (chi_only || slot < 0) continue;
        double A[3][3], B[3][6];
        ba_jacobians(v.cam, R, pc, D, A, B);
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int c = a; c < 6; ++c) {
                double s2 = 0;
                _Pragma("unroll") for (int r = 0; r < 3; ++r) s2 += B[r][a] * w * B[r][c];
                h[idx++] += s2;
            }
            double s3 = 0;
            _Pragma("unroll") for (int r = 0; r < 3; ++r) s3 += B[r][a] * (-w * e[r]);
            b[a] += s3;
        }
    }
    chi = wave_sum(chi);
    if (lane == 0) chi_out[p] = chi;
    if (chi_only || slot < 0) return;
#pragma unroll
    for (int i = 0; i < 21; ++i) h[i] = wave_sum(h[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = wave_sum(b[i]);
    if (lane == 0) {
        double* H = v.Hpp + 36 * (size_t)slot;
        int idx = 0;
        for (int a = 0; a < 6; ++a)
            for (int c = a; c < 6; ++c) { H[a * 6 + c] = h[idx]; H[c * 6 + a] = h[idx]; ++idx; }
        for (int a = 0; a < 6; ++a) { v.bp[6 * slot + a] = b[a]; v.hppdiag[6 * slot + a] = H[a * 7]; }
    }
}

// ---- small deterministic reductions (single workgroup of one wavefront) -------------------------------------------
// mode 0: out[0] = sum(in[0..n));  mode 1: out[0] = max |in|
__global__ __launch_bounds__(64) void k_ba_reduce(const double* in, int n, int stride, double* out, int mode)
{
    const int lane = threadIdx.x;
    double acc = 0;
    for (int i = lane; i < n; i += 64) { const double x = in[(size_t)i * stride]; acc = mode ? fmax(acc, fabs(x)) : acc + x; }
    for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(acc, o); acc = mode ? fmax(acc, y) : acc + y; }
    if (lane == 0) out[0] = acc;
}

// max |diag H_ll| over landmarks: block partial maxima (max is order independent)
__global__ __launch_bounds__(256) void k_ba_maxdiag_ll(BaView v, double* part)
{
    __shared__ double sm[4];
    const int j = blockIdx.x * 256 + threadIdx.x;
    double m = 0;
    if (j < v.n_points) { const double* h = v.Hll + 6 * (size_t)j; m = fmax(fabs(h[0]), fmax(fabs(h[3]), fabs(h[5]))); }
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}

// ---- per trial: (H_ll + lambda I)^-1, Y = W H^-1, Y b_l -----------------------------------------------------------
__global__ __launch_bounds__(128) void k_ba_point_inv(BaView v, double lambda)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.n_points) return;
    const double* hl = v.Hll + 6 * (size_t)j;
    const double a = hl[0] + lambda, b = hl[1], c = hl[2], d = hl[3] + lambda, e = hl[4], f = hl[5] + lambda;
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    double Hi[9];
    if (fabs(det) > 0) {
        const double id = 1.0 / det;
        Hi[0] = c00 * id; Hi[1] = c01 * id; Hi[2] = c02 * id;
        Hi[3] = Hi[1]; Hi[4] = (a * f - c * c) * id; Hi[5] = (b * c - a * e) * id;
        Hi[6] = Hi[2]; Hi[7] = Hi[5]; Hi[8] = (a * d - b * b) * id;
    } else {
        for (int i = 0; i < 9; ++i) Hi[i] = 0;
    }
    double* ho = v.Hinv + 6 * (size_t)j;
    ho[0] = Hi[0]; ho[1] = Hi[1]; ho[2] = Hi[2]; ho[3] = Hi[4]; ho[4] = Hi[5]; ho[5] = Hi[8];
    const double b0 = v.bl[3 * (size_t)j], b1 = v.bl[3 * (size_t)j + 1], b2 = v.bl[3 * (size_t)j + 2];
    for (int s = v.pt_start[j]; s < v.pt_start[j + 1]; ++s) {
        const int k = v.pt_obs[s];
        const double* Wk = v.W + 18 * (size_t)k;
        double* Yk = v.Y + 18 * (size_t)k;
        double* yb = v.Ybl + 6 * (size_t)k;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const double w0 = Wk[r * 3], w1 = Wk[r * 3 + 1], w2 = Wk[r * 3 + 2];
            const double y0 = w0 * Hi[0] + w1 * Hi[3] + w2 * Hi[6];
            const double y1 = w0 * Hi[1] + w1 * Hi[4] + w2 * Hi[7];
            const double y2 = w0 * Hi[2] + w1 * Hi[5] + w2 * Hi[8];
            Yk[r * 3] = y0; Yk[r * 3 + 1] = y1; Yk[r * 3 + 2] = y2;
            yb[r] = y0 * b0 + y1 * b1 + y2 * b2;
        }
    }
}

// ---- Schur complement: one wavefront per block pair (i <= k) -------------------------------------------------------
// S_ik = [i == k] H_pp,i - sum_terms Y_a W_b^T ; rhs_i = b_p,i - sum_{obs of i} Y b_l (diagonal blocks)
__global__ __launch_bounds__(64) void k_ba_schur(BaView v)
{
    const int lane = threadIdx.x;
    // block pair index -> (i, k), i <= k, row-major upper triangle
    int pidx = blockIdx.x;
    int i = 0;
    {
        int rowlen = v.n_free;
        while (pidx >= rowlen) { pidx -= rowlen; --rowlen; ++i; }
    }
    const int k = i + pidx;
    double acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0;
    for (int t = v.blk_start[blockIdx.x] + lane; t < v.blk_start[blockIdx.x + 1]; t += 64) {
        const int2 ab = v.blk_terms[t];
        const double* Ya = v.Y + 18 * (size_t)ab.x;
        const double* Wb = v.W + 18 * (size_t)ab.y;
        double y[18], w[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) { y[q] = Ya[q]; w[q] = Wb[q]; }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                acc[r * 6 + c] += y[r * 3] * w[c * 3] + y[r * 3 + 1] * w[c * 3 + 1] + y[r * 3 + 2] * w[c * 3 + 2];
    }
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = wave_sum(acc[q]);
    const int n = v.dim_pad;
    if (i == k) {
        const int p = v.free_pose[i];
        double r6[6] = {0, 0, 0, 0, 0, 0};
        for (int s = v.ps_start[p] + lane; s < v.ps_start[p + 1]; s += 64) {
            const double* yb = v.Ybl + 6 * (size_t)v.ps_obs[s];
#pragma unroll
            for (int q = 0; q < 6; ++q) r6[q] += yb[q];
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) r6[q] = wave_sum(r6[q]);
        if (lane == 0) {
            for (int q = 0; q < 6; ++q) v.rhs[6 * i + q] = v.bp[6 * i + q] - r6[q];
            const double* H = v.Hpp + 36 * (size_t)i;
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) v.S[(size_t)(6 * i + r) * n + 6 * i + c] = H[r * 6 + c] - acc[r * 6 + c];
        }
    } else if (lane == 0) {
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) {
                const double val = -acc[r * 6 + c];
                v.S[(size_t)(6 * i + r) * n + 6 * k + c] = val;
                v.S[(size_t)(6 * k + c) * n + 6 * i + r] = val;
            }
    }
}

// ---- blocked Cholesky of (S + lambda I), rhs carried as row `dim` so that L[dim][0..dim) = L^-1 rhs ------------------
__global__ __launch_bounds__(256) void k_chol_prep(double* S, const double* rhs, int dim, int n, double lambda, double* scal)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) scal[5] = 0.0;                      // failure flag
    if (i < dim) { S[(size_t)i * n + i] += lambda; S[(size_t)dim * n + i] = rhs[i]; }
    else if (i < n) {
        for (int c = 0; c < i; ++c) if (i != dim || c >= dim) S[(size_t)i * n + c] = 0.0;
        S[(size_t)i * n + i] = (i == dim) ? 1e200 : 1.0;
    }
}

// panel step kb: every workgroup (one wavefront) factors the diagonal block in LDS; workgroup b > 0 then solves
// row block kb + b against it.
__global__ __launch_bounds__(64) void k_chol_panel(double* S, int n, int kb, double* scal)
{
    __shared__ double Lk[NB][NB + 1];
    __shared__ double Ai[NB][NB + 1];
    const int lane = threadIdx.x;
    const int ib = kb + blockIdx.x;
    const size_t d0 = (size_t)kb * NB;
    for (int t = lane; t < NB * NB; t += 64) { const int r = t / NB, c = t % NB; Lk[r][c] = S[(d0 + r) * n + d0 + c]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    bool fail = false;
    for (int j = 0; j < NB; ++j) {
        double d = Lk[j][j];
        if (!(d > 0.0)) { fail = true; d = 1.0; }
        d = sqrt(d);
        const double lij = lane > j && lane < NB ? Lk[lane][j] / d : 0.0;
        __builtin_amdgcn_wave_barrier();
        if (lane == j) Lk[j][j] = d;
        if (lane > j && lane < NB) Lk[lane][j] = lij;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // trailing update of the block: rows lane (and lane+32 half handles the upper column range)
        const int r = lane & 31, half = lane >> 5;
        if (r > j) {
            const double lr = Lk[r][j];
            for (int c = j + 1 + half; c <= r; c += 2) Lk[r][c] -= lr * Lk[c][j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (blockIdx.x == 0) {
        if (fail && lane == 0) scal[5] = 1.0;
        for (int t = lane; t < NB * NB; t += 64) { const int r = t / NB, c = t % NB; if (c <= r) S[(d0 + r) * n + d0 + c] = Lk[r][c]; }
        return;
    }
    const size_t r0 = (size_t)ib * NB;
    for (int t = lane; t < NB * NB; t += 64) { const int r = t / NB, c = t % NB; Ai[r][c] = S[(r0 + r) * n + d0 + c]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < NB) {
        // X L^T = A  ->  x[c] = (a[c] - sum_{m<c} x[m] L[c][m]) / L[c][c]
        for (int c = 0; c < NB; ++c) {
            double s = Ai[lane][c];
            for (int m = 0; m < c; ++m) s -= Ai[lane][m] * Lk[c][m];
            Ai[lane][c] = s / Lk[c][c];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int t = lane; t < NB * NB; t += 64) { const int r = t / NB, c = t % NB; S[(r0 + r) * n + d0 + c] = Ai[r][c]; }
}

// trailing update after panel kb: A_ij -= L_ik L_jk^T for block pairs i >= j > kb (lower triangle)
__global__ __launch_bounds__(256) void k_chol_update(double* S, int n, int kb)
{
    __shared__ double Li[NB][NB + 1];
    __shared__ double Lj[NB][NB + 1];
    // pair index -> (i, j) in the lower triangle of the trailing (nb - kb - 1) blocks
    int pidx = blockIdx.x, bi = 0;
    while (pidx > bi) { pidx -= bi + 1; ++bi; }
    const int i = kb + 1 + bi, j = kb + 1 + pidx;
    const size_t ri = (size_t)i * NB, rj = (size_t)j * NB, ck = (size_t)kb * NB;
    for (int t = threadIdx.x; t < NB * NB; t += 256) {
        const int r = t / NB, c = t % NB;
        Li[r][c] = S[(ri + r) * n + ck + c];
        Lj[r][c] = S[(rj + r) * n + ck + c];
    }
    __syncthreads();
    const int r = threadIdx.x / 8, c0 = (threadIdx.x % 8) * 4;
    double acc[4] = {0, 0, 0, 0};
    for (int m = 0; m < NB; ++m) {
        const double a = Li[r][m];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += a * Lj[c0 + q][m];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) S[(ri + r) * n + rj + c0 + q] -= acc[q];
}

// backward substitution L^T x = y with y = L[dim][0..dim); single workgroup, x in LDS
__global__ __launch_bounds__(256) void k_chol_backsolve(const double* S, int n, int dim, double* xp)
{
    extern __shared__ double xs[];                 // [n]
    __shared__ double part[8][NB];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) xs[i] = i < dim ? S[(size_t)dim * n + i] : 0.0;
    __syncthreads();
    const int nb = (dim + NB - 1) / NB;
    for (int kb = nb - 1; kb >= 0; --kb) {
        const int c0 = kb * NB;
        // s_c = sum_{i >= c0 + NB, i < dim} L[i][c] x[i] : 32 columns x 8 row groups
        const int c = tid & 31, g = tid >> 5;
        double s = 0;
        for (int i = c0 + NB + g; i < dim; i += 8) s += S[(size_t)i * n + c0 + c] * xs[i];
        part[g][c] = s;
        __syncthreads();
        if (tid < 64) {
            // diagonal block back-solve by one wavefront, columns from the last to the first
            double y = 0;
            if (tid < NB) { y = xs[c0 + tid]; for (int q = 0; q < 8; ++q) y -= part[q][tid]; }
            for (int cc = NB - 1; cc >= 0; --cc) {
                const int gi = c0 + cc;
                double xv = 0;
                if (gi < dim) xv = __shfl(y, cc) / S[(size_t)gi * n + gi];
                else xv = 0;
                if (tid == cc) y = xv;
                if (tid < cc && gi < dim) y -= S[(size_t)gi * n + c0 + tid] * xv;
            }
            if (tid < NB) xs[c0 + tid] = y;
        }
        __syncthreads();
    }
    for (int i = tid; i < dim; i += 256) xp[i] = xs[i];
}

// ---- landmark back substitution, state update, scale terms ----------------------------------------------------------
__global__ __launch_bounds__(128) void k_ba_backsub(BaView v, double lambda, double* points_out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double sc = 0;
    if (j < v.n_points) {
        double r[3] = {v.bl[3 * (size_t)j], v.bl[3 * (size_t)j + 1], v.bl[3 * (size_t)j + 2]};
        for (int s = v.pt_start[j]; s < v.pt_start[j + 1]; ++s) {
            const int k = v.pt_obs[s];
            const int slot = v.pose_slot[v.o_pose[k]];
            if (slot < 0) continue;
            const double* Wk = v.W + 18 * (size_t)k;
            const double* x = v.xp + 6 * slot;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) r[c] -= Wk[rr * 3 + c] * x[rr];
        }
        const double* h = v.Hinv + 6 * (size_t)j;
        const double x0 = h[0] * r[0] + h[1] * r[1] + h[2] * r[2];
        const double x1 = h[1] * r[0] + h[3] * r[1] + h[4] * r[2];
        const double x2 = h[2] * r[0] + h[4] * r[1] + h[5] * r[2];
        points_out[3 * (size_t)j] = v.points[3 * (size_t)j] + x0;
        points_out[3 * (size_t)j + 1] = v.points[3 * (size_t)j + 1] + x1;
        points_out[3 * (size_t)j + 2] = v.points[3 * (size_t)j + 2] + x2;
        sc = x0 * (lambda * x0 + v.bl[3 * (size_t)j]) + x1 * (lambda * x1 + v.bl[3 * (size_t)j + 1]) + x2 * (lambda * x2 + v.bl[3 * (size_t)j + 2]);
    }
    // per-block partial of the landmark part of computeScale (fixed order inside the block)
    __shared__ double sm[2];
    sc = wave_sum(sc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = sc;
    __syncthreads();
    if (threadIdx.x == 0) v.part[blockIdx.x] = sm[0] + sm[1];
}

__device__ __forceinline__ void pose_oplus(const double* pose, const double* d, double* out)
{
    const double wx = d[0], wy = d[1], wz = d[2];
    const double theta2 = wx * wx + wy * wy + wz * wz;
    const double theta = sqrt(theta2);
    double a, b, c, qe[4];
    if (theta < 0.00001) {
        a = 1.0; b = 0.5; c = 1.0 / 6.0;
        qe[0] = 1.0; qe[1] = 0.5 * wx; qe[2] = 0.5 * wy; qe[3] = 0.5 * wz;
    } else {
        a = sin(theta) / theta;
        b = (1 - cos(theta)) / theta2;
        c = (theta - sin(theta)) / (theta2 * theta);
        const double sh = sin(0.5 * theta) / theta;
        qe[0] = cos(0.5 * theta); qe[1] = sh * wx; qe[2] = sh * wy; qe[3] = sh * wz;
    }
    const double Wm[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += Wm[i * 3 + k] * Wm[k * 3 + j]; W2[i * 3 + j] = s; }
    double Re[9], V[9];
    for (int i = 0; i < 9; ++i) { const double I = (i % 4 == 0) ? 1.0 : 0.0; Re[i] = I + a * Wm[i] + b * W2[i]; V[i] = I + b * Wm[i] + c * W2[i]; }
    const double* t = pose + 4;
    double tn[3];
    for (int i = 0; i < 3; ++i)
        tn[i] = V[i * 3] * d[3] + V[i * 3 + 1] * d[4] + V[i * 3 + 2] * d[5] + Re[i * 3] * t[0] + Re[i * 3 + 1] * t[1] + Re[i * 3 + 2] * t[2];
    const double* q = pose;
    double qn[4];
    qn[0] = qe[0] * q[0] - qe[1] * q[1] - qe[2] * q[2] - qe[3] * q[3];
    qn[1] = qe[0] * q[1] + qe[1] * q[0] + qe[2] * q[3] - qe[3] * q[2];
    qn[2] = qe[0] * q[2] - qe[1] * q[3] + qe[2] * q[0] + qe[3] * q[1];
    qn[3] = qe[0] * q[3] + qe[1] * q[2] - qe[2] * q[1] + qe[3] * q[0];
    const double nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int i = 0; i < 4; ++i) out[i] = qn[i] / nn;
    for (int i = 0; i < 3; ++i) out[4 + i] = tn[i];
}

// poses_out = exp(x_p) * poses; scal[3] = pose part of computeScale (single workgroup, fixed order)
__global__ __launch_bounds__(64) void k_ba_pose_update(BaView v, double lambda, double* poses_out)
{
    const int lane = threadIdx.x;
    double sc = 0;
    for (int p = lane; p < v.n_poses; p += 64) {
        const int slot = v.pose_slot[p];
        if (slot < 0) { for (int i = 0; i < 7; ++i) poses_out[7 * p + i] = v.poses[7 * p + i]; continue; }
        pose_oplus(v.poses + 7 * p, v.xp + 6 * slot, poses_out + 7 * p);
        for (int a = 0; a < 6; ++a) { const double x = v.xp[6 * slot + a]; sc += x * (lambda * x + v.bp[6 * slot + a]); }
    }
    sc = wave_sum(sc);
    if (lane == 0) v.scal[3] = sc;
}

// per-observation chi2 (non robust) and depth sign
__global__ __launch_bounds__(256) void k_ba_obs_chi2(BaView v, double* chi2, uint8_t* depth_pos)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= v.n_obs) return;
    const int p = v.o_pose[k], j = v.o_point[k];
    double R[9], e[3], pc[3];
    quat_to_rot(v.poses + 7 * p, R);
    const double X[3] = {v.points[3 * j], v.points[3 * j + 1], v.points[3 * j + 2]};
    const int D = ba_residual(v, k, R, v.poses + 7 * p + 4, X, e, pc);
    chi2[k] = v.o_w[k] * (e[0] * e[0] + e[1] * e[1] + (D == 3 ? e[2] * e[2] : 0.0));
    depth_pos[k] = pc[2] > 0 ? 1 : 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct lpslam_hip_ba {
    lpslam_hip_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    int n_poses = 0, n_points = 0, n_obs = 0, n_free = 0, dim = 0, dim_pad = 0, n_blocks = 0;
    int cur = 0;                                  // index of the accepted state
    double *d_poses[2] = {nullptr, nullptr}, *d_points[2] = {nullptr, nullptr};
    double *d_poses0 = nullptr, *d_points0 = nullptr;      // state given at creation (lpslam_hip_ba_reset)
    int *d_pose_slot = nullptr, *d_free_pose = nullptr, *d_o_pose = nullptr, *d_o_point = nullptr;
    double *d_o_u = nullptr, *d_o_v = nullptr, *d_o_ur = nullptr, *d_o_w = nullptr;
    uint8_t* d_o_active = nullptr;
    int *d_pt_start = nullptr, *d_pt_obs = nullptr, *d_ps_start = nullptr, *d_ps_obs = nullptr;
    double *d_W = nullptr, *d_Y = nullptr, *d_Ybl = nullptr, *d_Hll = nullptr, *d_bl = nullptr, *d_Hinv = nullptr, *d_Hpp = nullptr;
    double* d_red = nullptr; int64_t red_n = 0; bool red_external = false;
    double *d_xp = nullptr, *d_xl = nullptr, *d_chi_pose = nullptr, *d_part = nullptr, *d_scal = nullptr;
    double* d_chi_obs = nullptr; uint8_t* d_depth = nullptr;
    int* d_blk_start = nullptr; int2* d_blk_terms = nullptr;
    int part_n = 0;
    BaCam cam{};
    std::vector<double> h_ur;                      // mono/stereo classification for the outlier thresholds
    // LM state (g2o OptimizationAlgorithmLevenberg)
    double lambda = 0, ni = 2, current_chi = 0, rho = 0;
    int qmax = 0; int robust = 1; double chi_before = 0;
    int points_fixed = 0;
    std::vector<void*> allocs;
};

namespace {

template <class T>
int dalloc(lpslam_hip_ba* b, T** p, size_t n)
{
    LP_HIP(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
    b->allocs.push_back(*p);
    return LPSLAM_HIP_OK;
}
template <class T>
int upload(lpslam_hip_ba* b, T** p, const std::vector<T>& h)
{
    int rc = dalloc(b, p, h.size()); if (rc) return rc;
    if (!h.empty()) LP_HIP(hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return LPSLAM_HIP_OK;
}

BaView make_view(lpslam_hip_ba* b, int state)
{
    BaView v{};
    v.n_poses = b->n_poses; v.n_points = b->n_points; v.n_obs = b->n_obs; v.n_free = b->n_free; v.dim = b->dim; v.dim_pad = b->dim_pad;
    v.poses = b->d_poses[state]; v.points = b->d_points[state];
    v.pose_slot = b->d_pose_slot; v.free_pose = b->d_free_pose; v.o_pose = b->d_o_pose; v.o_point = b->d_o_point;
    v.o_u = b->d_o_u; v.o_v = b->d_o_v; v.o_ur = b->d_o_ur; v.o_w = b->d_o_w; v.o_active = b->d_o_active;
    v.pt_start = b->d_pt_start; v.pt_obs = b->d_pt_obs; v.ps_start = b->d_ps_start; v.ps_obs = b->d_ps_obs;
    v.W = b->d_W; v.Y = b->d_Y; v.Ybl = b->d_Ybl; v.Hll = b->d_Hll; v.bl = b->d_bl; v.Hinv = b->d_Hinv; v.Hpp = b->d_Hpp;
    const size_t n = (size_t)b->dim_pad;
    v.S = b->d_red; v.rhs = b->d_red + n * n; v.bp = v.rhs + n; v.hppdiag = v.bp + n; v.chi_cur = v.hppdiag + n;
    v.xp = b->d_xp; v.xl = b->d_xl; v.chi_pose = b->d_chi_pose; v.part = b->d_part; v.scal = b->d_scal;
    v.blk_start = b->d_blk_start; v.blk_terms = b->d_blk_terms;
    v.cam = b->cam;
    return v;
}

// linearise at the accepted state: blocks, b, chi2 (-> chi_cur), max diagonal (-> scal[4] = landmark part)
int ba_linearize(lpslam_hip_ba* b)
{
    BaView v = make_view(b, b->cur);
    hipStream_t s = b->stream;
    if (b->n_points) hipLaunchKernelGGL(k_ba_point_pass, dim3((b->n_points + 127) / 128), dim3(128), 0, s, v, b->robust, b->points_fixed);
    hipLaunchKernelGGL(k_ba_pose_pass, dim3((b->n_poses + 3) / 4), dim3(256), 0, s, v, b->robust, 0, b->d_chi_pose);
    hipLaunchKernelGGL(k_ba_reduce, dim3(1), dim3(64), 0, s, b->d_chi_pose, b->n_poses, 1, v.chi_cur, 0);
    if (b->n_points) {
        const int nb = (b->n_points + 255) / 256;
        hipLaunchKernelGGL(k_ba_maxdiag_ll, dim3(nb), dim3(256), 0, s, v, b->d_part);
        hipLaunchKernelGGL(k_ba_reduce, dim3(1), dim3(64), 0, s, b->d_part, nb, 1, b->d_scal + 4, 1);
    }
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

// Schur complement for the current lambda into the reduced buffer (pose diagonal WITHOUT lambda)
int ba_reduce_system(lpslam_hip_ba* b)
{
    BaView v = make_view(b, b->cur);
    hipStream_t s = b->stream;
    if (b->n_points) hipLaunchKernelGGL(k_ba_point_inv, dim3((b->n_points + 127) / 128), dim3(128), 0, s, v, b->lambda);
    if (b->n_blocks) hipLaunchKernelGGL(k_ba_schur, dim3(b->n_blocks), dim3(64), 0, s, v);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

// factor + solve the (all-reduced) system, update into the trial state, trial chi2 -> scal[1], scale parts -> scal[2], scal[3]
int ba_solve_update(lpslam_hip_ba* b)
{
    BaView v = make_view(b, b->cur);
    hipStream_t s = b->stream;
    const int n = b->dim_pad, nb = n / NB;
    const int trial = b->cur ^ 1;
    if (b->dim > 0) {
        hipLaunchKernelGGL(k_chol_prep, dim3((n + 255) / 256), dim3(256), 0, s, v.S, v.rhs, b->dim, n, b->lambda, b->d_scal);
        for (int kb = 0; kb < nb; ++kb) {
            hipLaunchKernelGGL(k_chol_panel, dim3(nb - kb), dim3(64), 0, s, v.S, n, kb, b->d_scal);
            const int t = nb - kb - 1;
            if (t > 0) hipLaunchKernelGGL(k_chol_update, dim3(t * (t + 1) / 2), dim3(256), 0, s, v.S, n, kb);
        }
        hipLaunchKernelGGL(k_chol_backsolve, dim3(1), dim3(256), n * sizeof(double), s, v.S, n, b->dim, b->d_xp);
    }
    const int pb = (b->n_points + 127) / 128;
    if (b->n_points) {
        hipLaunchKernelGGL(k_ba_backsub, dim3(pb), dim3(128), 0, s, v, b->lambda, b->d_points[trial]);
        hipLaunchKernelGGL(k_ba_reduce, dim3(1), dim3(64), 0, s, b->d_part, pb, 1, b->d_scal + 2, 0);
    }
    hipLaunchKernelGGL(k_ba_pose_update, dim3(1), dim3(64), 0, s, v, b->lambda, b->d_poses[trial]);
    BaView vt = make_view(b, trial);
    hipLaunchKernelGGL(k_ba_pose_pass, dim3((b->n_poses + 3) / 4), dim3(256), 0, s, vt, b->robust, 1, b->d_chi_pose);
    hipLaunchKernelGGL(k_ba_reduce, dim3(1), dim3(64), 0, s, b->d_chi_pose, b->n_poses, 1, b->d_scal + 1, 0);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

int read_scal(lpslam_hip_ba* b, double* h8)
{
    LP_HIP(hipMemcpyAsync(h8, b->d_scal, 8 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    LP_HIP(hipStreamSynchronize(b->stream));
    return LPSLAM_HIP_OK;
}

// g2o's lambda control for one finished trial; returns true when the outer iteration is over
bool lm_decide(lpslam_hip_ba* b, double temp_chi, double scale, bool ok2, int* accepted)
{
    if (!ok2) temp_chi = DBL_MAX;
    double rho = b->current_chi - temp_chi;
    scale += 1e-3;
    rho /= scale;
    if (rho > 0 && std::isfinite(temp_chi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        const double sf = std::max(1. / 3., alpha);
        b->lambda *= sf;
        b->ni = 2;
        b->current_chi = temp_chi;
        b->cur ^= 1;                                 // discardTop: the trial state becomes the accepted one
        *accepted = 1;
    } else {
        b->lambda *= b->ni;
        b->ni *= 2;
        *accepted = 0;                               // pop: keep the accepted state
    }
    b->rho = rho;
    b->qmax++;
    return !(rho < 0 && b->qmax < 10);
}

}  // namespace

extern "C" {

int lpslam_hip_ba_create(lpslam_hip_ctx* ctx, const double* poses, const uint8_t* fixed, int32_t n_poses, const double* points,
                         int32_t n_points, const lpslam_hip_ba_obs* obs, int32_t n_obs, const lpslam_hip_ba_camera* cam,
                         lpslam_hip_ba** out)
{
    if (!ctx || !poses || !points || !obs || !cam || !out || n_poses < 1 || n_points < 0 || n_obs < 0) {
        set_error("invalid bundle-adjustment arguments"); return LPSLAM_HIP_ERR_INVALID;
    }
    *out = nullptr;
    for (int k = 0; k < n_obs; ++k)
        if (obs[k].pose < 0 || obs[k].pose >= n_poses || obs[k].point < 0 || obs[k].point >= n_points) {
            set_error("observation %d references pose %d / point %d out of range", k, obs[k].pose, obs[k].point);
            return LPSLAM_HIP_ERR_INVALID;
        }
    LP_HIP(hipSetDevice(ctx->cfg.device));
    lpslam_hip_ba* b = new lpslam_hip_ba();
    b->ctx = ctx; b->stream = ctx->stream;
    b->n_poses = n_poses; b->n_points = n_points; b->n_obs = n_obs;
    b->cam = BaCam{cam->fx, cam->fy, cam->cx, cam->cy, cam->focal_x_baseline, cam->huber_mono, cam->huber_stereo};
    std::vector<int> slot(n_poses), free_pose;
    for (int i = 0; i < n_poses; ++i) { if (fixed && fixed[i]) slot[i] = -1; else { slot[i] = (int)free_pose.size(); free_pose.push_back(i); } }
    b->n_free = (int)free_pose.size();
    b->dim = 6 * b->n_free;
    b->dim_pad = ((b->dim + 1 + NB - 1) / NB) * NB;           // room for the rhs row
    b->n_blocks = b->n_free * (b->n_free + 1) / 2;
    // structure: CSR by landmark and by keyframe (observation order inside a segment = input order)
    std::vector<int> o_pose(n_obs), o_point(n_obs);
    std::vector<double> ou(n_obs), ov(n_obs), our(n_obs), ow(n_obs);
    std::vector<int> pt_start(n_points + 1, 0), ps_start(n_poses + 1, 0), pt_obs(n_obs), ps_obs(n_obs);
    for (int k = 0; k < n_obs; ++k) {
        o_pose[k] = obs[k].pose; o_point[k] = obs[k].point; ou[k] = obs[k].u; ov[k] = obs[k].v; our[k] = obs[k].ur; ow[k] = obs[k].inv_sigma2;
        pt_start[obs[k].point + 1]++; ps_start[obs[k].pose + 1]++;
    }
    b->h_ur = our;
    for (int j = 0; j < n_points; ++j) pt_start[j + 1] += pt_start[j];
    for (int i = 0; i < n_poses; ++i) ps_start[i + 1] += ps_start[i];
    {
        std::vector<int> fp(pt_start.begin(), pt_start.end() - 1), fs(ps_start.begin(), ps_start.end() - 1);
        for (int k = 0; k < n_obs; ++k) { pt_obs[fp[o_point[k]]++] = k; ps_obs[fs[o_pose[k]]++] = k; }
    }
    // pair lists per block pair (slot_a <= slot_b), terms in landmark order
    std::vector<int> blk_count((size_t)b->n_blocks + 1, 0);
    auto blk_index = [&](int i, int k) { return i * b->n_free - i * (i - 1) / 2 + (k - i); };
    std::vector<std::pair<int, int>> tmp;      // (slot, obs) of one landmark
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int> fill;
        std::vector<int2> terms;
        if (pass == 1) {
            for (int q = 0; q < b->n_blocks; ++q) blk_count[q + 1] += blk_count[q];
            fill.assign(blk_count.begin(), blk_count.end() - 1);
            terms.resize((size_t)blk_count[b->n_blocks]);
        }
        for (int j = 0; j < n_points; ++j) {
            tmp.clear();
            for (int s = pt_start[j]; s < pt_start[j + 1]; ++s) { const int k = pt_obs[s]; if (slot[o_pose[k]] >= 0) tmp.emplace_back(slot[o_pose[k]], k); }
            std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; });
            for (size_t a = 0; a < tmp.size(); ++a)
                for (size_t c = a; c < tmp.size(); ++c) {
                    if (c > a && tmp[c].first == tmp[a].first) {
                        // two observations of one landmark in the same keyframe: contributes to the diagonal block twice (a,c) and (c,a)
                        const int q = blk_index(tmp[a].first, tmp[a].first);
                        if (pass == 0) blk_count[q + 1] += 2;
                        else { terms[fill[q]++] = make_int2(tmp[a].second, tmp[c].second); terms[fill[q]++] = make_int2(tmp[c].second, tmp[a].second); }
                        continue;
                    }
                    const int q = blk_index(tmp[a].first, tmp[c].first);
                    if (pass == 0) blk_count[q + 1]++;
                    else terms[fill[q]++] = make_int2(tmp[a].second, tmp[c].second);
                }
        }
        if (pass == 1) {
            int rc = upload(b, &b->d_blk_terms, terms);
            if (rc) { lpslam_hip_ba_destroy(b); return rc; }
        }
    }
    int rc = 0;
    auto fail = [&](int code) { lpslam_hip_ba_destroy(b); return code; };
#define BA_TRY(x) do { rc = (x); if (rc) return fail(rc); } while (0)
    BA_TRY(upload(b, &b->d_blk_start, blk_count));
    BA_TRY(upload(b, &b->d_pose_slot, slot));
    BA_TRY(upload(b, &b->d_free_pose, free_pose));
    BA_TRY(upload(b, &b->d_o_pose, o_pose)); BA_TRY(upload(b, &b->d_o_point, o_point));
    BA_TRY(upload(b, &b->d_o_u, ou)); BA_TRY(upload(b, &b->d_o_v, ov)); BA_TRY(upload(b, &b->d_o_ur, our)); BA_TRY(upload(b, &b->d_o_w, ow));
    BA_TRY(upload(b, &b->d_pt_start, pt_start)); BA_TRY(upload(b, &b->d_pt_obs, pt_obs));
    BA_TRY(upload(b, &b->d_ps_start, ps_start)); BA_TRY(upload(b, &b->d_ps_obs, ps_obs));
    std::vector<uint8_t> act((size_t)std::max(n_obs, 1), 1);
    BA_TRY(upload(b, &b->d_o_active, act));
    for (int s = 0; s < 2; ++s) { BA_TRY(dalloc(b, &b->d_poses[s], 7 * (size_t)n_poses)); BA_TRY(dalloc(b, &b->d_points[s], 3 * (size_t)n_points)); }
    if (hipMemcpy(b->d_poses[0], poses, 7 * (size_t)n_poses * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        (n_points && hipMemcpy(b->d_points[0], points, 3 * (size_t)n_points * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)) {
        set_error("state upload failed"); return fail(LPSLAM_HIP_ERR_DEVICE);
    }
    BA_TRY(dalloc(b, &b->d_poses0, 7 * (size_t)n_poses)); BA_TRY(dalloc(b, &b->d_points0, 3 * (size_t)n_points));
    if (hipMemcpy(b->d_poses0, poses, 7 * (size_t)n_poses * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        (n_points && hipMemcpy(b->d_points0, points, 3 * (size_t)n_points * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)) {
        set_error("state upload failed"); return fail(LPSLAM_HIP_ERR_DEVICE);
    }
    BA_TRY(dalloc(b, &b->d_W, 18 * (size_t)n_obs)); BA_TRY(dalloc(b, &b->d_Y, 18 * (size_t)n_obs)); BA_TRY(dalloc(b, &b->d_Ybl, 6 * (size_t)n_obs));
    BA_TRY(dalloc(b, &b->d_Hll, 6 * (size_t)n_points)); BA_TRY(dalloc(b, &b->d_bl, 3 * (size_t)n_points)); BA_TRY(dalloc(b, &b->d_Hinv, 6 * (size_t)n_points));
    BA_TRY(dalloc(b, &b->d_Hpp, 36 * (size_t)b->n_free));
    b->red_n = (int64_t)b->dim_pad * b->dim_pad + 3 * (int64_t)b->dim_pad + 8;
    BA_TRY(dalloc(b, &b->d_red, (size_t)b->red_n));
    if (hipMemset(b->d_red, 0, (size_t)b->red_n * sizeof(double)) != hipSuccess) { set_error("memset failed"); return fail(LPSLAM_HIP_ERR_DEVICE); }
    BA_TRY(dalloc(b, &b->d_xp, (size_t)b->dim_pad)); BA_TRY(dalloc(b, &b->d_xl, 3 * (size_t)n_points));
    BA_TRY(dalloc(b, &b->d_chi_pose, (size_t)n_poses));
    b->part_n = std::max((n_points + 127) / 128, 1);
    BA_TRY(dalloc(b, &b->d_part, (size_t)b->part_n)); BA_TRY(dalloc(b, &b->d_scal, 8));
    if (hipMemset(b->d_scal, 0, 8 * sizeof(double)) != hipSuccess || hipMemset(b->d_xp, 0, b->dim_pad * sizeof(double)) != hipSuccess) {
        set_error("memset failed"); return fail(LPSLAM_HIP_ERR_DEVICE);
    }
    BA_TRY(dalloc(b, &b->d_chi_obs, (size_t)n_obs)); BA_TRY(dalloc(b, &b->d_depth, (size_t)n_obs));
#undef BA_TRY
    *out = b;
    return LPSLAM_HIP_OK;
}

void lpslam_hip_ba_destroy(lpslam_hip_ba* b)
{
    if (!b) return;
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (void* p : b->allocs) (void)hipFree(p);
    delete b;
}

int lpslam_hip_ba_set_active(lpslam_hip_ba* b, const uint8_t* active)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b->n_obs) return LPSLAM_HIP_OK;
    if (active) LP_HIP(hipMemcpyAsync(b->d_o_active, active, b->n_obs, hipMemcpyHostToDevice, b->stream));
    else LP_HIP(hipMemsetAsync(b->d_o_active, 1, b->n_obs, b->stream));
    LP_HIP(hipStreamSynchronize(b->stream));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_reduced_buffer(lpslam_hip_ba* b, void** dev_ptr, int64_t* n_doubles)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (dev_ptr) *dev_ptr = b->d_red;
    if (n_doubles) *n_doubles = b->red_n;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_scalar_buffer(lpslam_hip_ba* b, void** dev_ptr, int64_t* n_doubles)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (dev_ptr) *dev_ptr = b->d_scal;
    if (n_doubles) *n_doubles = 8;
    return LPSLAM_HIP_OK;
}

// One LM trial in three device phases so that a landmark-partitioned multi-GPU solve can all-reduce in between:
//   step_begin : (first trial of an iteration: linearise) + Schur complement for the current lambda -> reduced buffer
//                [S | rhs | b_p | diag H_pp | chi2] (sum all-reduce) and scal[4] = max diag H_ll (max all-reduce, first only)
//   step_solve : lambda_0 (first iteration), factor, solve, update into the trial state; scal[1] = trial chi2 and
//                scal[2] = landmark part of computeScale (sum all-reduce), scal[3] = pose part (identical on all ranks)
//   step_end   : lambda control; *accepted, *iteration_finished
int lpslam_hip_ba_step_begin(lpslam_hip_ba* b, int32_t robust, int32_t first)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(b->ctx->cfg.device));
    b->robust = robust;
    int rc;
    if (b->qmax == 0 || first) {
        if (first) { b->lambda = -1.0; b->ni = 2; }
        b->qmax = 0;
        if ((rc = ba_linearize(b))) return rc;
    }
    if (b->lambda < 0) {
        // lambda_0 needs max diag over H_pp (all-reduced) and H_ll: Schur complement is built after it is known.
        // Single-GPU callers go through lpslam_hip_ba_optimize, which handles this; the multi-GPU protocol runs
        // step_begin twice on the first trial (first with lambda unknown -> only the linearisation is published).
        BaView v = make_view(b, b->cur);
        (void)v;
        return LPSLAM_HIP_OK;
    }
    return ba_reduce_system(b);
}

int lpslam_hip_ba_step_solve(lpslam_hip_ba* b)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    return ba_solve_update(b);
}

int lpslam_hip_ba_step_end(lpslam_hip_ba* b, int32_t* accepted, int32_t* iteration_finished)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    double h[8];
    int rc = read_scal(b, h); if (rc) return rc;
    int acc = 0;
    const bool done = lm_decide(b, h[1], h[2] + h[3], h[5] == 0.0, &acc);
    if (accepted) *accepted = acc;
    if (iteration_finished) *iteration_finished = done ? 1 : 0;
    if (done) b->qmax = 0;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_optimize(lpslam_hip_ba* b, int32_t robust, int32_t iters, lpslam_hip_ba_iter_log* log, int32_t* done_out)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(b->ctx->cfg.device));
    b->robust = robust;
    int it = 0, rc;
    for (; it < iters; ++it) {
        if ((rc = ba_linearize(b))) return rc;
        double h[8];
        // chi2 of the accepted state and the diagonal maxima
        std::vector<double> tail(b->dim_pad + 1);
        BaView v = make_view(b, b->cur);
        LP_HIP(hipMemcpyAsync(tail.data(), v.hppdiag, (b->dim_pad + 1) * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        if ((rc = read_scal(b, h))) return rc;
        b->current_chi = tail[b->dim_pad];
        if (it == 0) {
            double maxd = b->n_points ? h[4] : 0.0;
            for (int i = 0; i < b->dim; ++i) maxd = std::max(maxd, std::fabs(tail[i]));
            b->lambda = 1e-5 * maxd;
            b->ni = 2;
        }
        b->chi_before = b->current_chi;
        b->qmax = 0;
        bool finished = false;
        while (!finished) {
            if ((rc = ba_reduce_system(b))) return rc;
            if ((rc = ba_solve_update(b))) return rc;
            if ((rc = read_scal(b, h))) return rc;
            int acc;
            finished = lm_decide(b, h[1], h[2] + h[3], h[5] == 0.0, &acc);
        }
        const bool terminate = (b->qmax == 10 || b->rho == 0);
        if (log) { log[it].chi2_before = b->chi_before; log[it].chi2_after = b->current_chi; log[it].lambda = b->lambda; log[it].trials = b->qmax; log[it].status = terminate; }
        if (terminate) { ++it; break; }
    }
    if (done_out) *done_out = it;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_reset(lpslam_hip_ba* b)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    b->cur = 0; b->lambda = 0; b->ni = 2; b->qmax = 0; b->rho = 0;
    LP_HIP(hipMemcpyAsync(b->d_poses[0], b->d_poses0, 7 * (size_t)b->n_poses * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
    if (b->n_points) LP_HIP(hipMemcpyAsync(b->d_points[0], b->d_points0, 3 * (size_t)b->n_points * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
    if (b->n_obs) LP_HIP(hipMemsetAsync(b->d_o_active, 1, b->n_obs, b->stream));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_get(lpslam_hip_ba* b, double* poses, double* points)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (poses) LP_HIP(hipMemcpyAsync(poses, b->d_poses[b->cur], 7 * (size_t)b->n_poses * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    if (points && b->n_points) LP_HIP(hipMemcpyAsync(points, b->d_points[b->cur], 3 * (size_t)b->n_points * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    LP_HIP(hipStreamSynchronize(b->stream));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_chi2(lpslam_hip_ba* b, double* chi2, uint8_t* depth_positive)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b->n_obs) return LPSLAM_HIP_OK;
    BaView v = make_view(b, b->cur);
    hipLaunchKernelGGL(k_ba_obs_chi2, dim3((b->n_obs + 255) / 256), dim3(256), 0, b->stream, v, b->d_chi_obs, b->d_depth);
    LP_HIP(hipGetLastError());
    if (chi2) LP_HIP(hipMemcpyAsync(chi2, b->d_chi_obs, (size_t)b->n_obs * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    if (depth_positive) LP_HIP(hipMemcpyAsync(depth_positive, b->d_depth, (size_t)b->n_obs, hipMemcpyDeviceToHost, b->stream));
    LP_HIP(hipStreamSynchronize(b->stream));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_set_points_fixed(lpslam_hip_ba* b, int32_t points_fixed)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    b->points_fixed = points_fixed ? 1 : 0;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_pose_optimize(lpslam_hip_ba* b, uint8_t* outlier, int32_t* n_inliers)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    const int n = b->n_obs;
    std::vector<uint8_t> active((size_t)std::max(n, 1), 1), out((size_t)std::max(n, 1), 0);
    std::vector<double> chi((size_t)std::max(n, 1));
    const int keep_fixed = b->points_fixed;
    b->points_fixed = 1;
    int rc, done, bad = 0, robust = 1;
    if ((rc = lpslam_hip_ba_set_active(b, nullptr))) return rc;
    for (int trial = 0; trial < 4; ++trial) {
        if ((rc = lpslam_hip_ba_optimize(b, robust, 10, nullptr, &done))) return rc;
        if ((rc = lpslam_hip_ba_chi2(b, chi.data(), nullptr))) return rc;
        bad = 0;
        for (int k = 0; k < n; ++k) {
            const double thr = b->h_ur[k] < 0 ? 5.99146 : 7.81473;
            if (thr < chi[k]) { out[k] = 1; active[k] = 0; ++bad; } else { out[k] = 0; active[k] = 1; }
        }
        if ((rc = lpslam_hip_ba_set_active(b, active.data()))) return rc;
        if (trial == 4 - 2) robust = 0;
        if (n - bad < 5) break;
    }
    b->points_fixed = keep_fixed;
    if (outlier) std::copy(out.begin(), out.begin() + n, outlier);
    if (n_inliers) *n_inliers = n - bad;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_local(lpslam_hip_ba* b, int32_t first_iters, int32_t second_iters, uint8_t* outlier)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    const int n = b->n_obs;
    std::vector<uint8_t> active((size_t)std::max(n, 1), 1), pos((size_t)std::max(n, 1));
    std::vector<double> chi((size_t)std::max(n, 1));
    int rc, done;
    if ((rc = lpslam_hip_ba_set_active(b, nullptr))) return rc;
    if ((rc = lpslam_hip_ba_optimize(b, 1, first_iters, nullptr, &done))) return rc;
    if ((rc = lpslam_hip_ba_chi2(b, chi.data(), pos.data()))) return rc;
    for (int k = 0; k < n; ++k) { const double thr = b->h_ur[k] < 0 ? 5.99146 : 7.81473; if (thr < chi[k] || !pos[k]) active[k] = 0; }
    if ((rc = lpslam_hip_ba_set_active(b, active.data()))) return rc;
    if ((rc = lpslam_hip_ba_optimize(b, 0, second_iters, nullptr, &done))) return rc;
    if ((rc = lpslam_hip_ba_chi2(b, chi.data(), pos.data()))) return rc;
    if (outlier) for (int k = 0; k < n; ++k) { const double thr = b->h_ur[k] < 0 ? 5.99146 : 7.81473; outlier[k] = (!active[k]) || (thr < chi[k]) || !pos[k]; }
    return LPSLAM_HIP_OK;
}

}  // extern "C"
