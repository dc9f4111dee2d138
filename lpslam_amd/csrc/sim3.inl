// sim3.inl -- Sim3 pose-graph optimisation on gfx950 (FP64); included at the end of ba.hip (same translation unit: it reuses the
// device-driven Levenberg control block, the blocked Cholesky kernels and the last-workgroup hand-over defined there).
//
// [UPSTREAM] g2o@691dc51 types/sim3 (Sim3 exp / log / inverse / product, VertexSim3Expmap::oplusImpl, EdgeSim3::computeError,
// numeric BaseBinaryEdge::linearizeOplus with delta = 1e-9), OptimizationAlgorithmLevenberg with BlockSolver_7_3, and
// OpenVSLAM's optimize::graph_optimizer (identity information, loop keyframe fixed, scale fixed for stereo), which the reference
// reaches through openvslam::system when the loop detector is on (/root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:250-255).
// SURVEY.md section 8(a) row a23.
//
// One LM unit = [k_sim3_lin if the state changed] k_sim3_assemble, k_chol_pair x nb/2, k_chol_xsolve, k_sim3_update, k_sim3_trial.
//   k_sim3_lin       one workgroup per edge: 28 perturbed error evaluations side by side (central differences of both
//                    vertices), J_i, J_j, then the edge's J^T J blocks / J^T e; the last workgroup totals chi2, finds
//                    max diag H and starts the outer iteration (lambda_0)
//   k_sim3_assemble  one wavefront per 7x7 block pair of the dense H + lambda I (edge lists in edge order), rhs as row `dim`
//   k_sim3_update    trial vertices = exp(x) * estimate, scale term x^T (lambda x + b)
//   k_sim3_trial     chi2 of the trial state; the last workgroup runs g2o's lambda control
// The essential graph has a few edges per keyframe, but the Cholesky of H fills in; H is kept dense (200 keyframes: 1408^2
// doubles = 15.9 MB of 288 GB) and factored by the same 32-wide panel kernels as the reduced camera system of the BA.

namespace {

constexpr int S3_EB = 162;            // per-edge block storage: A_ii 49 | A_ij 49 | A_jj 49 | g_i 7 | g_j 7 | chi2 1

struct Sim3d { double q[4], t[3], s; };

struct Sim3View {
    int n, n_free, n_edges, dim, dim_pad, fix_scale, n_blocks;
    double* verts_buf[2];
    const int* slot; const int* free_vert;
    const int* e_i; const int* e_j; const double* meas;
    double* eblk;
    const int* vt_start; const int2* vt_inc;          // per free vertex: (edge, side)
    const int* blk_start; const int2* blk_terms;      // per block pair (i <= k): (edge, transposed)
    const int2* blk_ik;                               // block pair -> (i, k)
    double* bp; double* part;
    BaView cv;                                        // the view the Cholesky kernels take (S, Minv, Ldiag, xp, scal, ctl, log)
};

__device__ __forceinline__ void s3_load(const double* p, Sim3d& s)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) s.q[i] = p[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) s.t[i] = p[4 + i];
    s.s = p[7];
}
__device__ __forceinline__ void s3_store(const Sim3d& s, double* p)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = s.q[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) p[4 + i] = s.t[i];
    p[7] = s.s;
}
__device__ __forceinline__ void s3_q_to_R(const double* q, double* R)
{
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void s3_R_to_q(const double* R, double* q)
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else {
        // largest diagonal element first (Eigen's Quaternion(Matrix3)); written out so that no array is indexed dynamically
        if (R[0] >= R[4] && R[0] >= R[8]) {
            t = sqrt(R[0] - R[4] - R[8] + 1.0);
            q[1] = 0.5 * t; t = 0.5 / t;
            q[0] = (R[7] - R[5]) * t; q[2] = (R[3] + R[1]) * t; q[3] = (R[6] + R[2]) * t;
        } else if (R[4] >= R[8]) {
            t = sqrt(R[4] - R[8] - R[0] + 1.0);
            q[2] = 0.5 * t; t = 0.5 / t;
            q[0] = (R[2] - R[6]) * t; q[3] = (R[7] + R[5]) * t; q[1] = (R[1] + R[3]) * t;
        } else {
            t = sqrt(R[8] - R[0] - R[4] + 1.0);
            q[3] = 0.5 * t; t = 0.5 / t;
            q[0] = (R[3] - R[1]) * t; q[1] = (R[2] + R[6]) * t; q[2] = (R[5] + R[7]) * t;
        }
    }
}
__device__ __forceinline__ void s3_q_mul(const double* a, const double* b, double* o)
{
    const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    const double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
__device__ __forceinline__ void s3_q_rot(const double* q, const double* v, double* o)
{
    const double ux = 2 * (q[2] * v[2] - q[3] * v[1]), uy = 2 * (q[3] * v[0] - q[1] * v[2]), uz = 2 * (q[1] * v[1] - q[2] * v[0]);
    o[0] = v[0] + q[0] * ux + (q[2] * uz - q[3] * uy);
    o[1] = v[1] + q[0] * uy + (q[3] * ux - q[1] * uz);
    o[2] = v[2] + q[0] * uz + (q[1] * uy - q[2] * ux);
}
__device__ __forceinline__ void s3_mat3_mul(const double* A, const double* B, double* C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void s3_skew(const double* w, double* O)
{
    O[0] = 0; O[1] = -w[2]; O[2] = w[1]; O[3] = w[2]; O[4] = 0; O[5] = -w[0]; O[6] = -w[1]; O[7] = w[0]; O[8] = 0;
}
// the coefficients A, B, C of W = A Omega + B Omega^2 + C I shared by exp and log (g2o sim3.h)
__device__ __forceinline__ void s3_abc(double sigma, double s, double theta, bool small_theta, double* A, double* B, double* C)
{
    const double eps = 0.00001;
    if (fabs(sigma) < eps) {
        *C = 1;
        if (small_theta) { *A = 0.5; *B = 1. / 6.; }
        else {
            const double theta2 = theta * theta;
            *A = (1 - cos(theta)) / theta2;
            *B = (theta - sin(theta)) / (theta2 * theta);
        }
    } else {
        *C = (s - 1) / sigma;
        if (small_theta) {
            const double sigma2 = sigma * sigma;
            *A = ((sigma - 1) * s + 1) / sigma2;
            *B = ((0.5 * sigma2 - sigma + 1) * s - 1) / (sigma2 * sigma);
        } else {
            const double a = s * sin(theta), b = s * cos(theta);
            const double theta2 = theta * theta, sigma2 = sigma * sigma, c = theta2 + sigma2;
            *A = (a * sigma + (1 - b) * theta) / (theta * c);
            *B = (*C - ((b - 1) * sigma + a * theta) / c) * 1 / theta2;
        }
    }
}
__device__ void s3_exp(const double* u, Sim3d& o)
{
    const double sigma = u[6];
    const double theta = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    double Om[9], Om2[9], R[9];
    s3_skew(u, Om);
    s3_mat3_mul(Om, Om, Om2);
    o.s = exp(sigma);
    const bool small = theta < 0.00001;
    double A, B, C;
    s3_abc(sigma, o.s, theta, small, &A, &B, &C);
    const double a1 = small ? 1.0 : sin(theta) / theta, a2 = small ? 1.0 : (1 - cos(theta)) / (theta * theta);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a1 * Om[i] + a2 * Om2[i];
    s3_R_to_q(R, o.q);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double acc = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) acc += (A * Om[i * 3 + j] + B * Om2[i * 3 + j] + (i == j ? C : 0.0)) * u[3 + j];
        o.t[i] = acc;
    }
}
// x = W^-1 t, LU with partial pivoting (Eigen PartialPivLU); row swaps by value so that everything stays in registers
__device__ __forceinline__ void s3_lu_solve3(const double* Win, const double* t, double* x)
{
    double r0[4] = {Win[0], Win[1], Win[2], t[0]}, r1[4] = {Win[3], Win[4], Win[5], t[1]}, r2[4] = {Win[6], Win[7], Win[8], t[2]};
    auto swap4 = [](double* a, double* b) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const double tmp = a[c]; a[c] = b[c]; b[c] = tmp; } };
    {   // column 0: pivot = first row of maximal magnitude
        int piv = 0;
        if (fabs(r1[0]) > fabs(r0[0])) piv = 1;
        if (fabs(r2[0]) > fabs(piv == 1 ? r1[0] : r0[0])) piv = 2;
        if (piv == 1) swap4(r0, r1); else if (piv == 2) swap4(r0, r2);
    }
    const double f1 = r1[0] / r0[0], f2 = r2[0] / r0[0];
    r1[1] -= f1 * r0[1]; r1[2] -= f1 * r0[2];
    r2[1] -= f2 * r0[1]; r2[2] -= f2 * r0[2];
    double g1 = f1, g2 = f2;
    if (fabs(r2[1]) > fabs(r1[1])) { swap4(r1, r2); const double tmp = g1; g1 = g2; g2 = tmp; }
    const double f3 = r2[1] / r1[1];
    r2[2] -= f3 * r1[2];
    const double b0 = r0[3], b1 = r1[3] - g1 * b0, b2 = (r2[3] - g2 * b0) - f3 * b1;
    x[2] = b2 / r2[2];
    x[1] = (b1 - r1[2] * x[2]) / r1[1];
    x[0] = ((b0 - r0[1] * x[1]) - r0[2] * x[2]) / r0[0];
}
__device__ void s3_log(const Sim3d& S, double* res)
{
    const double sigma = log(S.s);
    double R[9], omega[3], Om[9], Om2[9];
    s3_q_to_R(S.q, R);
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const bool small = d > 1 - 0.00001;
    const double theta = small ? 0.0 : acos(d);
    const double f = small ? 0.5 : theta / (2 * sqrt(1 - d * d));
#pragma unroll
    for (int i = 0; i < 3; ++i) omega[i] = f * dR[i];
    double A, B, C;
    s3_abc(sigma, S.s, theta, small, &A, &B, &C);
    s3_skew(omega, Om);
    s3_mat3_mul(Om, Om, Om2);
    double W[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) W[i] = A * Om[i] + B * Om2[i] + (i % 4 == 0 ? C : 0.0);
    s3_lu_solve3(W, S.t, res + 3);
#pragma unroll
    for (int i = 0; i < 3; ++i) res[i] = omega[i];
    res[6] = sigma;
}
__device__ __forceinline__ void s3_mul(const Sim3d& a, const Sim3d& b, Sim3d& o)
{
    Sim3d r;
    s3_q_mul(a.q, b.q, r.q);
    double rt[3];
    s3_q_rot(a.q, b.t, rt);
#pragma unroll
    for (int i = 0; i < 3; ++i) r.t[i] = a.s * rt[i] + a.t[i];
    r.s = a.s * b.s;
    o = r;
}
__device__ __forceinline__ void s3_inv(const Sim3d& a, Sim3d& o)
{
    Sim3d r;
    r.q[0] = a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = -a.q[3];
    const double v[3] = {(-1. / a.s) * a.t[0], (-1. / a.s) * a.t[1], (-1. / a.s) * a.t[2]};
    s3_q_rot(r.q, v, r.t);
    r.s = 1. / a.s;
    o = r;
}
__device__ void s3_edge_error(const Sim3d& meas, const Sim3d& vi, const Sim3d& vj, double* e)
{
    Sim3d inv, t1, t2;
    s3_inv(vj, inv);
    s3_mul(meas, vi, t1);
    s3_mul(t1, inv, t2);
    s3_log(t2, e);
}
__device__ void s3_oplus(const Sim3d& est, const double* update, int fix_scale, Sim3d& out)
{
    double u[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) u[i] = update[i];
    if (fix_scale) u[6] = 0;
    Sim3d d;
    s3_exp(u, d);
    s3_mul(d, est, out);
}

// ---- linearisation: one workgroup (64 threads) per edge ------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_sim3_lin(Sim3View v)
{
    BaCtl* ctl = v.cv.ctl;
    if (ba_idle(ctl) || !ctl->need_lin) return;
    const double* verts = ctl->cur ? v.verts_buf[1] : v.verts_buf[0];      // selects: see ba_lin_set
    const int k = blockIdx.x, t = threadIdx.x;
    __shared__ double err[29][7];
    __shared__ double J[2][7][7];
    Sim3d m, vi, vj;
    s3_load(v.meas + 8 * (size_t)k, m);
    s3_load(verts + 8 * (size_t)v.e_i[k], vi);
    s3_load(verts + 8 * (size_t)v.e_j[k], vj);
    if (t < 29) {
        double e[7];
        if (t < 28) {
            const int which = t / 14, d = (t % 14) >> 1;
            double add[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) add[i] = (i == d) ? ((t & 1) ? -1e-9 : 1e-9) : 0.0;
            Sim3d pert;
            s3_oplus(which == 0 ? vi : vj, add, v.fix_scale, pert);
            if (which == 0) s3_edge_error(m, pert, vj, e); else s3_edge_error(m, vi, pert, e);
        } else {
            s3_edge_error(m, vi, vj, e);
        }
#pragma unroll
        for (int r = 0; r < 7; ++r) err[t][r] = e[r];
    }
    __syncthreads();
    const double scalar = 1.0 / (2 * 1e-9);
    for (int idx = t; idx < 98; idx += 64) {
        const int which = idx / 49, r = (idx % 49) / 7, d = idx % 7;
        J[which][r][d] = scalar * (err[which * 14 + d * 2][r] - err[which * 14 + d * 2 + 1][r]);
    }
    __syncthreads();
    double* out = v.eblk + (size_t)k * S3_EB;
    for (int idx = t; idx < S3_EB; idx += 64) {
        double s = 0;
        if (idx < 147) {
            const int blk = idx / 49, a = (idx % 49) / 7, c = idx % 7;
            const int wa = blk == 2 ? 1 : 0, wc = blk == 0 ? 0 : 1;
            for (int r = 0; r < 7; ++r) s += J[wa][r][a] * J[wc][r][c];
        } else if (idx < 161) {
            const int which = (idx - 147) / 7, a = (idx - 147) % 7;
            for (int r = 0; r < 7; ++r) s += J[which][r][a] * err[28][r];
        } else {
            for (int r = 0; r < 7; ++r) s += err[28][r] * err[28][r];
        }
        out[idx] = s;
    }
    if (!ba_last_block(ctl, gridDim.x)) return;
    // last workgroup: chi2 of the accepted state, max |diag H|, start of the outer iteration
    double chi = 0, md = 0;
    for (int e = t; e < v.n_edges; e += 64) chi += v.eblk[(size_t)e * S3_EB + 161];
    for (int i = t; i < v.dim; i += 64) {
        const int f = i / 7, a = i - 7 * f;
        double s = 0;
        for (int q = v.vt_start[f]; q < v.vt_start[f + 1]; ++q) {
            const int2 es = v.vt_inc[q];
            s += v.eblk[(size_t)es.x * S3_EB + (es.y ? 98 : 0) + a * 8];
        }
        md = fmax(md, fabs(s));
    }
    chi = wave_sum(chi);
    md = wave_max(md);
    if (t == 0) { v.cv.scal[0] = chi; lm_begin(v.cv, md, 0.0, chi); }
}

// ---- per trial: dense H + lambda I (one wavefront per 7x7 block pair), rhs into row `dim` --------------------------------
__global__ __launch_bounds__(64) void k_sim3_assemble(Sim3View v)
{
    BaCtl* ctl = v.cv.ctl;
    if (ba_idle(ctl)) return;
    const double lambda = ctl->lambda;
    const int t = threadIdx.x, n = v.dim_pad;
    double* S = v.cv.S;
    if ((int)blockIdx.x >= v.n_blocks) {
        const int f = blockIdx.x - v.n_blocks;
        if (t < 7) {
            double s = 0;
            for (int q = v.vt_start[f]; q < v.vt_start[f + 1]; ++q) {
                const int2 es = v.vt_inc[q];
                s += v.eblk[(size_t)es.x * S3_EB + (es.y ? 154 : 147) + t];
            }
            v.bp[7 * f + t] = -s;
            S[(size_t)v.dim * n + 7 * f + t] = -s;
        }
        if (f == 0 && t == 63) { S[(size_t)v.dim * n + v.dim] = 1e200; v.cv.scal[5] = 0.0; }
        return;
    }
    if (t >= 49) return;
    const int2 ik = v.blk_ik[blockIdx.x];
    const int i = ik.x, k = ik.y, r = t / 7, c = t - 7 * r;
    double s = 0;
    if (i == k) {
        for (int q = v.vt_start[i]; q < v.vt_start[i + 1]; ++q) {
            const int2 es = v.vt_inc[q];
            s += v.eblk[(size_t)es.x * S3_EB + (es.y ? 98 : 0) + t];
        }
        S[(size_t)(7 * i + r) * n + 7 * i + c] = s + (r == c ? lambda : 0.0);
    } else {
        for (int q = v.blk_start[blockIdx.x]; q < v.blk_start[blockIdx.x + 1]; ++q) {
            const int2 et = v.blk_terms[q];
            s += v.eblk[(size_t)et.x * S3_EB + 49 + (et.y ? c * 7 + r : t)];
        }
        S[(size_t)(7 * i + r) * n + 7 * k + c] = s;
        S[(size_t)(7 * k + c) * n + 7 * i + r] = s;
    }
}

// ---- trial vertices and the scale term x^T (lambda x + b); the last block holds the reduction --------------------------------
__global__ __launch_bounds__(256) void k_sim3_update(Sim3View v, int vert_blocks)
{
    BaCtl* ctl = v.cv.ctl;
    if (ba_idle(ctl)) return;
    const int cur = ctl->cur;
    const double* verts = cur ? v.verts_buf[1] : v.verts_buf[0];
    double* out = cur ? v.verts_buf[0] : v.verts_buf[1];
    const double* xp = v.cv.xp;
    if ((int)blockIdx.x < vert_blocks) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= v.n) return;
        Sim3d est;
        s3_load(verts + 8 * (size_t)i, est);
        const int f = v.slot[i];
        if (f >= 0) { Sim3d upd; s3_oplus(est, xp + 7 * (size_t)f, v.fix_scale, upd); est = upd; }
        s3_store(est, out + 8 * (size_t)i);
        return;
    }
    const double lambda = ctl->lambda;
    __shared__ double sm[4];
    double sc = 0;
    for (int j = threadIdx.x; j < v.dim; j += 256) { const double x = xp[j]; sc += x * (lambda * x + v.bp[j]); }
    sc = wave_sum(sc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = sc;
    __syncthreads();
    if (threadIdx.x == 0) v.cv.scal[3] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

// ---- chi2 of the trial state (thread per edge); the last workgroup runs the lambda control --------------------------------
__global__ __launch_bounds__(256) void k_sim3_trial(Sim3View v)
{
    BaCtl* ctl = v.cv.ctl;
    if (ba_idle(ctl)) return;
    const double* verts = ctl->cur ? v.verts_buf[0] : v.verts_buf[1];
    const int k = blockIdx.x * 256 + threadIdx.x;
    double chi = 0;
    if (k < v.n_edges) {
        Sim3d m, vi, vj; double e[7];
        s3_load(v.meas + 8 * (size_t)k, m);
        s3_load(verts + 8 * (size_t)v.e_i[k], vi);
        s3_load(verts + 8 * (size_t)v.e_j[k], vj);
        s3_edge_error(m, vi, vj, e);
#pragma unroll
        for (int r = 0; r < 7; ++r) chi += e[r] * e[r];
    }
    __shared__ double sm[4];
    chi = wave_sum(chi);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = chi;
    __syncthreads();
    if (threadIdx.x == 0) v.part[blockIdx.x] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
    if (!ba_last_block(ctl, gridDim.x)) return;
    if (threadIdx.x < 64) {
        double acc = 0;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) acc += v.part[i];
        acc = wave_sum(acc);
        if (threadIdx.x == 0) {
            const double fail = v.cv.scal[5], scale_p = v.cv.scal[3];
            v.cv.scal[1] = acc;
            lm_decide(v.cv, acc, fail, 0.0, scale_p);
        }
    }
}

// chi2 per edge of the accepted state (API read-back)
__global__ __launch_bounds__(256) void k_sim3_chi2(Sim3View v, double* chi2)
{
    const double* verts = v.cv.ctl->cur ? v.verts_buf[1] : v.verts_buf[0];
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= v.n_edges) return;
    Sim3d m, vi, vj; double e[7];
    s3_load(v.meas + 8 * (size_t)k, m);
    s3_load(verts + 8 * (size_t)v.e_i[k], vi);
    s3_load(verts + 8 * (size_t)v.e_j[k], vj);
    s3_edge_error(m, vi, vj, e);
    double chi = 0;
#pragma unroll
    for (int r = 0; r < 7; ++r) chi += e[r] * e[r];
    chi2[k] = chi;
}

// ---- Sim3 between two keyframes ([UPSTREAM] optimize::transform_optimizer / ORB-SLAM2 OptimizeSim3) --------------------------
// One workgroup per candidate pair of keyframes runs the whole flow -- 5 Levenberg iterations, outlier cut, 5 / 10 more --
// without returning to the host: the 7x7 system lives in LDS, the 14 perturbed transforms of the numeric Jacobian are built
// once per iteration and shared by all pairs, sums are butterflies + the four wavefronts in order (reproducible).
struct S3Pair { double p1c[3], p2c[3], obs1[2], obs2[2], w1, w2; };

__device__ __forceinline__ void s3_t_error(const Sim3d& S, const Sim3d& Sinv, const S3Pair& p, const double* c1, const double* c2, double* e)
{
    double r[3], x[3];
    s3_q_rot(S.q, p.p2c, r);
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = S.s * r[i] + S.t[i];
    e[0] = p.obs1[0] - (c1[0] * x[0] / x[2] + c1[2]);
    e[1] = p.obs1[1] - (c1[1] * x[1] / x[2] + c1[3]);
    s3_q_rot(Sinv.q, p.p1c, r);
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = Sinv.s * r[i] + Sinv.t[i];
    e[2] = p.obs2[0] - (c2[0] * x[0] / x[2] + c2[2]);
    e[3] = p.obs2[1] - (c2[1] * x[1] / x[2] + c2[3]);
}

struct S3TShared {
    Sim3d S, Si, St, Sti, Sp[14], Spi[14];
    double red[4][36];
    double H[49], b[7], x[7];
    double lambda, ni, current_chi, rho;
    int ok, accepted, again, stop, n_bad, n_in;
};

// sum of one value over the workgroup, returned to every thread (fixed order)
__device__ __forceinline__ double s3_block_sum(double v, S3TShared& sh)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6][35] = v;
    __syncthreads();
    return ((sh.red[0][35] + sh.red[1][35]) + sh.red[2][35]) + sh.red[3][35];
}

__device__ double s3_t_chi2(const Sim3d& S, const Sim3d& Si, const S3Pair* pairs, const uint8_t* active, int n, const double* c1, const double* c2,
                            double delta, S3TShared& sh)
{
    double chi = 0;
    for (int k = threadIdx.x; k < n; k += 256) {
        if (!active[k]) continue;
        double e[4], r0, r1;
        s3_t_error(S, Si, pairs[k], c1, c2, e);
        huber(pairs[k].w1 * (e[0] * e[0] + e[1] * e[1]), delta, &r0, &r1); chi += r0;
        huber(pairs[k].w2 * (e[2] * e[2] + e[3] * e[3]), delta, &r0, &r1); chi += r0;
    }
    return s3_block_sum(chi, sh);
}

__device__ void s3_t_levenberg(const S3Pair* pairs, const uint8_t* active, int n, const double* c1, const double* c2, double delta,
                               int fix_scale, int iters, S3TShared& sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sh.stop = 0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (sh.stop) break;                                            // uniform: written before the barrier that ends an iteration
        const double cur = s3_t_chi2(sh.S, sh.Si, pairs, active, n, c1, c2, delta, sh);
        if (tid < 14) {
            double add[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) add[i] = (i == (tid >> 1)) ? ((tid & 1) ? -1e-9 : 1e-9) : 0.0;
            Sim3d pert, pinv;
            s3_oplus(sh.S, add, fix_scale, pert);
            s3_inv(pert, pinv);
            sh.Sp[tid] = pert; sh.Spi[tid] = pinv;
        }
        __syncthreads();
        double acc[35];                                                // upper triangle of H (28) and b (7)
#pragma unroll
        for (int q = 0; q < 35; ++q) acc[q] = 0;
        for (int k = tid; k < n; k += 256) {
            if (!active[k]) continue;
            const S3Pair p = pairs[k];
            double e[4], J[4][7];
            s3_t_error(sh.S, sh.Si, p, c1, c2, e);
#pragma unroll
            for (int d = 0; d < 7; ++d) {
                double e1[4], e2[4];
                s3_t_error(sh.Sp[2 * d], sh.Spi[2 * d], p, c1, c2, e1);
                s3_t_error(sh.Sp[2 * d + 1], sh.Spi[2 * d + 1], p, c1, c2, e2);
#pragma unroll
                for (int r = 0; r < 4; ++r) J[r][d] = (1.0 / (2 * 1e-9)) * (e1[r] - e2[r]);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const double om = half ? p.w2 : p.w1;
                double r0, r1;
                huber(om * (e[2 * half] * e[2 * half] + e[2 * half + 1] * e[2 * half + 1]), delta, &r0, &r1);
                const double w = om * r1;
                int idx = 0;
#pragma unroll
                for (int a = 0; a < 7; ++a) {
#pragma unroll
                    for (int c = a; c < 7; ++c) acc[idx++] += J[2 * half][a] * w * J[2 * half][c] + J[2 * half + 1][a] * w * J[2 * half + 1][c];
                    acc[28 + a] -= J[2 * half][a] * w * e[2 * half] + J[2 * half + 1][a] * w * e[2 * half + 1];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 35; ++q) { const double sq = wave_sum(acc[q]); if (lane == 0) sh.red[wave][q] = sq; }
        __syncthreads();
        if (tid == 0) {
            int idx = 0;
            double maxd = 0;
            for (int a = 0; a < 7; ++a) {
                for (int c = a; c < 7; ++c, ++idx) {
                    const double hv = ((sh.red[0][idx] + sh.red[1][idx]) + sh.red[2][idx]) + sh.red[3][idx];
                    sh.H[a * 7 + c] = hv; sh.H[c * 7 + a] = hv;
                }
                sh.b[a] = ((sh.red[0][28 + a] + sh.red[1][28 + a]) + sh.red[2][28 + a]) + sh.red[3][28 + a];
                maxd = fmax(maxd, fabs(sh.H[a * 8]));
            }
            if (it == 0) { sh.lambda = 1e-5 * maxd; sh.ni = 2; }
            sh.current_chi = cur;
        }
        __syncthreads();
        for (int qmax = 1; qmax <= 10; ++qmax) {
            if (tid == 0) {
                // 7x7 Cholesky + substitutions fully unrolled with constant indices: registers, not scratch memory
                double A[49];
#pragma unroll
                for (int i = 0; i < 49; ++i) A[i] = sh.H[i];
#pragma unroll
                for (int j = 0; j < 7; ++j) A[j * 8] += sh.lambda;
                int ok = 1;
#pragma unroll
                for (int j = 0; j < 7; ++j) {                           // dense Cholesky, lower
                    double d = A[j * 7 + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) d -= A[j * 7 + k] * A[j * 7 + k];
                    if (!(d > 0.0)) { ok = 0; d = 1.0; }                // keep going on harmless numbers; the result is discarded
                    d = sqrt(d);
                    A[j * 7 + j] = d;
#pragma unroll
                    for (int i = j + 1; i < 7; ++i) {
                        double s2 = A[i * 7 + j];
#pragma unroll
                        for (int k = 0; k < j; ++k) s2 -= A[i * 7 + k] * A[j * 7 + k];
                        A[i * 7 + j] = s2 / d;
                    }
                }
                if (ok) {
                    double x[7];
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        double s2 = sh.b[i];
#pragma unroll
                        for (int k = 0; k < i; ++k) s2 -= A[i * 7 + k] * x[k];
                        x[i] = s2 / A[i * 8];
                    }
#pragma unroll
                    for (int i = 6; i >= 0; --i) {
                        double s2 = x[i];
#pragma unroll
                        for (int k = i + 1; k < 7; ++k) s2 -= A[k * 7 + i] * x[k];
                        x[i] = s2 / A[i * 8];
                    }
                    for (int i = 0; i < 7; ++i) sh.x[i] = x[i];
                    Sim3d upd, uinv;
                    s3_oplus(sh.S, x, fix_scale, upd);
                    s3_inv(upd, uinv);
                    sh.St = upd; sh.Sti = uinv;
                } else { sh.St = sh.S; sh.Sti = sh.Si; }
                sh.ok = ok;
            }
            __syncthreads();
            double temp = s3_t_chi2(sh.St, sh.Sti, pairs, active, n, c1, c2, delta, sh);
            if (tid == 0) {
                if (!sh.ok) temp = DBL_MAX;
                double rho = sh.current_chi - temp, scale = 0;
                if (sh.ok) for (int j = 0; j < 7; ++j) scale += sh.x[j] * (sh.lambda * sh.x[j] + sh.b[j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && isfinite(temp)) {
                    const double t3 = 2 * rho - 1;
                    double alpha = 1. - t3 * t3 * t3;
                    alpha = fmin(alpha, 2. / 3.);
                    sh.lambda *= fmax(1. / 3., alpha);
                    sh.ni = 2;
                    sh.current_chi = temp;
                    sh.S = sh.St; sh.Si = sh.Sti;
                } else {
                    sh.lambda *= sh.ni; sh.ni *= 2;
                }
                sh.rho = rho;
                sh.again = (rho < 0 && qmax < 10) ? 1 : 0;
                if (!sh.again && (qmax == 10 || rho == 0)) sh.stop = 1;
            }
            __syncthreads();
            if (!sh.again) break;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_sim3_transform(double* s12, const S3Pair* pairs_all, const int* pair_start, const double* cams,
                                                        double chi_sq, int fix_scale, uint8_t* active_all, uint8_t* inlier_all, int* n_inliers)
{
    __shared__ S3TShared sh;
    const int pb = blockIdx.x, tid = threadIdx.x;
    const S3Pair* pairs = pairs_all + pair_start[pb];
    const int n = pair_start[pb + 1] - pair_start[pb];
    uint8_t* active = active_all + pair_start[pb];
    uint8_t* inlier = inlier_all + pair_start[pb];
    const double* c1 = cams; const double* c2 = cams + 4;
    const double delta = sqrt(chi_sq);
    if (tid == 0) { Sim3d S, Si; s3_load(s12 + 8 * (size_t)pb, S); s3_inv(S, Si); sh.S = S; sh.Si = Si; sh.n_bad = 0; sh.n_in = 0; }
    for (int k = tid; k < n; k += 256) active[k] = 1;
    __syncthreads();
    s3_t_levenberg(pairs, active, n, c1, c2, delta, fix_scale, 5, sh);
    int bad = 0;
    for (int k = tid; k < n; k += 256) {
        double e[4];
        s3_t_error(sh.S, sh.Si, pairs[k], c1, c2, e);
        if (pairs[k].w1 * (e[0] * e[0] + e[1] * e[1]) > chi_sq || pairs[k].w2 * (e[2] * e[2] + e[3] * e[3]) > chi_sq) { active[k] = 0; ++bad; }
    }
    const int n_bad = (int)s3_block_sum((double)bad, sh);
    int n_in = 0;
    if (n - n_bad >= 10) {
        s3_t_levenberg(pairs, active, n, c1, c2, delta, fix_scale, n_bad > 0 ? 10 : 5, sh);
        int in_cnt = 0;
        for (int k = tid; k < n; k += 256) {
            int in = 0;
            if (active[k]) {
                double e[4];
                s3_t_error(sh.S, sh.Si, pairs[k], c1, c2, e);
                in = !(pairs[k].w1 * (e[0] * e[0] + e[1] * e[1]) > chi_sq || pairs[k].w2 * (e[2] * e[2] + e[3] * e[3]) > chi_sq);
            }
            inlier[k] = (uint8_t)in;
            in_cnt += in;
        }
        n_in = (int)s3_block_sum((double)in_cnt, sh);
    } else {
        for (int k = tid; k < n; k += 256) inlier[k] = 0;
    }
    if (tid == 0) { s3_store(sh.S, s12 + 8 * (size_t)pb); n_inliers[pb] = n_in; }
}

}  // namespace

struct lpslam_hip_sim3 {
    lpslam_hip_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    Sim3View view{};
    int nb = 0, trial_blocks = 0;
    BaCtl h_ctl{};
    BaView* d_cv = nullptr;                 // device copy of view.cv: the Cholesky kernels read their view from device memory
    double* d_chi = nullptr;
    std::vector<void*> allocs;
};

namespace {

template <class T>
int s3_alloc(lpslam_hip_sim3* g, T** p, size_t n, bool zero = false)
{
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { set_error("hipMalloc failed (%zu bytes)", n * sizeof(T)); return LPSLAM_HIP_ERR_DEVICE; }
    g->allocs.push_back(d);
    if (zero && hipMemset(d, 0, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { set_error("hipMemset failed"); return LPSLAM_HIP_ERR_DEVICE; }
    *p = (T*)d;
    return LPSLAM_HIP_OK;
}
template <class T>
int s3_upload(lpslam_hip_sim3* g, const T** p, const std::vector<T>& h)
{
    T* d = nullptr;
    int rc = s3_alloc(g, &d, h.size());
    if (rc) return rc;
    if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); return LPSLAM_HIP_ERR_DEVICE; }
    *p = d;
    return LPSLAM_HIP_OK;
}

}  // namespace

extern "C" {

void lpslam_hip_sim3_destroy(lpslam_hip_sim3* g)
{
    if (!g) return;
    if (g->stream) { (void)hipStreamSynchronize(g->stream); (void)hipStreamDestroy(g->stream); }
    for (void* p : g->allocs) (void)hipFree(p);
    delete g;
}

int lpslam_hip_sim3_create(lpslam_hip_ctx* ctx, const double* verts, const uint8_t* fixed, int32_t n,
                           const lpslam_hip_sim3_edge* edges, int32_t n_edges, int32_t fix_scale, lpslam_hip_sim3** out)
{
    if (!ctx || !verts || !out || n < 1 || n_edges < 0 || (n_edges > 0 && !edges)) { set_error("invalid pose-graph arguments"); return LPSLAM_HIP_ERR_INVALID; }
    *out = nullptr;
    for (int k = 0; k < n_edges; ++k)
        if (edges[k].i < 0 || edges[k].i >= n || edges[k].j < 0 || edges[k].j >= n || edges[k].i == edges[k].j) {
            set_error("edge %d joins vertices %d / %d (of %d)", k, edges[k].i, edges[k].j, n); return LPSLAM_HIP_ERR_INVALID;
        }
    LP_HIP(hipSetDevice(ctx->cfg.device));
    lpslam_hip_sim3* g = new lpslam_hip_sim3();
    g->ctx = ctx;
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) { delete g; set_error("hipStreamCreate failed"); return LPSLAM_HIP_ERR_DEVICE; }
    Sim3View& v = g->view;
    std::vector<int> slot(n), free_vert;
    for (int i = 0; i < n; ++i) { if (fixed && fixed[i]) slot[i] = -1; else { slot[i] = (int)free_vert.size(); free_vert.push_back(i); } }
    v.n = n; v.n_free = (int)free_vert.size(); v.n_edges = n_edges; v.fix_scale = fix_scale ? 1 : 0;
    v.dim = 7 * v.n_free;
    v.dim_pad = ((v.dim + 1 + NB - 1) / NB) * NB;
    v.n_blocks = v.n_free * (v.n_free + 1) / 2;
    g->nb = v.dim_pad / NB;
    // structure: incidence lists per free vertex and edge lists per off-diagonal block pair, both in edge order
    std::vector<int> e_i(n_edges), e_j(n_edges);
    std::vector<double> meas(8 * (size_t)n_edges);
    std::vector<int> vt_start(v.n_free + 1, 0), blk_start((size_t)v.n_blocks + 1, 0);
    auto blk_index = [&](int i, int k) { return i * v.n_free - i * (i - 1) / 2 + (k - i); };
    for (int k = 0; k < n_edges; ++k) {
        e_i[k] = edges[k].i; e_j[k] = edges[k].j;
        for (int q = 0; q < 8; ++q) meas[8 * (size_t)k + q] = edges[k].meas[q];
        const int si = slot[e_i[k]], sj = slot[e_j[k]];
        if (si >= 0) vt_start[si + 1]++;
        if (sj >= 0) vt_start[sj + 1]++;
        if (si >= 0 && sj >= 0) blk_start[blk_index(std::min(si, sj), std::max(si, sj)) + 1]++;
    }
    for (int f = 0; f < v.n_free; ++f) vt_start[f + 1] += vt_start[f];
    for (int q = 0; q < v.n_blocks; ++q) blk_start[q + 1] += blk_start[q];
    std::vector<int2> vt_inc((size_t)vt_start[v.n_free]), blk_terms((size_t)blk_start[v.n_blocks]), blk_ik((size_t)v.n_blocks);
    {
        std::vector<int> fv(vt_start.begin(), vt_start.end() - 1), fb(blk_start.begin(), blk_start.end() - 1);
        for (int k = 0; k < n_edges; ++k) {
            const int si = slot[e_i[k]], sj = slot[e_j[k]];
            if (si >= 0) vt_inc[fv[si]++] = make_int2(k, 0);
            if (sj >= 0) vt_inc[fv[sj]++] = make_int2(k, 1);
            if (si >= 0 && sj >= 0) blk_terms[fb[blk_index(std::min(si, sj), std::max(si, sj))]++] = make_int2(k, si > sj ? 1 : 0);
        }
        for (int i = 0; i < v.n_free; ++i) for (int k = i; k < v.n_free; ++k) blk_ik[blk_index(i, k)] = make_int2(i, k);
    }
    int rc = 0;
    auto fail = [&](int code) { lpslam_hip_sim3_destroy(g); return code; };
#define S3_TRY(x) do { rc = (x); if (rc) return fail(rc); } while (0)
    S3_TRY(s3_upload(g, &v.slot, slot)); S3_TRY(s3_upload(g, &v.free_vert, free_vert));
    S3_TRY(s3_upload(g, &v.e_i, e_i)); S3_TRY(s3_upload(g, &v.e_j, e_j)); S3_TRY(s3_upload(g, &v.meas, meas));
    S3_TRY(s3_upload(g, &v.vt_start, vt_start)); S3_TRY(s3_upload(g, &v.vt_inc, vt_inc));
    S3_TRY(s3_upload(g, &v.blk_start, blk_start)); S3_TRY(s3_upload(g, &v.blk_terms, blk_terms)); S3_TRY(s3_upload(g, &v.blk_ik, blk_ik));
    for (int s = 0; s < 2; ++s) S3_TRY(s3_alloc(g, &v.verts_buf[s], 8 * (size_t)n));
    if (hipMemcpy(v.verts_buf[0], verts, 8 * (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); return fail(LPSLAM_HIP_ERR_DEVICE); }
    S3_TRY(s3_alloc(g, &v.eblk, (size_t)n_edges * S3_EB, true));
    S3_TRY(s3_alloc(g, &v.bp, (size_t)v.dim_pad, true));
    g->trial_blocks = std::max((n_edges + 255) / 256, 1);
    S3_TRY(s3_alloc(g, &v.part, (size_t)g->trial_blocks, true));
    S3_TRY(s3_alloc(g, &g->d_chi, (size_t)n_edges));
    BaView& cv = v.cv;
    cv = BaView{};
    cv.band_hbw = -1;                     // the dense factorisation (a zero here would read as "banded": ba_band.inl)
    cv.dim = v.dim; cv.dim_pad = v.dim_pad; cv.n_points = 0; cv.n_poses = n; cv.n_free = v.n_free;
    S3_TRY(s3_alloc(g, &cv.S, (size_t)v.dim_pad * v.dim_pad, true));
    {   // rows beyond the rhs row: identity (they stay 1 / 0 through every factorisation)
        const double one = 1.0;
        for (int r = v.dim + 1; r < v.dim_pad; ++r)
            if (hipMemcpy(cv.S + (size_t)r * v.dim_pad + r, &one, sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); return fail(LPSLAM_HIP_ERR_DEVICE); }
    }
    S3_TRY(s3_alloc(g, &cv.Minv, (size_t)v.dim_pad * v.dim_pad, true));
    S3_TRY(s3_alloc(g, &cv.Ldiag, (size_t)v.dim_pad * NB, true));
    S3_TRY(s3_alloc(g, &cv.Lsub, (size_t)v.dim_pad * NB, true));
    S3_TRY(s3_alloc(g, &cv.xp, (size_t)v.dim_pad, true));
    S3_TRY(s3_alloc(g, &cv.scal, 8, true));
    S3_TRY(s3_alloc(g, &cv.ctl, 1, true));
    S3_TRY(s3_alloc(g, &cv.log, MAX_LOG, true));
    S3_TRY(s3_alloc(g, &g->d_cv, 1));
    if (hipMemcpy(g->d_cv, &cv, sizeof(BaView), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); return fail(LPSLAM_HIP_ERR_DEVICE); }
#undef S3_TRY
    *out = g;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_sim3_optimize(lpslam_hip_sim3* g, int32_t iters, lpslam_hip_ba_iter_log* log, int32_t* done_out)
{
    if (!g) { set_error("null pose graph"); return LPSLAM_HIP_ERR_INVALID; }
    if (iters < 0 || iters > MAX_LOG) { set_error("iterations must be in [0,%d]", MAX_LOG); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(g->ctx->cfg.device));
    Sim3View& v = g->view;
    hipStream_t s = g->stream;
    BaCtl c = g->h_ctl;                     // keeps `cur` (the accepted state) from earlier calls; lambda_0 is recomputed per call
    c.max_outer = iters; c.outer_done = 0; c.need_lin = 1; c.first = 1; c.qmax = 0; c.stopped = 0; c.ni = 2; c.rho = 0; c.last_accepted = 0; c.ticket = 0;
    LP_HIP(hipMemcpyAsync(v.cv.ctl, &c, sizeof(BaCtl), hipMemcpyHostToDevice, s));
    LP_HIP(hipStreamSynchronize(s));
    g->h_ctl = c;
    const int vb = (v.n + 255) / 256;
    int guard = 0;
    while (!g->h_ctl.stopped && g->h_ctl.outer_done < iters && guard++ < 16 * MAX_LOG) {
        const int units = iters - g->h_ctl.outer_done;
        for (int u = 0; u < units; ++u) {
            if (v.n_edges) hipLaunchKernelGGL(k_sim3_lin, dim3(v.n_edges), dim3(64), 0, s, v);
            if (v.dim > 0) {
                hipLaunchKernelGGL(k_sim3_assemble, dim3(v.n_blocks + v.n_free), dim3(64), 0, s, v);
                enqueue_factor_solve(s, g->d_cv, 1, g->nb, v.dim, false, cw_fits(v.dim), !cw_fits(v.dim));
            }
            hipLaunchKernelGGL(k_sim3_update, dim3(vb + 1), dim3(256), 0, s, v, vb);
            hipLaunchKernelGGL(k_sim3_trial, dim3(g->trial_blocks), dim3(256), 0, s, v);
        }
        LP_HIP(hipGetLastError());
        LP_HIP(hipMemcpyAsync(&g->h_ctl, v.cv.ctl, sizeof(BaCtl), hipMemcpyDeviceToHost, s));
        LP_HIP(hipStreamSynchronize(s));
    }
    const int done = g->h_ctl.outer_done;
    if (log && done) LP_HIP(hipMemcpy(log, v.cv.log, std::min(done, MAX_LOG) * sizeof(lpslam_hip_ba_iter_log), hipMemcpyDeviceToHost));
    if (done_out) *done_out = done;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_sim3_get(lpslam_hip_sim3* g, double* verts)
{
    if (!g || !verts) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipMemcpyAsync(verts, g->view.verts_buf[g->h_ctl.cur], 8 * (size_t)g->view.n * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    LP_HIP(hipStreamSynchronize(g->stream));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_sim3_chi2(lpslam_hip_sim3* g, double* chi2)
{
    if (!g || !chi2) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    if (!g->view.n_edges) return LPSLAM_HIP_OK;
    hipLaunchKernelGGL(k_sim3_chi2, dim3((g->view.n_edges + 255) / 256), dim3(256), 0, g->stream, g->view, g->d_chi);
    LP_HIP(hipGetLastError());
    LP_HIP(hipMemcpyAsync(chi2, g->d_chi, (size_t)g->view.n_edges * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    LP_HIP(hipStreamSynchronize(g->stream));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_sim3_transform_optimize(lpslam_hip_ctx* ctx, int32_t n_problems, double* s12, const lpslam_hip_sim3_pair* pairs,
                                       const int32_t* pair_start, const double* cam1, const double* cam2, double chi_sq,
                                       int32_t fix_scale, uint8_t* inlier, int32_t* n_inliers)
{
    if (!ctx || n_problems < 0 || (n_problems > 0 && (!s12 || !pair_start || !cam1 || !cam2 || !n_inliers)) || !(chi_sq > 0)) {
        set_error("invalid Sim3 transform arguments"); return LPSLAM_HIP_ERR_INVALID;
    }
    if (n_problems == 0) return LPSLAM_HIP_OK;
    for (int i = 0; i < n_problems; ++i) if (pair_start[i + 1] < pair_start[i] || pair_start[0] != 0) { set_error("pair_start must be a non-decreasing prefix sum from 0"); return LPSLAM_HIP_ERR_INVALID; }
    const size_t total = (size_t)pair_start[n_problems];
    if (total > 0 && !pairs) { set_error("pairs is NULL"); return LPSLAM_HIP_ERR_INVALID; }
    static_assert(sizeof(lpslam_hip_sim3_pair) == sizeof(S3Pair), "pair layout");
    LP_HIP(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->stream;
    double* d_s = nullptr; S3Pair* d_p = nullptr; int* d_start = nullptr; double* d_cam = nullptr; uint8_t* d_act = nullptr; uint8_t* d_in = nullptr; int* d_n = nullptr;
    auto release = [&]() { for (void* p : {(void*)d_s, (void*)d_p, (void*)d_start, (void*)d_cam, (void*)d_act, (void*)d_in, (void*)d_n}) if (p) (void)hipFree(p); };
#define T_HIP(x) do { if ((x) != hipSuccess) { release(); set_error("HIP call failed: %s", #x); return LPSLAM_HIP_ERR_DEVICE; } } while (0)
    T_HIP(hipMalloc((void**)&d_s, 8 * (size_t)n_problems * sizeof(double)));
    T_HIP(hipMalloc((void**)&d_p, std::max<size_t>(total, 1) * sizeof(S3Pair)));
    T_HIP(hipMalloc((void**)&d_start, ((size_t)n_problems + 1) * sizeof(int)));
    T_HIP(hipMalloc((void**)&d_cam, 8 * sizeof(double)));
    T_HIP(hipMalloc((void**)&d_act, std::max<size_t>(total, 1)));
    T_HIP(hipMalloc((void**)&d_in, std::max<size_t>(total, 1)));
    T_HIP(hipMalloc((void**)&d_n, (size_t)n_problems * sizeof(int)));
    T_HIP(hipMemcpyAsync(d_s, s12, 8 * (size_t)n_problems * sizeof(double), hipMemcpyHostToDevice, s));
    if (total) T_HIP(hipMemcpyAsync(d_p, pairs, total * sizeof(S3Pair), hipMemcpyHostToDevice, s));
    T_HIP(hipMemcpyAsync(d_start, pair_start, ((size_t)n_problems + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    T_HIP(hipMemcpyAsync(d_cam, cam1, 4 * sizeof(double), hipMemcpyHostToDevice, s));
    T_HIP(hipMemcpyAsync(d_cam + 4, cam2, 4 * sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sim3_transform, dim3(n_problems), dim3(256), 0, s, d_s, d_p, d_start, d_cam, chi_sq, fix_scale ? 1 : 0, d_act, d_in, d_n);
    T_HIP(hipGetLastError());
    T_HIP(hipMemcpyAsync(s12, d_s, 8 * (size_t)n_problems * sizeof(double), hipMemcpyDeviceToHost, s));
    T_HIP(hipMemcpyAsync(n_inliers, d_n, (size_t)n_problems * sizeof(int), hipMemcpyDeviceToHost, s));
    if (inlier && total) T_HIP(hipMemcpyAsync(inlier, d_in, total, hipMemcpyDeviceToHost, s));
    T_HIP(hipStreamSynchronize(s));
#undef T_HIP
    release();
    return LPSLAM_HIP_OK;
}

}  // extern "C"
