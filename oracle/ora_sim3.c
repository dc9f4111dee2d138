/*
 * oracle/ora_sim3.c -- CPU restatement of the Sim3 pose-graph optimisation.  TEST INFRASTRUCTURE ONLY (see ora.h).
 * PARITY UNPINNED (see ora.h).  Restates, in plain C / FP64:
 *   [UPSTREAM] g2o@691dc51 types/sim3: Sim3 (exp constructor, log, inverse, operator*), VertexSim3Expmap::oplusImpl
 *              (estimate <- Sim3(update) * estimate, update[6] = 0 with _fix_scale), EdgeSim3::computeError
 *              (error = log(measurement * v0 * v1^-1)), BaseBinaryEdge numeric linearizeOplus (central differences,
 *              delta = 1e-9), OptimizationAlgorithmLevenberg, BlockSolver_7_3 without marginalisation
 *              (pin: conan-packages/g2o-conan/conanfile.py:6,24-27)
 *   [UPSTREAM] OpenVSLAM optimize::graph_optimizer (identity information, 50 LM iterations, loop keyframe fixed,
 *              scale fixed for stereo) -- SURVEY.md section 8(a) row a23; driven from the reference through
 *              openvslam::system (src/Trackers/OpenVSLAMTrackerBase.cpp:238-255: loop detector enabled / disabled).
 * Choices where the pinned sources cannot be consulted: quaternion products are not re-normalised (the classic Sim3
 * operator*); the small-angle / non-zero-sigma coefficient is B = ((sigma^2/2 - sigma + 1) s - 1) / sigma^3, the limit of the
 * general-branch expression (older g2o copies print it without the "- 1", which is off by ~1/sigma^3 and makes the
 * free-scale error meaningless inside 0.26 degrees; if the pin turns out to carry that form, this line is the one to change).
 * A Sim3 is 8 doubles: qw qx qy qz tx ty tz s (maps world -> camera: x_c = s R x_w + t).
 */
#include "ora.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

typedef struct { double q[4]; double t[3]; double s; } sim3;

static void q_to_R(const double* q, double* R)       /* Eigen toRotationMatrix (no normalisation) */
{
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
static void R_to_q(const double* R, double* q)       /* Eigen Quaternion(Matrix3) */
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        q[1 + i] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}
static void q_mul(const double* a, const double* b, double* o)
{
    const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    const double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static void q_rot(const double* q, const double* v, double* o)   /* Eigen _transformVector: v + w uv + qv x uv, uv = 2 qv x v */
{
    const double ux = 2 * (q[2] * v[2] - q[3] * v[1]), uy = 2 * (q[3] * v[0] - q[1] * v[2]), uz = 2 * (q[1] * v[1] - q[2] * v[0]);
    o[0] = v[0] + q[0] * ux + (q[2] * uz - q[3] * uy);
    o[1] = v[1] + q[0] * uy + (q[3] * ux - q[1] * uz);
    o[2] = v[2] + q[0] * uz + (q[1] * uy - q[2] * ux);
}
static void mat3_mul(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
static void skew(const double* w, double* O)
{
    O[0] = 0; O[1] = -w[2]; O[2] = w[1]; O[3] = w[2]; O[4] = 0; O[5] = -w[0]; O[6] = -w[1]; O[7] = w[0]; O[8] = 0;
}

/* Sim3(const Vector7& update): omega = update[0..2], upsilon = update[3..5], sigma = update[6] */
static void sim3_exp(const double* u, sim3* o)
{
    const double* omega = u; const double* ups = u + 3; const double sigma = u[6];
    const double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
    double Om[9], Om2[9], R[9];
    skew(omega, Om);
    mat3_mul(Om, Om, Om2);
    o->s = exp(sigma);
    const double eps = 0.00001;
    double A, B, C;
    if (fabs(sigma) < eps) {
        C = 1;
        if (theta < eps) {
            A = 0.5; B = 1. / 6.;
            for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i];
        } else {
            const double theta2 = theta * theta;
            A = (1 - cos(theta)) / theta2;
            B = (theta - sin(theta)) / (theta2 * theta);
            const double a1 = sin(theta) / theta, a2 = (1 - cos(theta)) / (theta * theta);
            for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a1 * Om[i] + a2 * Om2[i];
        }
    } else {
        C = (o->s - 1) / sigma;
        if (theta < eps) {
            const double sigma2 = sigma * sigma;
            A = ((sigma - 1) * o->s + 1) / sigma2;
            B = ((0.5 * sigma2 - sigma + 1) * o->s - 1) / (sigma2 * sigma);
            for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i];
        } else {
            const double a1 = sin(theta) / theta, a2 = (1 - cos(theta)) / (theta * theta);
            for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a1 * Om[i] + a2 * Om2[i];
            const double a = o->s * sin(theta), b = o->s * cos(theta);
            const double theta2 = theta * theta, sigma2 = sigma * sigma, c = theta2 + sigma2;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1 / theta2;
        }
    }
    R_to_q(R, o->q);
    for (int i = 0; i < 3; ++i) {
        double acc = 0;
        for (int j = 0; j < 3; ++j) acc += (A * Om[i * 3 + j] + B * Om2[i * 3 + j] + (i == j ? C : 0.0)) * ups[j];
        o->t[i] = acc;
    }
}

/* x = W^-1 t by LU with partial pivoting (Eigen PartialPivLU) */
static void lu_solve3(const double* Win, const double* t, double* x)
{
    double W[9], b[3] = {t[0], t[1], t[2]};
    memcpy(W, Win, sizeof(W));
    for (int k = 0; k < 3; ++k) {
        int piv = k;
        for (int r = k + 1; r < 3; ++r) if (fabs(W[r * 3 + k]) > fabs(W[piv * 3 + k])) piv = r;
        if (piv != k) {
            for (int c = 0; c < 3; ++c) { const double tmp = W[k * 3 + c]; W[k * 3 + c] = W[piv * 3 + c]; W[piv * 3 + c] = tmp; }
            const double tb = b[k]; b[k] = b[piv]; b[piv] = tb;
        }
        for (int r = k + 1; r < 3; ++r) {
            const double f = W[r * 3 + k] / W[k * 3 + k];
            W[r * 3 + k] = f;
            for (int c = k + 1; c < 3; ++c) W[r * 3 + c] -= f * W[k * 3 + c];
        }
    }
    for (int r = 1; r < 3; ++r) for (int c = 0; c < r; ++c) b[r] -= W[r * 3 + c] * b[c];
    for (int r = 2; r >= 0; --r) { for (int c = r + 1; c < 3; ++c) b[r] -= W[r * 3 + c] * x[c]; x[r] = b[r] / W[r * 3 + r]; }
}

static void sim3_log(const sim3* S, double* res)
{
    const double sigma = log(S->s);
    double R[9], omega[3], Om[9], Om2[9];
    q_to_R(S->q, R);
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double eps = 0.00001;
    double A, B, C;
    if (fabs(sigma) < eps) {
        C = 1;
        if (d > 1 - eps) {
            for (int i = 0; i < 3; ++i) omega[i] = 0.5 * dR[i];
            A = 0.5; B = 1. / 6.;
        } else {
            const double theta = acos(d), theta2 = theta * theta;
            const double f = theta / (2 * sqrt(1 - d * d));
            for (int i = 0; i < 3; ++i) omega[i] = f * dR[i];
            A = (1 - cos(theta)) / theta2;
            B = (theta - sin(theta)) / (theta2 * theta);
        }
    } else {
        C = (S->s - 1) / sigma;
        if (d > 1 - eps) {
            const double sigma2 = sigma * sigma;
            for (int i = 0; i < 3; ++i) omega[i] = 0.5 * dR[i];
            A = ((sigma - 1) * S->s + 1) / sigma2;
            B = ((0.5 * sigma2 - sigma + 1) * S->s - 1) / (sigma2 * sigma);
        } else {
            const double theta = acos(d);
            const double f = theta / (2 * sqrt(1 - d * d));
            for (int i = 0; i < 3; ++i) omega[i] = f * dR[i];
            const double theta2 = theta * theta;
            const double a = S->s * sin(theta), b = S->s * cos(theta), c = theta2 + sigma * sigma;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1 / theta2;
        }
    }
    skew(omega, Om);
    mat3_mul(Om, Om, Om2);
    double W[9];
    for (int i = 0; i < 9; ++i) W[i] = A * Om[i] + B * Om2[i] + (i % 4 == 0 ? C : 0.0);
    lu_solve3(W, S->t, res + 3);
    for (int i = 0; i < 3; ++i) res[i] = omega[i];
    res[6] = sigma;
}

static void sim3_mul(const sim3* a, const sim3* b, sim3* o)     /* o = a * b */
{
    sim3 r;
    q_mul(a->q, b->q, r.q);
    double rt[3];
    q_rot(a->q, b->t, rt);
    for (int i = 0; i < 3; ++i) r.t[i] = a->s * rt[i] + a->t[i];
    r.s = a->s * b->s;
    *o = r;
}
static void sim3_inv(const sim3* a, sim3* o)
{
    sim3 r;
    r.q[0] = a->q[0]; r.q[1] = -a->q[1]; r.q[2] = -a->q[2]; r.q[3] = -a->q[3];
    const double v[3] = {(-1. / a->s) * a->t[0], (-1. / a->s) * a->t[1], (-1. / a->s) * a->t[2]};
    q_rot(r.q, v, r.t);
    r.s = 1. / a->s;
    *o = r;
}
static void load(const double* p, sim3* s) { memcpy(s->q, p, 4 * sizeof(double)); memcpy(s->t, p + 4, 3 * sizeof(double)); s->s = p[7]; }
static void store(const sim3* s, double* p) { memcpy(p, s->q, 4 * sizeof(double)); memcpy(p + 4, s->t, 3 * sizeof(double)); p[7] = s->s; }

void ora_sim3_exp(const double* update7, double* sim3_out) { sim3 s; sim3_exp(update7, &s); store(&s, sim3_out); }
void ora_sim3_log(const double* sim3_in, double* log7) { sim3 s; load(sim3_in, &s); sim3_log(&s, log7); }
void ora_sim3_mul(const double* a, const double* b, double* out) { sim3 x, y, z; load(a, &x); load(b, &y); sim3_mul(&x, &y, &z); store(&z, out); }
void ora_sim3_inv(const double* a, double* out) { sim3 x, z; load(a, &x); sim3_inv(&x, &z); store(&z, out); }

/* EdgeSim3::computeError */
static void edge_error(const sim3* meas, const sim3* vi, const sim3* vj, double* e)
{
    sim3 inv, t1, t2;
    sim3_inv(vj, &inv);
    sim3_mul(meas, vi, &t1);
    sim3_mul(&t1, &inv, &t2);
    sim3_log(&t2, e);
}
/* VertexSim3Expmap::oplusImpl */
static void vertex_oplus(const sim3* est, const double* update, int fix_scale, sim3* out)
{
    double u[7];
    memcpy(u, update, sizeof(u));
    if (fix_scale) u[6] = 0;
    sim3 d;
    sim3_exp(u, &d);
    sim3_mul(&d, est, out);
}

double ora_sim3_graph_chi2(const double* verts, const ora_sim3_edge* edges, int n_edges)
{
    double chi = 0;
    for (int k = 0; k < n_edges; ++k) {
        sim3 m, a, b; double e[7];
        load(edges[k].meas, &m); load(verts + 8 * edges[k].i, &a); load(verts + 8 * edges[k].j, &b);
        edge_error(&m, &a, &b, e);
        double c = 0;
        for (int r = 0; r < 7; ++r) c += e[r] * e[r];
        chi += c;
    }
    return chi;
}

static int chol_factor(double* A, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    return 1;
}
static void chol_solve(const double* L, int n, double* b)
{
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
}

/* numeric Jacobian of one edge w.r.t. one of its vertices (BaseBinaryEdge::linearizeOplus, central differences) */
static void numeric_jacobian(const sim3* meas, const sim3* vi, const sim3* vj, int which, int fix_scale, double* J /* 7x7 row-major */)
{
    const double delta = 1e-9, scalar = 1.0 / (2 * delta);
    for (int d = 0; d < 7; ++d) {
        double add[7] = {0, 0, 0, 0, 0, 0, 0}, e1[7], e2[7];
        sim3 pert;
        add[d] = delta;
        vertex_oplus(which == 0 ? vi : vj, add, fix_scale, &pert);
        edge_error(meas, which == 0 ? &pert : vi, which == 0 ? vj : &pert, e1);
        add[d] = -delta;
        vertex_oplus(which == 0 ? vi : vj, add, fix_scale, &pert);
        edge_error(meas, which == 0 ? &pert : vi, which == 0 ? vj : &pert, e2);
        for (int r = 0; r < 7; ++r) J[r * 7 + d] = scalar * (e1[r] - e2[r]);
    }
}

int ora_sim3_graph_optimize(double* verts, const uint8_t* fixed, int n, const ora_sim3_edge* edges, int n_edges,
                            int fix_scale, int iters, ora_ba_iter_log* log)
{
    int* slot = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int n_free = 0;
    for (int i = 0; i < n; ++i) slot[i] = (fixed && fixed[i]) ? -1 : n_free++;
    const int dim = 7 * n_free;
    const size_t dd = (size_t)(dim > 0 ? dim : 1);
    double* H = (double*)malloc(sizeof(double) * dd * dd);
    double* A = (double*)malloc(sizeof(double) * dd * dd);
    double* b = (double*)malloc(sizeof(double) * dd);
    double* x = (double*)malloc(sizeof(double) * dd);
    double* bak = (double*)malloc(sizeof(double) * 8 * (size_t)(n > 0 ? n : 1));
    double lambda = 0, ni = 2;
    int it = 0;
    for (; it < iters; ++it) {
        double current_chi = ora_sim3_graph_chi2(verts, edges, n_edges);
        double temp_chi = current_chi;
        /* buildSystem: H = sum J^T J, b = -sum J^T e (information = identity) */
        memset(H, 0, sizeof(double) * dd * dd);
        memset(b, 0, sizeof(double) * dd);
        for (int k = 0; k < n_edges; ++k) {
            sim3 m, vi, vj; double e[7], Ji[49], Jj[49];
            load(edges[k].meas, &m); load(verts + 8 * edges[k].i, &vi); load(verts + 8 * edges[k].j, &vj);
            const int si = slot[edges[k].i], sj = slot[edges[k].j];
            edge_error(&m, &vi, &vj, e);
            if (si >= 0) numeric_jacobian(&m, &vi, &vj, 0, fix_scale, Ji);
            if (sj >= 0) numeric_jacobian(&m, &vi, &vj, 1, fix_scale, Jj);
            for (int a = 0; a < 7; ++a) {
                if (si >= 0) {
                    double s = 0;
                    for (int r = 0; r < 7; ++r) s += Ji[r * 7 + a] * e[r];
                    b[7 * si + a] -= s;
                    for (int c = 0; c < 7; ++c) {
                        double h = 0;
                        for (int r = 0; r < 7; ++r) h += Ji[r * 7 + a] * Ji[r * 7 + c];
                        H[(size_t)(7 * si + a) * dim + 7 * si + c] += h;
                    }
                }
                if (sj >= 0) {
                    double s = 0;
                    for (int r = 0; r < 7; ++r) s += Jj[r * 7 + a] * e[r];
                    b[7 * sj + a] -= s;
                    for (int c = 0; c < 7; ++c) {
                        double h = 0;
                        for (int r = 0; r < 7; ++r) h += Jj[r * 7 + a] * Jj[r * 7 + c];
                        H[(size_t)(7 * sj + a) * dim + 7 * sj + c] += h;
                    }
                }
                if (si >= 0 && sj >= 0 && si != sj) {
                    for (int c = 0; c < 7; ++c) {
                        double h = 0;
                        for (int r = 0; r < 7; ++r) h += Ji[r * 7 + a] * Jj[r * 7 + c];
                        H[(size_t)(7 * si + a) * dim + 7 * sj + c] += h;
                        H[(size_t)(7 * sj + c) * dim + 7 * si + a] += h;
                    }
                }
            }
        }
        if (it == 0) {
            double maxd = 0;
            for (int j = 0; j < dim; ++j) { const double v = fabs(H[(size_t)j * dim + j]); if (v > maxd) maxd = v; }
            lambda = 1e-5 * maxd;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        const double chi_before = current_chi;
        do {
            memcpy(bak, verts, sizeof(double) * 8 * (size_t)n);
            memcpy(A, H, sizeof(double) * dd * dd);
            for (int j = 0; j < dim; ++j) A[(size_t)j * dim + j] += lambda;
            int ok = dim == 0 ? 1 : chol_factor(A, dim);
            if (ok) {
                memcpy(x, b, sizeof(double) * dd);
                if (dim) chol_solve(A, dim, x);
                for (int i = 0; i < n; ++i) if (slot[i] >= 0) {
                    sim3 est, upd;
                    load(verts + 8 * i, &est);
                    vertex_oplus(&est, x + 7 * slot[i], fix_scale, &upd);
                    store(&upd, verts + 8 * i);
                }
            }
            temp_chi = ora_sim3_graph_chi2(verts, edges, n_edges);
            if (!ok) temp_chi = DBL_MAX;
            rho = current_chi - temp_chi;
            double scale = 0;
            if (ok) for (int j = 0; j < dim; ++j) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(temp_chi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                const double sf = alpha > 1. / 3. ? alpha : 1. / 3.;
                lambda *= sf;
                ni = 2;
                current_chi = temp_chi;
            } else {
                lambda *= ni;
                ni *= 2;
                memcpy(verts, bak, sizeof(double) * 8 * (size_t)n);
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        const int terminate = (qmax == 10 || rho == 0);
        if (log) {
            log[it].chi2_before = chi_before; log[it].chi2_after = current_chi;
            log[it].lambda = lambda; log[it].trials = qmax; log[it].status = terminate;
        }
        if (terminate) { ++it; break; }
    }
    free(slot); free(H); free(A); free(b); free(x); free(bak);
    return it;
}

/* ---- Sim3 between two keyframes ([UPSTREAM] OpenVSLAM optimize::transform_optimizer; ORB-SLAM2 Optimizer::OptimizeSim3) ----
 * One Sim3 vertex S12 (camera 2 -> camera 1); per matched landmark pair two reprojection edges with identity-scaled
 * information and Huber kernel sqrt(chi_sq): forward  e = obs1 - proj1(S12 * P2c), backward e = obs2 - proj2(S12^-1 * P1c)
 * (g2o EdgeSim3ProjectXYZ / EdgeInverseSim3ProjectXYZ: numeric Jacobians).  Flow: 5 Levenberg iterations, pairs with
 * chi2 > chi_sq on either edge are dropped, 10 more iterations if any was dropped (else 5), 0 is returned when fewer than
 * 10 pairs survive the first cut; inlier[k] = both edges within chi_sq at the end. */
static void t_error(const sim3* S, const sim3* Sinv, const ora_sim3_pair* p, const double* cam1, const double* cam2, double* e)
{
    double x[3];
    sim3 tmp = *S;
    double r[3];
    q_rot(tmp.q, p->p2c, r);
    for (int i = 0; i < 3; ++i) x[i] = tmp.s * r[i] + tmp.t[i];
    e[0] = p->obs1[0] - (cam1[0] * x[0] / x[2] + cam1[2]);
    e[1] = p->obs1[1] - (cam1[1] * x[1] / x[2] + cam1[3]);
    q_rot(Sinv->q, p->p1c, r);
    for (int i = 0; i < 3; ++i) x[i] = Sinv->s * r[i] + Sinv->t[i];
    e[2] = p->obs2[0] - (cam2[0] * x[0] / x[2] + cam2[2]);
    e[3] = p->obs2[1] - (cam2[1] * x[1] / x[2] + cam2[3]);
}
static void huber2(double e2, double delta, double* rho0, double* rho1)
{
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; }
    else { const double sq = sqrt(e2); *rho0 = 2 * sq * delta - dsqr; *rho1 = delta / sq; }
}
static double t_chi2(const sim3* S, const ora_sim3_pair* pairs, const uint8_t* active, int n, const double* cam1, const double* cam2, double delta)
{
    sim3 Si; sim3_inv(S, &Si);
    double chi = 0;
    for (int k = 0; k < n; ++k) {
        if (!active[k]) continue;
        double e[4], r0, r1;
        t_error(S, &Si, &pairs[k], cam1, cam2, e);
        huber2(pairs[k].inv_sigma2_1 * (e[0] * e[0] + e[1] * e[1]), delta, &r0, &r1); chi += r0;
        huber2(pairs[k].inv_sigma2_2 * (e[2] * e[2] + e[3] * e[3]), delta, &r0, &r1); chi += r0;
    }
    return chi;
}
static int t_levenberg(sim3* S, const ora_sim3_pair* pairs, const uint8_t* active, int n, const double* cam1, const double* cam2,
                       double delta, int fix_scale, int iters)
{
    double lambda = 0, ni = 2;
    int it = 0;
    for (; it < iters; ++it) {
        double current_chi = t_chi2(S, pairs, active, n, cam1, cam2, delta), temp_chi;
        double H[49], b[7];
        memset(H, 0, sizeof(H)); memset(b, 0, sizeof(b));
        sim3 Sp[14], Spi[14], Si;
        sim3_inv(S, &Si);
        for (int d = 0; d < 7; ++d)
            for (int sgn = 0; sgn < 2; ++sgn) {
                double add[7] = {0, 0, 0, 0, 0, 0, 0};
                add[d] = sgn ? -1e-9 : 1e-9;
                vertex_oplus(S, add, fix_scale, &Sp[2 * d + sgn]);
                sim3_inv(&Sp[2 * d + sgn], &Spi[2 * d + sgn]);
            }
        for (int k = 0; k < n; ++k) {
            if (!active[k]) continue;
            double e[4], J[4][7];
            t_error(S, &Si, &pairs[k], cam1, cam2, e);
            for (int d = 0; d < 7; ++d) {
                double e1[4], e2[4];
                t_error(&Sp[2 * d], &Spi[2 * d], &pairs[k], cam1, cam2, e1);
                t_error(&Sp[2 * d + 1], &Spi[2 * d + 1], &pairs[k], cam1, cam2, e2);
                for (int r = 0; r < 4; ++r) J[r][d] = (1.0 / (2 * 1e-9)) * (e1[r] - e2[r]);
            }
            for (int half = 0; half < 2; ++half) {          /* the two edges of the pair, forward first */
                const double om = half ? pairs[k].inv_sigma2_2 : pairs[k].inv_sigma2_1;
                const double* eh = e + 2 * half;
                double r0, r1;
                huber2(om * (eh[0] * eh[0] + eh[1] * eh[1]), delta, &r0, &r1);
                const double w = om * r1;
                for (int a = 0; a < 7; ++a) {
                    b[a] -= J[2 * half][a] * w * eh[0] + J[2 * half + 1][a] * w * eh[1];
                    for (int c = 0; c < 7; ++c) H[a * 7 + c] += J[2 * half][a] * w * J[2 * half][c] + J[2 * half + 1][a] * w * J[2 * half + 1][c];
                }
            }
        }
        if (it == 0) {
            double maxd = 0;
            for (int j = 0; j < 7; ++j) if (fabs(H[j * 8]) > maxd) maxd = fabs(H[j * 8]);
            lambda = 1e-5 * maxd; ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            const sim3 bak = *S;
            double A[49], x[7];
            memcpy(A, H, sizeof(A));
            for (int j = 0; j < 7; ++j) A[j * 8] += lambda;
            const int ok = chol_factor(A, 7);
            if (ok) {
                memcpy(x, b, sizeof(x));
                chol_solve(A, 7, x);
                sim3 upd;
                vertex_oplus(S, x, fix_scale, &upd);
                *S = upd;
            }
            temp_chi = t_chi2(S, pairs, active, n, cam1, cam2, delta);
            if (!ok) temp_chi = DBL_MAX;
            rho = current_chi - temp_chi;
            double scale = 0;
            if (ok) for (int j = 0; j < 7; ++j) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(temp_chi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                lambda *= alpha > 1. / 3. ? alpha : 1. / 3.;
                ni = 2;
                current_chi = temp_chi;
            } else {
                lambda *= ni; ni *= 2;
                *S = bak;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) { ++it; break; }
    }
    return it;
}

int ora_sim3_transform_optimize(double* s12, const ora_sim3_pair* pairs, int n, const double* cam1, const double* cam2,
                                double chi_sq, int fix_scale, uint8_t* inlier)
{
    sim3 S; load(s12, &S);
    const double delta = sqrt(chi_sq);
    uint8_t* active = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    memset(active, 1, (size_t)(n > 0 ? n : 1));
    t_levenberg(&S, pairs, active, n, cam1, cam2, delta, fix_scale, 5);
    int n_bad = 0;
    {
        sim3 Si; sim3_inv(&S, &Si);
        for (int k = 0; k < n; ++k) {
            double e[4];
            t_error(&S, &Si, &pairs[k], cam1, cam2, e);
            if (pairs[k].inv_sigma2_1 * (e[0] * e[0] + e[1] * e[1]) > chi_sq || pairs[k].inv_sigma2_2 * (e[2] * e[2] + e[3] * e[3]) > chi_sq) { active[k] = 0; ++n_bad; }
        }
    }
    if (n - n_bad < 10) { if (inlier) memset(inlier, 0, (size_t)n); store(&S, s12); free(active); return 0; }
    t_levenberg(&S, pairs, active, n, cam1, cam2, delta, fix_scale, n_bad > 0 ? 10 : 5);
    int n_in = 0;
    {
        sim3 Si; sim3_inv(&S, &Si);
        for (int k = 0; k < n; ++k) {
            int in = 0;
            if (active[k]) {
                double e[4];
                t_error(&S, &Si, &pairs[k], cam1, cam2, e);
                in = !(pairs[k].inv_sigma2_1 * (e[0] * e[0] + e[1] * e[1]) > chi_sq || pairs[k].inv_sigma2_2 * (e[2] * e[2] + e[3] * e[3]) > chi_sq);
            }
            if (inlier) inlier[k] = (uint8_t)in;
            n_in += in;
        }
    }
    store(&S, s12);
    free(active);
    return n_in;
}
