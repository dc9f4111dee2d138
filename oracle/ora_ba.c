/*
 * oracle/ora_ba.c -- CPU restatement of the SE3 bundle adjustment.  TEST INFRASTRUCTURE ONLY (see ora.h).
 * PARITY UNPINNED (see ora.h).  Restates, in plain C / FP64:
 *   [UPSTREAM] g2o@691dc51 OptimizationAlgorithmLevenberg::solve / computeLambdaInit / computeScale,
 *              BlockSolver<6,3> buildSystem + Schur complement, BaseBinaryEdge::constructQuadraticForm,
 *              RobustKernelHuber::robustify, SE3Quat::exp  (pin: conan-packages/g2o-conan/conanfile.py:6,24-27)
 *   [UPSTREAM] OpenVSLAM optimize::local_bundle_adjuster, optimize::pose_optimizer and the
 *              se3::{mono,stereo}_perspective_reproj_edge Jacobians.
 * Camera parameters arrive from src/Trackers/OpenVSLAMTrackerBase.cpp:161-190 (fx, fy, cx, cy,
 * focal_x_baseline; semantics src/Interface/LpSlamTypes.h:219-222).
 */
#include "ora.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define CHI2_2D 5.99146
#define CHI2_3D 7.81473

static void quat_to_rot(const double* q, double* R)
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* pose <- exp([omega, upsilon]) * pose   (SE3Quat::exp, rotation first) */
static void pose_oplus(double* pose, const double* d)
{
    const double wx = d[0], wy = d[1], wz = d[2];
    const double theta2 = wx * wx + wy * wy + wz * wz;
    const double theta = sqrt(theta2);
    double a, b, c;      /* R = I + a*W + b*W^2 ; V = I + b*W + c*W^2 */
    double qe[4];
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
    /* W and W^2 */
    const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0; for (int k = 0; k < 3; ++k) s += W[i * 3 + k] * W[k * 3 + j];
            W2[i * 3 + j] = s;
        }
    double Re[9], V[9];
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        Re[i] = I + a * W[i] + b * W2[i];
        V[i] = I + b * W[i] + c * W2[i];
    }
    const double* t = pose + 4;
    double tn[3];
    for (int i = 0; i < 3; ++i) {
        tn[i] = V[i * 3 + 0] * d[3] + V[i * 3 + 1] * d[4] + V[i * 3 + 2] * d[5]
              + Re[i * 3 + 0] * t[0] + Re[i * 3 + 1] * t[1] + Re[i * 3 + 2] * t[2];
    }
    const double* q = pose;
    double qn[4];
    qn[0] = qe[0] * q[0] - qe[1] * q[1] - qe[2] * q[2] - qe[3] * q[3];
    qn[1] = qe[0] * q[1] + qe[1] * q[0] + qe[2] * q[3] - qe[3] * q[2];
    qn[2] = qe[0] * q[2] - qe[1] * q[3] + qe[2] * q[0] + qe[3] * q[1];
    qn[3] = qe[0] * q[3] + qe[1] * q[2] - qe[2] * q[1] + qe[3] * q[0];
    const double nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int i = 0; i < 4; ++i) pose[i] = qn[i] / nn;
    for (int i = 0; i < 3; ++i) pose[4 + i] = tn[i];
}

/* residual (obs - projection) and camera-frame point; returns dimension (2 mono / 3 stereo) */
static int residual(const double* R, const double* t, const double* X, const ora_ba_obs* o,
                    const ora_ba_cam* cam, double* e, double* pc)
{
    for (int i = 0; i < 3; ++i) pc[i] = R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2] + t[i];
    const double iz = 1.0 / pc[2];
    const double u = cam->fx * pc[0] * iz + cam->cx;
    const double v = cam->fy * pc[1] * iz + cam->cy;
    e[0] = o->u - u; e[1] = o->v - v;
    if (o->ur < 0) { e[2] = 0; return 2; }
    e[2] = o->ur - (u - cam->fxb * iz);
    return 3;
}

static void huber(double e2, double delta, double* rho0, double* rho1)
{
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; }
    else { const double sq = sqrt(e2); *rho0 = 2 * sq * delta - dsqr; *rho1 = delta / sq; }
}

static double robust_chi2(const double* poses, const double* points, const ora_ba_obs* obs,
                          const uint8_t* active, int n_obs, const ora_ba_cam* cam, int robust)
{
    double sum = 0;
    for (int k = 0; k < n_obs; ++k) {
        if (active && !active[k]) continue;
        double R[9], e[3], pc[3];
        quat_to_rot(poses + 7 * obs[k].pose, R);
        const int D = residual(R, poses + 7 * obs[k].pose + 4, points + 3 * obs[k].point, &obs[k], cam, e, pc);
        double chi = obs[k].inv_sigma2 * (e[0] * e[0] + e[1] * e[1] + (D == 3 ? e[2] * e[2] : 0.0));
        const double delta = D == 3 ? cam->huber_stereo : cam->huber_mono;
        if (robust && delta > 0) { double r0, r1; huber(chi, delta, &r0, &r1); chi = r0; }
        sum += chi;
    }
    return sum;
}

void ora_ba_chi2(const double* poses, const double* points, const ora_ba_obs* obs, int n_obs,
                 const ora_ba_cam* cam, double* chi2, uint8_t* depth_positive)
{
    for (int k = 0; k < n_obs; ++k) {
        double R[9], e[3], pc[3];
        quat_to_rot(poses + 7 * obs[k].pose, R);
        const int D = residual(R, poses + 7 * obs[k].pose + 4, points + 3 * obs[k].point, &obs[k], cam, e, pc);
        chi2[k] = obs[k].inv_sigma2 * (e[0] * e[0] + e[1] * e[1] + (D == 3 ? e[2] * e[2] : 0.0));
        if (depth_positive) depth_positive[k] = pc[2] > 0;
    }
}

/* dense Cholesky A = L L^T in place (lower); returns 0 if not positive definite */
static int chol_factor(double* A, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 1;
}
static void chol_solve(const double* L, int n, double* b)
{
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k]; b[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k]; b[i] = s / L[i * n + i]; }
}

static int inv3_sym(const double* A, double* Ai)
{
    const double a = A[0], b = A[1], c = A[2], d = A[4], e = A[5], f = A[8];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    if (!(fabs(det) > 0)) return 0;
    const double id = 1.0 / det;
    Ai[0] = c00 * id; Ai[1] = c01 * id; Ai[2] = c02 * id;
    Ai[3] = Ai[1]; Ai[4] = (a * f - c * c) * id; Ai[5] = (b * c - a * e) * id;
    Ai[6] = Ai[2]; Ai[7] = Ai[5]; Ai[8] = (a * d - b * b) * id;
    return 1;
}

typedef struct {
    int n_poses, n_points, n_obs, n_free, dim;
    int* pose_slot;          /* pose -> free slot or -1 */
    double *Hpp, *bp;        /* n_free x 36, n_free x 6 */
    double *Hll, *bl;        /* n_points x 9, n_points x 3 */
    double *Hpl;             /* n_obs x 18  (6x3, B^T W A) */
    double *S, *rhs, *xp, *xl;
    int points_fixed;
} ba_sys;

static void build_system(ba_sys* s, const double* poses, const double* points, const ora_ba_obs* obs,
                         const uint8_t* active, const ora_ba_cam* cam, int robust)
{
    memset(s->Hpp, 0, sizeof(double) * 36 * (size_t)(s->n_free > 0 ? s->n_free : 1));
    memset(s->bp, 0, sizeof(double) * 6 * (size_t)(s->n_free > 0 ? s->n_free : 1));
    memset(s->Hll, 0, sizeof(double) * 9 * (size_t)s->n_points);
    memset(s->bl, 0, sizeof(double) * 3 * (size_t)s->n_points);
    memset(s->Hpl, 0, sizeof(double) * 18 * (size_t)s->n_obs);
    for (int k = 0; k < s->n_obs; ++k) {
        if (active && !active[k]) continue;
        const ora_ba_obs* o = &obs[k];
        double R[9], e[3], pc[3];
        quat_to_rot(poses + 7 * o->pose, R);
        const int D = residual(R, poses + 7 * o->pose + 4, points + 3 * o->point, o, cam, e, pc);
        const double x = pc[0], y = pc[1], z = pc[2], z2 = z * z;
        double A[3][3], B[3][6];
        for (int c = 0; c < 3; ++c) {
            A[0][c] = -cam->fx * R[c] / z + cam->fx * x * R[6 + c] / z2;
            A[1][c] = -cam->fy * R[3 + c] / z + cam->fy * y * R[6 + c] / z2;
            A[2][c] = A[0][c] - cam->fxb * R[6 + c] / z2;
        }
        B[0][0] = x * y / z2 * cam->fx;          B[0][1] = -(1.0 + (x * x / z2)) * cam->fx; B[0][2] = y / z * cam->fx;
        B[0][3] = -1.0 / z * cam->fx;            B[0][4] = 0.0;                              B[0][5] = x / z2 * cam->fx;
        B[1][0] = (1.0 + y * y / z2) * cam->fy;  B[1][1] = -x * y / z2 * cam->fy;            B[1][2] = -x / z * cam->fy;
        B[1][3] = 0.0;                           B[1][4] = -1.0 / z * cam->fy;               B[1][5] = y / z2 * cam->fy;
        B[2][0] = B[0][0] - cam->fxb * y / z2;   B[2][1] = B[0][1] + cam->fxb * x / z2;      B[2][2] = B[0][2];
        B[2][3] = B[0][3];                       B[2][4] = 0.0;                              B[2][5] = B[0][5] - cam->fxb / z2;
        const double chi = o->inv_sigma2 * (e[0] * e[0] + e[1] * e[1] + (D == 3 ? e[2] * e[2] : 0.0));
        double w = o->inv_sigma2;      /* weighted Omega = rho1 * inv_sigma2 * I ; omega_r = -w * e */
        const double delta = D == 3 ? cam->huber_stereo : cam->huber_mono;
        if (robust && delta > 0) { double r0, r1; huber(chi, delta, &r0, &r1); w *= r1; }
        const int slot = s->pose_slot[o->pose];
        if (!s->points_fixed) {
            double* Hl = s->Hll + 9 * (size_t)o->point;
            double* bl = s->bl + 3 * (size_t)o->point;
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) {
                    double v = 0; for (int r = 0; r < D; ++r) v += A[r][a] * w * A[r][b];
                    Hl[a * 3 + b] += v;
                }
                double v = 0; for (int r = 0; r < D; ++r) v += A[r][a] * (-w * e[r]);
                bl[a] += v;
            }
        }
        if (slot >= 0) {
            double* Hp = s->Hpp + 36 * (size_t)slot;
            double* bp = s->bp + 6 * (size_t)slot;
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) {
                    double v = 0; for (int r = 0; r < D; ++r) v += B[r][a] * w * B[r][b];
                    Hp[a * 6 + b] += v;
                }
                double v = 0; for (int r = 0; r < D; ++r) v += B[r][a] * (-w * e[r]);
                bp[a] += v;
            }
            if (!s->points_fixed) {
                double* W = s->Hpl + 18 * (size_t)k;
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 3; ++b) {
                        double v = 0; for (int r = 0; r < D; ++r) v += B[r][a] * w * A[r][b];
                        W[a * 3 + b] = v;
                    }
            }
        }
    }
}

/* solve (H + lambda I) x = b with landmark Schur complement; returns 0 on failure */
static int solve_system(ba_sys* s, const ora_ba_obs* obs, const uint8_t* active, double lambda)
{
    const int n = s->dim;
    memset(s->S, 0, sizeof(double) * (size_t)n * n);
    for (int p = 0; p < s->n_free; ++p) {
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) s->S[(size_t)(6 * p + a) * n + 6 * p + b] = s->Hpp[36 * (size_t)p + a * 6 + b];
            s->S[(size_t)(6 * p + a) * n + 6 * p + a] += lambda;
            s->rhs[6 * p + a] = s->bp[6 * (size_t)p + a];
        }
    }
    double* Hinv = NULL;
    int *pt_start = NULL, *pt_list = NULL;
    if (!s->points_fixed) {
        Hinv = (double*)malloc(sizeof(double) * 9 * (size_t)s->n_points);
        for (int j = 0; j < s->n_points; ++j) {
            double H[9];
            memcpy(H, s->Hll + 9 * (size_t)j, sizeof(H));
            H[0] += lambda; H[4] += lambda; H[8] += lambda;
            if (!inv3_sym(H, Hinv + 9 * (size_t)j)) memset(Hinv + 9 * (size_t)j, 0, sizeof(double) * 9);
        }
        /* observations grouped by landmark (free poses only) */
        pt_start = (int*)calloc((size_t)s->n_points + 1, sizeof(int));
        for (int k = 0; k < s->n_obs; ++k)
            if ((!active || active[k]) && s->pose_slot[obs[k].pose] >= 0) pt_start[obs[k].point + 1]++;
        for (int j = 0; j < s->n_points; ++j) pt_start[j + 1] += pt_start[j];
        pt_list = (int*)malloc(sizeof(int) * (size_t)(pt_start[s->n_points] > 0 ? pt_start[s->n_points] : 1));
        int* fill = (int*)calloc((size_t)s->n_points, sizeof(int));
        for (int k = 0; k < s->n_obs; ++k)
            if ((!active || active[k]) && s->pose_slot[obs[k].pose] >= 0) {
                const int j = obs[k].point; pt_list[pt_start[j] + fill[j]++] = k;
            }
        free(fill);
        for (int j = 0; j < s->n_points; ++j) {
            const double* Hi = Hinv + 9 * (size_t)j;
            const double* bl = s->bl + 3 * (size_t)j;
            for (int a = pt_start[j]; a < pt_start[j + 1]; ++a) {
                const int ka = pt_list[a];
                const int pa = s->pose_slot[obs[ka].pose];
                const double* Wa = s->Hpl + 18 * (size_t)ka;
                double Y[18];       /* Y = W_a * Hll^-1 (6x3) */
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 3; ++c)
                        Y[r * 3 + c] = Wa[r * 3] * Hi[c] + Wa[r * 3 + 1] * Hi[3 + c] + Wa[r * 3 + 2] * Hi[6 + c];
                for (int r = 0; r < 6; ++r)
                    s->rhs[6 * pa + r] -= Y[r * 3] * bl[0] + Y[r * 3 + 1] * bl[1] + Y[r * 3 + 2] * bl[2];
                for (int b = pt_start[j]; b < pt_start[j + 1]; ++b) {
                    const int kb = pt_list[b];
                    const int pb = s->pose_slot[obs[kb].pose];
                    const double* Wb = s->Hpl + 18 * (size_t)kb;
                    for (int r = 0; r < 6; ++r)
                        for (int c = 0; c < 6; ++c)
                            s->S[(size_t)(6 * pa + r) * n + 6 * pb + c] -=
                                Y[r * 3] * Wb[c * 3] + Y[r * 3 + 1] * Wb[c * 3 + 1] + Y[r * 3 + 2] * Wb[c * 3 + 2];
                }
            }
        }
    }
    int ok = 1;
    if (n > 0) {
        ok = chol_factor(s->S, n);
        if (ok) { memcpy(s->xp, s->rhs, sizeof(double) * n); chol_solve(s->S, n, s->xp); }
    }
    if (ok && !s->points_fixed) {
        for (int j = 0; j < s->n_points; ++j) {
            double r[3] = {s->bl[3 * (size_t)j], s->bl[3 * (size_t)j + 1], s->bl[3 * (size_t)j + 2]};
            for (int a = pt_start[j]; a < pt_start[j + 1]; ++a) {
                const int ka = pt_list[a];
                const int pa = s->pose_slot[obs[ka].pose];
                const double* Wa = s->Hpl + 18 * (size_t)ka;
                for (int c = 0; c < 3; ++c)
                    for (int rr = 0; rr < 6; ++rr) r[c] -= Wa[rr * 3 + c] * s->xp[6 * pa + rr];
            }
            const double* Hi = Hinv + 9 * (size_t)j;
            for (int c = 0; c < 3; ++c) s->xl[3 * (size_t)j + c] = Hi[c * 3] * r[0] + Hi[c * 3 + 1] * r[1] + Hi[c * 3 + 2] * r[2];
        }
    }
    free(Hinv); free(pt_start); free(pt_list);
    return ok;
}

static int ba_run(double* poses, const uint8_t* fixed, int n_poses, double* points, int n_points,
                  const ora_ba_obs* obs, const uint8_t* active, int n_obs, const ora_ba_cam* cam,
                  int robust, int iters, int points_fixed, ora_ba_iter_log* log)
{
    ba_sys s; memset(&s, 0, sizeof(s));
    s.n_poses = n_poses; s.n_points = n_points; s.n_obs = n_obs; s.points_fixed = points_fixed;
    s.pose_slot = (int*)malloc(sizeof(int) * (size_t)n_poses);
    for (int i = 0; i < n_poses; ++i) s.pose_slot[i] = (fixed && fixed[i]) ? -1 : s.n_free++;
    s.dim = 6 * s.n_free;
    const size_t nf = s.n_free > 0 ? s.n_free : 1, np = n_points > 0 ? n_points : 1, no = n_obs > 0 ? n_obs : 1;
    s.Hpp = (double*)malloc(sizeof(double) * 36 * nf); s.bp = (double*)malloc(sizeof(double) * 6 * nf);
    s.Hll = (double*)malloc(sizeof(double) * 9 * np);  s.bl = (double*)malloc(sizeof(double) * 3 * np);
    s.Hpl = (double*)malloc(sizeof(double) * 18 * no);
    s.S = (double*)malloc(sizeof(double) * (size_t)(s.dim > 0 ? s.dim : 1) * (s.dim > 0 ? s.dim : 1));
    s.rhs = (double*)malloc(sizeof(double) * 6 * nf); s.xp = (double*)calloc(6 * nf, sizeof(double));
    s.xl = (double*)calloc(3 * np, sizeof(double));
    double* poses_bak = (double*)malloc(sizeof(double) * 7 * (size_t)n_poses);
    double* points_bak = (double*)malloc(sizeof(double) * 3 * np);

    double lambda = 0, ni = 2;
    int it = 0;
    for (; it < iters; ++it) {
        double current_chi = robust_chi2(poses, points, obs, active, n_obs, cam, robust);
        double temp_chi = current_chi;
        build_system(&s, poses, points, obs, active, cam, robust);
        if (it == 0) {      /* computeLambdaInit: tau * max |diag(H)| over all free vertices */
            double maxd = 0;
            for (int p = 0; p < s.n_free; ++p)
                for (int a = 0; a < 6; ++a) { const double v = fabs(s.Hpp[36 * (size_t)p + a * 7]); if (v > maxd) maxd = v; }
            if (!points_fixed)
                for (int j = 0; j < n_points; ++j)
                    for (int a = 0; a < 3; ++a) { const double v = fabs(s.Hll[9 * (size_t)j + a * 4]); if (v > maxd) maxd = v; }
            lambda = 1e-5 * maxd;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        const double chi_before = current_chi;
        do {
            memcpy(poses_bak, poses, sizeof(double) * 7 * (size_t)n_poses);                 /* push */
            if (!points_fixed) memcpy(points_bak, points, sizeof(double) * 3 * (size_t)n_points);
            const int ok2 = solve_system(&s, obs, active, lambda);
            if (ok2) {
                for (int i = 0; i < n_poses; ++i) if (s.pose_slot[i] >= 0) pose_oplus(poses + 7 * i, s.xp + 6 * s.pose_slot[i]);
                if (!points_fixed) for (int j = 0; j < 3 * n_points; ++j) points[j] += s.xl[j];
            }
            temp_chi = robust_chi2(poses, points, obs, active, n_obs, cam, robust);
            if (!ok2) temp_chi = DBL_MAX;
            rho = current_chi - temp_chi;
            double scale = 0;                /* computeScale: x^T (lambda x + b) over all unknowns */
            if (ok2) {
                for (int j = 0; j < s.dim; ++j) scale += s.xp[j] * (lambda * s.xp[j] + s.bp[j]);
                if (!points_fixed) for (int j = 0; j < 3 * n_points; ++j) scale += s.xl[j] * (lambda * s.xl[j] + s.bl[j]);
            }
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
                memcpy(poses, poses_bak, sizeof(double) * 7 * (size_t)n_poses);               /* pop */
                if (!points_fixed) memcpy(points, points_bak, sizeof(double) * 3 * (size_t)n_points);
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
    free(s.pose_slot); free(s.Hpp); free(s.bp); free(s.Hll); free(s.bl); free(s.Hpl);
    free(s.S); free(s.rhs); free(s.xp); free(s.xl); free(poses_bak); free(points_bak);
    return it;
}

int ora_ba_optimize(double* poses, const uint8_t* fixed, int n_poses, double* points, int n_points,
                    const ora_ba_obs* obs, const uint8_t* active, int n_obs, const ora_ba_cam* cam,
                    int robust, int iters, ora_ba_iter_log* log)
{
    return ba_run(poses, fixed, n_poses, points, n_points, obs, active, n_obs, cam, robust, iters, 0, log);
}

/* [UPSTREAM] optimize::local_bundle_adjuster::optimize: robust stage, outlier classification, plain stage */
int ora_ba_local(double* poses, const uint8_t* fixed, int n_poses, double* points, int n_points,
                 const ora_ba_obs* obs, int n_obs, const ora_ba_cam* cam,
                 int first_iters, int second_iters, uint8_t* outlier)
{
    uint8_t* active = (uint8_t*)malloc(n_obs > 0 ? n_obs : 1);
    double* chi = (double*)malloc(sizeof(double) * (size_t)(n_obs > 0 ? n_obs : 1));
    uint8_t* pos = (uint8_t*)malloc(n_obs > 0 ? n_obs : 1);
    memset(active, 1, n_obs);
    int it = ba_run(poses, fixed, n_poses, points, n_points, obs, active, n_obs, cam, 1, first_iters, 0, NULL);
    ora_ba_chi2(poses, points, obs, n_obs, cam, chi, pos);
    for (int k = 0; k < n_obs; ++k) {
        const double thr = obs[k].ur < 0 ? CHI2_2D : CHI2_3D;
        if (thr < chi[k] || !pos[k]) active[k] = 0;
    }
    it += ba_run(poses, fixed, n_poses, points, n_points, obs, active, n_obs, cam, 0, second_iters, 0, NULL);
    ora_ba_chi2(poses, points, obs, n_obs, cam, chi, pos);
    for (int k = 0; k < n_obs; ++k) {
        const double thr = obs[k].ur < 0 ? CHI2_2D : CHI2_3D;
        outlier[k] = (!active[k]) || (thr < chi[k]) || !pos[k];
    }
    free(active); free(chi); free(pos);
    return it;
}

/* [UPSTREAM] optimize::pose_optimizer::optimize: 4 trials x 10 iterations, unary edges, Huber dropped on the
 * third trial onwards, outliers re-classified after every trial */
int ora_pose_optimize(double* pose7, const double* points, const ora_ba_obs* obs, int n_obs,
                      const ora_ba_cam* cam, uint8_t* outlier)
{
    uint8_t* active = (uint8_t*)malloc(n_obs > 0 ? n_obs : 1);
    double* chi = (double*)malloc(sizeof(double) * (size_t)(n_obs > 0 ? n_obs : 1));
    memset(active, 1, n_obs);
    memset(outlier, 0, n_obs);
    int robust = 1, bad = 0;
    for (int trial = 0; trial < 4; ++trial) {
        ba_run(pose7, NULL, 1, (double*)points, 0, obs, active, n_obs, cam, robust, 10, 1, NULL);
        ora_ba_chi2(pose7, points, obs, n_obs, cam, chi, NULL);
        bad = 0;
        for (int k = 0; k < n_obs; ++k) {
            const double thr = obs[k].ur < 0 ? CHI2_2D : CHI2_3D;
            if (thr < chi[k]) { outlier[k] = 1; active[k] = 0; ++bad; }
            else { outlier[k] = 0; active[k] = 1; }
        }
        if (trial == 4 - 2) robust = 0;
        if (n_obs - bad < 5) break;
    }
    free(active); free(chi);
    return n_obs - bad;
}
