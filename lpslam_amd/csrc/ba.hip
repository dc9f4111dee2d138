// ba.hip -- SE3 bundle adjustment on gfx950 (FP64): g2o-style Levenberg-Marquardt with landmark Schur complement.
//
// [UPSTREAM] g2o@691dc51 OptimizationAlgorithmLevenberg + BlockSolver<6,3> (buildSystem / Schur / solve) and the
// OpenVSLAM reprojection edges, which the reference runs on its mapping / global-optimisation threads
// (/root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:239,250-255; pin conan-packages/g2o-conan/conanfile.py:6).
//
// The whole LM loop is device driven: lambda, nu, the accepted-state index and the accept / reject decision live in a
// small control block in HBM, every kernel reads it, and the host only enqueues "trial units" and looks at the control
// block once per optimize() call.  A unit = [linearise if the state changed] + one LM trial:
//   linearise   k_ba_lin (observation side: thread / observation, W = B^T w A and shares of H_ll, b_l; pose side: 8
//               wavefronts / keyframe, H_pp, b_p, chi2; both in one launch), k_ba_point_sum (H_ll, b_l; its last workgroup
//               combines the pose partials in fixed order and computes lambda_0)
//   trial       k_ba_schur (wavefront / pose-block pair over a pair list; Y = W (H_ll + lambda)^-1 formed per term),
//               k_chol_pair x nb/2 (32-wide panels, two per launch, block products on the f64 matrix cores; the rhs is carried as an extra
//               row and L^-T as extra row blocks, so no triangular substitution is needed), k_chol_xsolve (x_p = L^-T y),
//               k_ba_backsub (landmarks, trial poses), k_ba_trial (trial chi2; its last workgroup runs g2o's lambda control)
// All sums are fixed-order segmented reductions (no float atomics): results are reproducible run to run.
// The reduced system [S | rhs | b_p | diag H_pp | chi2] is one contiguous buffer, so a landmark-partitioned multi-GPU
// solve needs one sum all-reduce of it per trial (lpslam_hip_ba_step_*).
#include "internal.h"
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstddef>
#include <cfloat>
#include <algorithm>
#include <atomic>
#include <map>
#include <string>
#include <dlfcn.h>

#pragma clang fp contract(off)

using namespace lpslam;

namespace {

constexpr int NB = 32;                // Cholesky panel width
constexpr int SCH_PV = 42;            // values a part of a pose-block pair's list hands over in k_ba_schur: the 6 x 6 sum + the 6 of the keyframe's rhs (diagonal blocks)
// How many workgroups of k_ba_schur share a pose-block pair's list: lists longer than 256 terms are cut into up to SCH_MAXP interleaved parts
// (32-term chunks round-robin), so that the longest list -- a keyframe's diagonal block, one term per observation -- does not set the
// kernel's duration; the part that finishes last adds the parts up in order (fixed summation order)
constexpr int SCH_MAXP = 4, SCH_PART = 256;      // (8 parts of ~100 terms measured: 23.0 us against 22.2 -- the surplus workgroups of the larger table cost what the shorter chains gain)
__host__ __device__ inline int schur_parts(int n_terms) { return n_terms > 256 ? ((n_terms + SCH_PART - 1) / SCH_PART < SCH_MAXP ? (n_terms + SCH_PART - 1) / SCH_PART : SCH_MAXP) : 1; }
constexpr int SPLIT = 8;              // wavefronts per keyframe in the pose pass
constexpr int PV = 28;                // partial-row stride per wavefront: 21 (H_pp upper) + 6 (b_p) (+1 pad; chi2 is kept apart)
constexpr int MAX_LOG = 64;
__host__ __device__ inline bool cw_fits(int dim);      // the system fits the single-workgroup factorisation (ba_solve.inl)

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct BaCam { double fx, fy, cx, cy, fxb, hub_mono, hub_stereo; };

struct BaCtl {                        // device-resident LM state (g2o OptimizationAlgorithmLevenberg)
    double lambda, ni, current_chi, chi_before, rho;
    int cur;                          // buffer index of the accepted state
    int need_lin;                     // the next unit must linearise first
    int first;                        // lambda_0 not yet computed in this optimize() call
    int qmax;                         // trials of the running outer iteration
    int outer_done, max_outer;
    int stopped;                      // g2o "Terminate"
    int last_accepted;
    int ticket;                       // workgroups of the running pass that have published their partials (last one combines)
    int cur_launch;                   // copy of `cur` that stays put while a trial launch runs (the decision flips `cur` inside it)
    int spec;                         // linearisation set [cur] already holds the linearisation of state cur (speculated beside the trial)
    int faults_band, faults_update;   // host copy only (k_ba_collect fills them from ba_sync_words): hand-overs that timed out, cumulative
};

// Pointer members of the view are typed as global-address-space pointers in the device pass: a view is read from device memory
// (views[blockIdx.y], scalar loads), and a pointer that comes out of memory is a generic ("flat") pointer to the compiler --
// flat loads count on both vmcnt and lgkmcnt and return out of order, so every wait on them is a full drain.  With the address
// space in the type every access through the view is a global_load / global_store with counted waits, as with by-value kernel
// arguments.  The host pass sees plain pointers of the same size and layout.
#if defined(__HIP_DEVICE_COMPILE__)
#define GPTR(T) __attribute__((address_space(1))) T*
#else
#define GPTR(T) T*
#endif
template <class D, class S> inline void vset(D& d, S* s) { d = (D)s; }       // host side: generic pointer into a view member

struct BaView {                       // one problem, resident in device memory (kernels index an array of them by blockIdx.y)
    int n_poses, n_points, n_obs, n_free, dim, dim_pad;
    // launch extents of this problem (a batch launches the maximum over its problems; surplus workgroups exit at once)
    int obs_blocks;                   // ceil(n_obs / 256): observation-side linearisation
    int pose_blocks;                  // ceil(n_poses * SPLIT / 4): pose-side linearisation / trial chi2
    int point_blocks;                 // ceil(n_points / 256): k_ba_point_sum
    int part_n;                       // max(ceil(n_points / 64), 1): k_ba_backsub landmark blocks = entries of `part`
    int n_blocks;                     // pose-block pairs (n_free (n_free + 1) / 2)
    int land_blocks;                  // ceil(n_points / LAND_B): landmark-major linearisation beside a trial (land_lin_body)
    GPTR(double) poses_buf[2]; GPTR(double) points_buf[2];
    GPTR(const double) poses; GPTR(const double) points;        // set by the kernel prologue (state being evaluated)
    GPTR(const double) poses0; GPTR(const double) points0;      // state given at creation (reset)
    GPTR(const int) pose_slot; GPTR(const int) free_pose;
    GPTR(const int) o_pose; GPTR(const int) o_point;
    GPTR(const double) o_u; GPTR(const double) o_v; GPTR(const double) o_ur; GPTR(const double) o_w;
    GPTR(uint8_t) o_active;
    GPTR(const int) pt_start; GPTR(const int) pt_obs; GPTR(const int) ps_start; GPTR(const int) o_orig;
    GPTR(double) W; GPTR(double) Hll; GPTR(double) bl; GPTR(double) Hpp; GPTR(double) hl_obs; GPTR(double) partial;   // linearisation set in use (ba_lin_set)
    // Both linearisation sets (indexed like the state buffers), each as TWO blocks whose sub-arrays sit at offsets that follow from
    // the problem's sizes (SetOff): set_z = [partial | b_p, diag H_pp, chi2] (zero-initialised), set_d = [H_ll | b_l | H_pp | W | hl].
    // One pointer pair per set instead of seven keeps the by-value view small enough to live in scalar registers (at 672 bytes the
    // compiler kept a copy in scratch memory and every member access became a scratch load: trial launch 14 -> 41 us).
    GPTR(double) set_z[2]; GPTR(double) set_d[2]; GPTR(double) partial_trial;     // + trial chi2 partials
    GPTR(const double) csr;            // observation constants once more in CSR (landmark-major) order [u | v | ur | w | pose, point (int) | pose slot (int)]
    GPTR(const int) land_start;        // landmark blocks of the landmark-major passes: (first landmark, first CSR entry) per block + a closing pair; <= LAND_B landmarks and -- unless one landmark alone has more -- <= 256 entries each, cut on the host at creation
    GPTR(double) S; GPTR(double) rhs; GPTR(double) bp; GPTR(double) hppdiag; GPTR(double) chi_cur;   // reduced buffer sections (all-reduced when partitioned)
    GPTR(double) bp_loc; GPTR(double) hppdiag_loc; GPTR(double) chi_loc;                    // this rank's own sums (equal to the above on one GPU)
    GPTR(double) Minv;                                                            // L^-T row blocks (dim_pad x dim_pad)
    GPTR(double) Ldiag;                                                           // factored diagonal blocks [nb][32][32]
    GPTR(double) Lsub;                                                            // L_j1,j of every panel pair, stored at [j1][32][32]
    GPTR(double) xp; GPTR(double) chi_pose; GPTR(double) part; GPTR(double) scal;
    GPTR(const int) blk_start; GPTR(const int4) blk_terms;        // Schur pair lists: (observation a, observation b, their landmark, -)
    GPTR(double) blk_part; GPTR(int) blk_ticket;                  // Schur partial sums [block][SCH_MAXP][SCH_PV], per-block tickets (+ the table of further parts, ba_build.inl)
    GPTR(int) blk_perm;                                           // k_ba_schur: which pose-block pair work item w takes (XCD tiles, see lpslam_hip_ba_prepare)
    GPTR(BaCtl) ctl; GPTR(lpslam_hip_ba_iter_log) log;
    BaCam cam;
    // block-banded windows (ba_band.inl): block half-bandwidth of the reduced system when the problem takes the band path (-1: pair
    // lists + dense chain), landmark groups, [group records | first / last candidate group per free slot], entry table, group partials
    int band_hbw, band_groups, band_groups_cap;
    int extra_pack;                   // k_ba_schur: workgroups for the further parts of long pair lists (table behind blk_ticket[n_blocks]: count, items):
                                      // (cap << 12) | first -- the table holds at most `cap` items, `first` of them get workgroups in FRONT of the pairs'
                                      // part 0 (what the host expects: the diagonal blocks' parts), the rest behind them (one int: the view's size matters)
    GPTR(const int) band_tab; GPTR(const int) band_ent; GPTR(double) band_part;
};

// Words beside the eight scalars of v.scal that are NOT part of the control block (lm_begin / lm_decide rewrite that as a whole):
// [0] hand-overs of the twisted band factorisation that timed out, [1] keyframe blocks of k_ba_update that timed out (both stay 0;
// lpslam_hip_ba_timeouts; [1] also counts k_ba_schur blocks whose wait for the pose side timed out), [2] unused, [3] "pose side
// pending": the accepted state's H_pp, b_p are the next Schur launch's to compute (set by k_ba_update's decision, ba_update.inl),
// [4] wavefronts of that launch that have stored theirs.
__device__ __forceinline__ int* ba_sync_words(const BaView& v) { return (int*)(double*)(v.scal + 8); }

// Which problems take the one-launch update behind the fused solve (k_ba_update, ba_update.inl), decided per PROBLEM so that a problem
// is solved by the same kernels -- to the same bytes -- alone or inside a mixed batch; the two-launch form (k_ba_backsub, k_ba_trial)
// skips those problems when it runs beside it.
constexpr int UPD_MAXP = 320;               // keyframes whose trial poses fit the landmark blocks' LDS (config 5: 200)
__host__ __device__ inline bool upd_takes(int n_points, int n_free, int n_poses) { return n_points >= 1 && n_free >= 1 && n_poses <= UPD_MAXP; }

// The view of problem blockIdx.y.  `views` is const __restrict__ and read before any store of the kernel: scalar loads.
#define BA_VIEW(v) BaView v = views[blockIdx.y]
// A batch of problems (grid.y = problems) with every problem's workgroups on ONE XCD: workgroups go to the eight XCDs round robin
// in dispatch order (x fastest), so workgroup L of the launch is given to problem L % 8 (+ 8 per full round of a problem's
// workgroups).  What a problem's workgroups read again and again -- the W rows in the Schur kernel, 8 times each -- then comes out
// of one L2 instead of being fetched into all eight (a 16-window batch: 444 MB of FETCH_SIZE per Schur launch against the 93 MB
// the batch holds; 4.24 -> 4.05 ms per batch of 16 x 10 iterations).  Speed only: the mapping is a bijection whatever the
// placement is.  Needs grid.y to be a multiple of 8.
struct BaWg { int x, y; };
__device__ __forceinline__ BaWg ba_wg_xcd()
{
    int x = blockIdx.x, y = blockIdx.y;
    const int ny = gridDim.y, nx = gridDim.x;
    if (ny >= 8 && (ny & 7) == 0) {
        const unsigned L = (unsigned)y * (unsigned)nx + (unsigned)x, slot = L >> 3;
        y = (int)(L & 7u) + 8 * (int)(slot / (unsigned)nx);
        x = (int)(slot % (unsigned)nx);
    }
    return BaWg{x, y};
}
#define BA_VIEW_XCD(v, bx) const BaWg wg_ = ba_wg_xcd(); BaView v = views[wg_.y]; const int bx = wg_.x
// The members a kernel needs before its first branch, made live together: one scalar round trip for the extents AND the control
// block pointer (left alone the compiler loads the pointer only behind the extent test, one round trip later).
#define BA_VIEW_HEAD(...) asm volatile("" :: __VA_ARGS__)

__device__ __forceinline__ bool ba_idle(const BaCtl* c) { return c->stopped || c->outer_done >= c->max_outer; }
// What the kernels read of the control block, fetched in ONE round trip (straight-line loads, no branch between them): written as
// `if (ba_idle(ctl) || !ctl->need_lin) return; idx = ctl->cur;` every member was a dependent round trip of its own in front of the
// kernel's real work.
struct BaFlags {
    double lambda; int cur, need_lin, outer_done, max_outer, stopped, cur_launch, spec;
    __device__ __forceinline__ bool idle() const { return (stopped != 0) | (outer_done >= max_outer); }
};
__device__ __forceinline__ BaFlags ba_flags(const BaCtl* c)
{
    BaFlags f;
    f.lambda = c->lambda; f.cur = c->cur; f.need_lin = c->need_lin; f.outer_done = c->outer_done; f.max_outer = c->max_outer;
    f.stopped = c->stopped; f.cur_launch = c->cur_launch; f.spec = c->spec;
    return f;
}
// Member [idx] of a two-element pointer array of the view, by ARITHMETIC on its two constant-index members: a select between two
// members (or a dynamic index) can end up as an indexed load from a copy of the whole view in scratch memory -- it did when the
// view grew (552 bytes of scratch per lane, every member access a scratch load, trial launch 14 -> 41 us).
template <class P> __device__ __forceinline__ P sel2(P a0, P a1, int idx) { return a0 + (ptrdiff_t)idx * (a1 - a0); }
// selects the evaluated state: the accepted one (trial = 0) or the trial one
__device__ __forceinline__ void ba_select(BaView& v, int trial)
{
    const int s = v.ctl->cur ^ trial;
    v.poses = sel2(v.poses_buf[0], v.poses_buf[1], s); v.points = sel2(v.points_buf[0], v.points_buf[1], s);
}
// state buffer and linearisation set by explicit index (kernels that must not follow a `cur` flipping under them)
__device__ __forceinline__ void ba_select_idx(BaView& v, int idx)
{
    v.poses = sel2(v.poses_buf[0], v.poses_buf[1], idx); v.points = sel2(v.points_buf[0], v.points_buf[1], idx);
}
// offsets (in doubles, multiples of 32 = 256 bytes) of the sub-arrays of a linearisation set's two blocks and of the CSR copies
struct SetOff { size_t loc, z_total, bl, Hpp, W, hl, d_total; };
__host__ __device__ inline size_t al32(size_t x) { return (x + 31) & ~(size_t)31; }
__host__ __device__ inline SetOff set_offsets(int n_poses, int n_points, int n_obs, int n_free, int dim_pad)
{
    const size_t np = (size_t)n_poses, npt = (size_t)(n_points > 1 ? n_points : 1), no = (size_t)(n_obs > 1 ? n_obs : 1), nf = (size_t)(n_free > 1 ? n_free : 1);
    SetOff o;
    o.loc = al32(np * SPLIT * (PV + 1)); o.z_total = o.loc + al32(2 * (size_t)dim_pad + 8);
    o.bl = al32(6 * npt); o.Hpp = o.bl + al32(3 * npt); o.W = o.Hpp + al32(36 * nf); o.hl = o.W + al32(18 * no); o.d_total = o.hl + al32(9 * no);
    return o;
}
__host__ __device__ inline size_t csr_stride(int n_obs) { return al32((size_t)(n_obs > 1 ? n_obs : 1)); }
__device__ __forceinline__ void ba_lin_set(BaView& v, int idx)
{
    GPTR(double) z = sel2(v.set_z[0], v.set_z[1], idx);
    GPTR(double) d = sel2(v.set_d[0], v.set_d[1], idx);
    const SetOff o = set_offsets(v.n_poses, v.n_points, v.n_obs, v.n_free, v.dim_pad);
    v.partial = z;
    GPTR(double) loc = z + o.loc;
    v.bp_loc = loc; v.hppdiag_loc = loc + v.dim_pad; v.chi_loc = loc + 2 * v.dim_pad;
    v.Hll = d; v.bl = d + o.bl; v.Hpp = d + o.Hpp; v.W = d + o.W; v.hl_obs = d + o.hl;
}

// Reciprocal and reciprocal square root for the per-observation arithmetic: v_rcp_f64 / v_rsq_f64 (2^-24, measured) plus ONE cubic
// correction step -- five instructions and 1.4e-16 maximum relative error (4M samples, tools/dev/rsq_acc.hip) where IEEE division
// and sqrt are ~30 instructions each.  A reprojection Jacobian held thirteen divisions: across a window's 39 k observations and
// their three passes per LM iteration that was most of the arithmetic of the linearising kernels.  Not correctly rounded: results
// move in the last bits against a libm evaluation (tests: chi2 trajectories 1e-9 relative, poses 1e-4 rad / 1e-3 m).
__device__ __forceinline__ double fast_rcp(double d)       // 1 / d
{
    const double y0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, y0, 1.0);
    return fma(y0, fma(e, e, e), y0);                   // y0 (1 + e + e^2)
}
__device__ __forceinline__ double fast_rsqrt(double d)     // 1 / sqrt(d), d > 0
{
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * y0), y0, 1.0);
    return fma(y0 * e, fma(0.375, e, 0.5), y0);         // y0 (1 + e / 2 + 3 e^2 / 8)
}
__device__ __forceinline__ void quat_to_rot(const double* q, double* R)
{
    const double rn = fast_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] * rn, x = q[1] * rn, y = q[2] * rn, z = q[3] * rn;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// residual e = obs - projection, camera-frame point pc; returns 2 (mono) or 3 (stereo).  The camera travels BY VALUE through these
// helpers: handed on as a reference into the by-value view, the compiler kept the whole view in scratch memory (552 bytes per lane).
__device__ __forceinline__ int ba_residual_vals(const BaCam cam, double ou, double ov, double ur, const double* R, const double* t, const double* X,
                                                double* e, double* pc)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2] + t[i];
    const double iz = fast_rcp(pc[2]);
    const double u = cam.fx * pc[0] * iz + cam.cx;
    const double vv = cam.fy * pc[1] * iz + cam.cy;
    e[0] = ou - u; e[1] = ov - vv;
    if (ur < 0) { e[2] = 0; return 2; }
    e[2] = ur - (u - cam.fxb * iz);
    return 3;
}
__device__ __forceinline__ int ba_residual(const BaView& v, int k, const double* R, const double* t, const double* X,
                                           double* e, double* pc)
{
    return ba_residual_vals(v.cam, v.o_u[k], v.o_v[k], v.o_ur[k], R, t, X, e, pc);
}

__device__ __forceinline__ void huber(double e2, double delta, double* rho0, double* rho1)
{
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; }
    else { const double rs = fast_rsqrt(e2); *rho0 = 2 * (e2 * rs) * delta - dsqr; *rho1 = delta * rs; }
}

// Jacobians of the reprojection error: A (D x 3, landmark), B (D x 6, pose, rotation first)
__device__ __forceinline__ void ba_jacobians(const BaCam c, const double* R, const double* pc, int D, double A[3][3], double B[3][6])
{
    const double x = pc[0], y = pc[1], iz = fast_rcp(pc[2]), iz2 = iz * iz;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        A[0][k] = -c.fx * R[k] * iz + c.fx * x * R[6 + k] * iz2;
        A[1][k] = -c.fy * R[3 + k] * iz + c.fy * y * R[6 + k] * iz2;
        A[2][k] = A[0][k] - c.fxb * R[6 + k] * iz2;
    }
    B[0][0] = x * y * iz2 * c.fx;          B[0][1] = -(1.0 + (x * x * iz2)) * c.fx; B[0][2] = y * iz * c.fx;
    B[0][3] = -iz * c.fx;                  B[0][4] = 0.0;                            B[0][5] = x * iz2 * c.fx;
    B[1][0] = (1.0 + y * y * iz2) * c.fy;  B[1][1] = -x * y * iz2 * c.fy;            B[1][2] = -x * iz * c.fy;
    B[1][3] = 0.0;                         B[1][4] = -iz * c.fy;                     B[1][5] = y * iz2 * c.fy;
    B[2][0] = B[0][0] - c.fxb * y * iz2;   B[2][1] = B[0][1] + c.fxb * x * iz2;      B[2][2] = B[0][2];
    B[2][3] = B[0][3];                     B[2][4] = 0.0;                            B[2][5] = B[0][5] - c.fxb * iz2;
    if (D == 2) {      // monocular edge: the third row does not exist; zero rows keep every sum exact and loops unrolled
#pragma unroll
        for (int k = 0; k < 3; ++k) A[2][k] = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) B[2][k] = 0.0;
    }
}

// weight (rho1 * inv_sigma2) and robustified chi2 of one observation
__device__ __forceinline__ double ba_weight_vals(const BaCam cam, double om, int D, const double* e, int robust, double* rho0)
{
    const double chi = om * (e[0] * e[0] + e[1] * e[1] + (D == 3 ? e[2] * e[2] : 0.0));
    const double delta = D == 3 ? cam.hub_stereo : cam.hub_mono;
    double w = om;
    *rho0 = chi;
    if (robust && delta > 0) { double r1; huber(chi, delta, rho0, &r1); w *= r1; }
    return w;
}
__device__ __forceinline__ double ba_weight(const BaView& v, int k, int D, const double* e, int robust, double* rho0)
{
    return ba_weight_vals(v.cam, v.o_w[k], D, e, robust, rho0);
}


// "Last workgroup done" hand-over: every workgroup of a pass calls this after its partials are stored.  Returns true in
// exactly one workgroup -- the one that arrives last -- with all other workgroups' stores visible (producer: every wavefront
// drains its stores, barrier, one lane's agent-scope release + ticket; consumer: agent-scope acquire by that lane, its wait,
// barrier, plain loads -- MI355X_MICROARCH.md, inter-workgroup visibility); that workgroup then runs the single-workgroup
// combine, which saves a launch.  No spinning, so the grid always drains.
// Hand-over to the workgroup that finishes last, without cache maintenance: every byte the last workgroup reads from the others
// is stored write-through (st_sc1) and read L1-bypassing (ld_sc1); each storing wavefront drains its stores, the workgroup
// meets at a barrier and one lane takes a relaxed agent-scope ticket (MI355X_MICROARCH, valid forms: one unsharded counter, the
// consumer is the workgroup whose add came last).  An acquire-release pair here would cost a buffer_wbl2 + buffer_inv, ~3.5 us.
__device__ __forceinline__ void st_sc1(double* p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void st_sc1(GPTR(double) p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // global_, not flat_
__device__ __forceinline__ double ld_sc1(GPTR(const double) p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(GPTR(double) p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
__device__ __forceinline__ bool ba_last_block_sc1(int* ticket, int total)
{
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wavefront drains its own stores (a barrier alone does not)
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == total - 1);
        if (s_last) *ticket = 0;
    }
    __syncthreads();
    return s_last != 0;
}
__device__ __forceinline__ bool ba_last_block_sc1(BaCtl* c, int total) { return ba_last_block_sc1(&c->ticket, total); }
// The general form (any plain stores before it are visible to the last workgroup's plain loads after it): agent-scope
// acquire-release on the ticket, i.e. an L2 write-back and an L1 invalidate per workgroup.  Used where the handed-over data
// are not confined to a few words (sim3.inl).
__device__ __forceinline__ bool ba_last_block(BaCtl* c, int total)
{
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wavefront drains its own stores (a barrier alone does not)
    __syncthreads();
    if (threadIdx.x == 0) {
        // release: the workgroup's stores leave this XCD's L2; acquire (only the last arrival needs it): stale lines are dropped
        const int t = __hip_atomic_fetch_add(&c->ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == total - 1);
        if (s_last) c->ticket = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // holds the barrier below until the invalidate has completed
    }
    __syncthreads();
    return s_last != 0;
}

// (H_ll + lambda I)^-1 of one landmark, symmetric 3x3 stored as 6 (zero when singular)
__device__ __forceinline__ void point_hinv(const double* hl, double lambda, double* ho)
{
    const double a = hl[0] + lambda, b = hl[1], c = hl[2], d = hl[3] + lambda, e = hl[4], f = hl[5] + lambda;
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    if (fabs(det) > 0) {
        const double id = fast_rcp(det);
        ho[0] = c00 * id; ho[1] = c01 * id; ho[2] = c02 * id;
        ho[3] = (a * f - c * c) * id; ho[4] = (b * c - a * e) * id; ho[5] = (a * d - b * b) * id;
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) ho[i] = 0;
    }
}

__device__ __forceinline__ double wave_sum(double x)
{
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ double wave_max(double x)
{
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
    return x;
}

// ---- linearisation, observation side: one thread per observation: W = B^T w A and the observation's share of H_ll, b_l --
__device__ __forceinline__ void obs_lin_body(BaView& v, int bid, int robust, int points_fixed, int set)
{
    ba_select_idx(v, set); ba_lin_set(v, set);
    const int k = bid * 256 + threadIdx.x;
    if (k >= v.n_obs) return;
    double* Wk = v.W + 18 * (size_t)k;
    double* ho = v.hl_obs + 9 * (size_t)k;
    const int p = v.o_pose[k];
    const int slot = v.pose_slot[p];
    if (!v.o_active[k] || points_fixed) {       // inactive edge, or motion-only mode (landmarks are constants: x_l = 0)
#pragma unroll
        for (int i = 0; i < 18; ++i) Wk[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) ho[i] = 0.0;
        return;
    }
    const int j = v.o_point[k];
    const double X[3] = {v.points[3 * j], v.points[3 * j + 1], v.points[3 * j + 2]};
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
#pragma unroll
            for (int r = 0; r < 3; ++r) s2 += A[r][a] * w * A[r][c];
            ho[idx++] = s2;
        }
        double s3 = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) s3 += A[r][a] * (-w * e[r]);
        ho[6 + a] = s3;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double s2 = 0;
            if (slot >= 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) s2 += B[r][a] * w * A[r][c];
            }
            Wk[a * 3 + c] = s2;
        }
}

// ---- linearisation, observation side, LANDMARK-MAJOR: what obs_lin_body + the landmark sums of k_ba_point_sum compute, in one pass.
//      A workgroup owns LAND_B consecutive landmarks and walks their observations in CSR order (= the order k_ba_point_sum adds
//      them in), 256 at a time: thread = observation -> W (to the observation's storage slot) and its 9 shares of (H_ll, b_l) into
//      LDS; then thread (landmark l, component c) adds the shares of its landmark's segment of the chunk in ascending order -- the
//      same sums, bit for bit, without the 72 bytes per observation going to memory and coming back in a launch of their own.
//      The observation constants are read from their CSR-ordered copies (c_*), coalesced.
constexpr int LAND_B = 32;
__device__ __forceinline__ void land_lin_body(BaView& v, int bid, int robust, int points_fixed, int set)
{
    ba_select_idx(v, set); ba_lin_set(v, set);
    __shared__ double sh[256 * 9];
    __shared__ int s_start[LAND_B + 1];
    const int tid = threadIdx.x;
    const int j0 = v.land_start[2 * bid], j1 = v.land_start[2 * bid + 2];
    if (tid <= LAND_B) s_start[tid] = v.pt_start[min(j0 + tid, j1)];
    __syncthreads();
    const int s_lo = s_start[0], s_hi = s_start[j1 - j0];
    const size_t cs = csr_stride(v.n_obs);
    GPTR(const double) c_u = v.csr; GPTR(const double) c_v = v.csr + cs; GPTR(const double) c_ur = v.csr + 2 * cs; GPTR(const double) c_w = v.csr + 3 * cs;
    GPTR(const int) c_pose = (GPTR(const int))(v.csr + 4 * cs); GPTR(const int) c_point = c_pose + cs;
    const int l = tid >> 3, c = tid & 7;                   // the sums: landmark j0 + l, components c (and 8 with c == 0)
    const int seg_lo = s_start[min(l, j1 - j0)], seg_hi = s_start[min(l + 1, j1 - j0)];
    double acc = 0, acc8 = 0;
    for (int chunk = s_lo; chunk < s_hi; chunk += 256) {
        const int s = chunk + tid;
        double hs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (s < s_hi) {
            const int k = v.pt_obs[s];
            const int p = c_pose[s];
            const int slot = v.pose_slot[p];
            double* Wk = v.W + 18 * (size_t)k;
            if (!v.o_active[k] || points_fixed) {
#pragma unroll
                for (int i = 0; i < 18; ++i) Wk[i] = 0.0;
            } else {
                const int j = c_point[s];
                const double X[3] = {v.points[3 * j], v.points[3 * j + 1], v.points[3 * j + 2]};
                double R[9], e[3], pc[3], A[3][3], B[3][6], rho0;
                quat_to_rot(v.poses + 7 * p, R);
                const int D = ba_residual_vals(v.cam, c_u[s], c_v[s], c_ur[s], R, v.poses + 7 * p + 4, X, e, pc);
                ba_jacobians(v.cam, R, pc, D, A, B);
                const double w = ba_weight_vals(v.cam, c_w[s], D, e, robust, &rho0);
                int idx = 0;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int cc = a; cc < 3; ++cc) {
                        double s2 = 0;
#pragma unroll
                        for (int r = 0; r < 3; ++r) s2 += A[r][a] * w * A[r][cc];
                        hs[idx++] = s2;
                    }
                    double s3 = 0;
#pragma unroll
                    for (int r = 0; r < 3; ++r) s3 += A[r][a] * (-w * e[r]);
                    hs[6 + a] = s3;
                }
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        double s2 = 0;
                        if (slot >= 0) {
#pragma unroll
                            for (int r = 0; r < 3; ++r) s2 += B[r][a] * w * A[r][cc];
                        }
                        Wk[a * 3 + cc] = s2;
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) sh[tid * 9 + i] = hs[i];
        __syncthreads();
        const int a0 = max(seg_lo, chunk), a1 = min(seg_hi, chunk + 256);
        for (int t = a0; t < a1; ++t) {
            acc += sh[(t - chunk) * 9 + c];
            if (c == 0) acc8 += sh[(t - chunk) * 9 + 8];
        }
        __syncthreads();
    }
    const int j = j0 + l;
    if (j < j1) {
        if (c < 6) v.Hll[6 * (size_t)j + c] = acc; else v.bl[3 * (size_t)j + (c - 6)] = acc;
        if (c == 0) v.bl[3 * (size_t)j + 2] = acc8;
    }
}

// ---- linearisation 2/2: H_ll, b_l per landmark = fixed-order sum over its observations; block maxima of diag H_ll.  The
//      One extra workgroup combines the pose partials of the linearisation (pose_combine_body, defined below); the workgroup
//      that finishes last starts the outer iteration.
__device__ __forceinline__ void pose_combine_body(BaView& v, int mode, int part_n, int fused);
__device__ __forceinline__ void lm_begin(BaView& v, double max_diag_pp, double max_diag_ll, double chi_cur);
__global__ __launch_bounds__(256) void k_ba_point_sum(const BaView* __restrict__ views, int fused)
{
    BA_VIEW(v);
    BA_VIEW_HEAD("s"(v.point_blocks), "s"(v.ctl));
    const int part_n = v.point_blocks;                     // landmark workgroups of this problem; workgroup part_n combines the pose partials
    if ((int)blockIdx.x > part_n) return;
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle() || !fl.need_lin) return;
    ba_lin_set(v, fl.cur);
    __shared__ double sm[4];
    if ((int)blockIdx.x == part_n) {
        // the extra workgroup: the pose partials of the linearisation are complete before this launch, so they are combined
        // here, beside the landmark sums, instead of on the critical path of the workgroup that finishes last
        pose_combine_body(v, 0, part_n, 0);
    } else {
        const int j = blockIdx.x * 256 + threadIdx.x;
        double m = 0;
        if (j < v.n_points) {
            double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            // four observations at a time: their index loads together, then their 36 doubles together, then the sums in the fixed
            // order (one observation per pass meant two dependent round trips each: the kernel was their latency)
            const int s_end = v.pt_start[j + 1];
            for (int s = v.pt_start[j]; s < s_end; s += 4) {
                int kk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kk[u] = v.pt_obs[min(s + u, s_end - 1)];
                double hv[4][9];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double* ho = v.hl_obs + 9 * (size_t)kk[u];
#pragma unroll
                    for (int i = 0; i < 9; ++i) hv[u][i] = ho[i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (s + u < s_end) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) acc[i] += hv[u][i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) v.Hll[6 * (size_t)j + i] = acc[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) v.bl[3 * (size_t)j + i] = acc[6 + i];
            m = fmax(fabs(acc[0]), fmax(fabs(acc[3]), fabs(acc[5])));
        }
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0 && (int)blockIdx.x < part_n) st_sc1(&v.part[blockIdx.x], fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3])));
    }
    if (!ba_last_block_sc1(v.ctl, part_n + 1)) return;
    // last workgroup: max diag H_ll over the block maxima, then the start of the outer iteration (lambda_0)
    if (threadIdx.x < 64) {
        double acc = 0;
        for (int i = threadIdx.x; i < part_n; i += 64) acc = fmax(acc, ld_sc1(&v.part[i]));
        acc = wave_max(acc);
        if (threadIdx.x == 0) {
            v.scal[4] = acc;
            const double chi = ld_sc1(&v.scal[6]), max_pp = ld_sc1(&v.scal[7]);
            *v.chi_cur = chi; *v.chi_loc = chi;
            if (fused) lm_begin(v, max_pp, acc, chi);
        }
    }
}

// ---- linearisation, pose side (mode 0) and trial chi2 (mode 1): SPLIT wavefronts per keyframe over slices of its
//      observations
__device__ __forceinline__ void pose_part_body(BaView& v, int bid, int robust, int mode, int set)
{
    ba_select_idx(v, set); ba_lin_set(v, set);
    double* chi_out = mode == 0 ? v.partial + (size_t)v.n_poses * SPLIT * PV : v.partial_trial;
    const int lane = threadIdx.x & 63;
    const int wv = bid * 4 + (threadIdx.x >> 6);
    const int p = wv / SPLIT, sp = wv - p * SPLIT;
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
    const bool full = mode == 0 && slot >= 0;
    for (int s = v.ps_start[p] + sp * 64 + lane; s < v.ps_start[p + 1]; s += 64 * SPLIT) {
        const int k = s;                        // observations are stored keyframe by keyframe
        if (!v.o_active[k]) continue;
        const int j = v.o_point[k];
        const double X[3] = {v.points[3 * j], v.points[3 * j + 1], v.points[3 * j + 2]};
        double e[3], pc[3], rho0;
        const int D = ba_residual(v, k, R, t, X, e, pc);
        const double w = ba_weight(v, k, D, e, robust, &rho0);
        chi += rho0;
        if (!full) continue;
        double A[3][3], B[3][6];
        ba_jacobians(v.cam, R, pc, D, A, B);
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int c = a; c < 6; ++c) {
                double s2 = 0;
#pragma unroll
                for (int r = 0; r < 3; ++r) s2 += B[r][a] * w * B[r][c];
                h[idx++] += s2;
            }
            double s3 = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) s3 += B[r][a] * (-w * e[r]);
            b[a] += s3;
        }
    }
    // partials: chi2 per (keyframe, slice) in the tail of v.partial, H_pp / b_p per (free-pose slot, slice) in its head,
    // so the combine addresses both without an index lookup
    chi = wave_sum(chi);
    if (lane == 0) st_sc1(&chi_out[(size_t)p * SPLIT + sp], chi);       // mode 1: read by the last workgroup of this launch
    if (!full) return;
    double* out = v.partial + ((size_t)slot * SPLIT + sp) * PV;
#pragma unroll
    for (int i = 0; i < 21; ++i) h[i] = wave_sum(h[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = wave_sum(b[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 21; ++i) out[i] = h[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) out[21 + i] = b[i];
    }
}

// ---- g2o's lambda control (one thread) ----------------------------------------------------------------------------------
// start of an outer iteration / after a linearisation: lambda_0 = tau * max diag(H) on the first one, chi2 bookkeeping
__device__ __forceinline__ void lm_begin(BaView& v, double max_diag_pp, double max_diag_ll, double chi_cur)
{
    BaCtl c = *v.ctl;                                      // one batch of loads, one batch of stores
    if (c.first) {
        const double maxd = fmax(v.n_points ? max_diag_ll : 0.0, max_diag_pp);
        c.lambda = 1e-5 * maxd;
        c.ni = 2;
        c.first = 0;
    }
    c.current_chi = chi_cur;
    if (c.qmax == 0) c.chi_before = chi_cur;
    c.need_lin = 0;
    c.spec = 0;
    *v.ctl = c;
}
// after a trial: rho, accept / reject, lambda update, iteration and termination bookkeeping
__device__ __forceinline__ int lm_decide(BaView& v, double temp_chi, double chol_failed, double scale_l, double scale_p, bool spec_ran = false, const BaCtl* preloaded = nullptr)      // returns "accepted"
{
    BaCtl c = preloaded ? *preloaded : *v.ctl;             // (preloaded: the caller fetched the block beside its other loads -- one round trip less on the chain)
    if (chol_failed != 0.0) temp_chi = DBL_MAX;            // factorisation failed
    double rho = c.current_chi - temp_chi;
    const double scale = (scale_l + scale_p) + 1e-3;
    rho /= scale;
    const bool accepted = rho > 0 && isfinite(temp_chi);
    if (accepted) {
        const double t = 2 * rho - 1;
        double alpha = 1. - t * t * t;
        alpha = fmin(alpha, 2. / 3.);
        const double sf = fmax(1. / 3., alpha);
        c.lambda *= sf;
        c.ni = 2;
        c.current_chi = temp_chi;
        c.cur ^= 1;                                        // discardTop: the trial state becomes the accepted one
        c.spec = spec_ran ? 1 : 0;                         // ... and its linearisation is already in its set
    } else {
        c.lambda *= c.ni;
        c.ni *= 2;                                         // pop: the accepted state stays
    }
    c.rho = rho;
    c.last_accepted = accepted ? 1 : 0;
    c.qmax++;
    const bool finished = !(rho < 0 && c.qmax < 10);
    if (finished) {
        const int terminate = (c.qmax == 10 || rho == 0) ? 1 : 0;
        if (c.outer_done < MAX_LOG) {
            lpslam_hip_ba_iter_log* l = v.log + c.outer_done;
            l->chi2_before = c.chi_before; l->chi2_after = c.current_chi; l->lambda = c.lambda; l->trials = c.qmax; l->status = terminate;
        }
        c.outer_done++;
        c.qmax = 0;
        if (accepted && spec_ran) {
            // the trial launch linearised the accepted state completely (W, landmark and pose sums, combined by its last workgroup):
            // the next outer iteration starts at the Schur complement.  What lm_begin would do: chi2 bookkeeping of a fresh iteration
            c.need_lin = 0; c.spec = 0; c.chi_before = c.current_chi;
        } else c.need_lin = 1;
        if (terminate) c.stopped = 1;
    } else {
        c.need_lin = 0;
    }
    *v.ctl = c;
    return c.last_accepted;
}

// ---- after the linearisation (mode 0) and the trial chi2 (mode 1): one workgroup (the last one of the pass that produced
//      the partials, see ba_last_block) combines the SPLIT partials in order, totals
//      chi2, reduces the landmark-side block partials in v.part (max diag H_ll / scale terms) and, in the single-GPU
//      ("fused") solve, runs the lambda control.  The partitioned solve runs k_lm_begin / k_lm_decide after its all-reduce.
__device__ __forceinline__ void pose_combine_body(BaView& v, int mode, int part_n, int fused)
{
    __shared__ double s_val[3];            // chi2, landmark-side term (max diag H_ll / scale term), max diag H_pp
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wave == 0) {                       // chi2 = sum over keyframes (64-strided, then butterfly) of the in-order sum of its SPLIT partials
        double acc = 0;
        for (int p = lane; p < v.n_poses; p += 64) {
            double s = 0;
            const double* chi_in = mode == 0 ? v.partial + (size_t)v.n_poses * SPLIT * PV : v.partial_trial;
            for (int sp = 0; sp < SPLIT; ++sp) s += mode == 1 ? ld_sc1(&chi_in[(size_t)p * SPLIT + sp]) : chi_in[(size_t)p * SPLIT + sp];
            v.chi_pose[p] = s;
            acc += s;
        }
        acc = wave_sum(acc);
        if (lane == 0) s_val[0] = acc;
    } else if (wave == 1 && mode == 1) {
        double acc = 0;
        for (int i = lane; i < part_n; i += 64) acc += v.part[i];
        acc = wave_sum(acc);
        if (lane == 0) s_val[1] = acc;
    } else if (wave == 2 && mode == 0) {   // max |diag H_pp| straight from the partials (same sums as the loop below)
        double m = 0;
        for (int i = lane; i < v.dim; i += 64) {
            const int slot = i / 6, a = i - 6 * slot;
            const int q = 6 * a - a * (a - 1) / 2;          // (a, a) in the row-major upper triangle
            double s = 0;
            for (int sp = 0; sp < SPLIT; ++sp) s += v.partial[((size_t)slot * SPLIT + sp) * PV + q];
            m = fmax(m, fabs(s));
        }
        m = wave_max(m);
        if (lane == 0) s_val[2] = m;
    }
    if (mode == 0) {
        // H_pp (21) and b_p (6) per free pose: in-order sums of the SPLIT partials; three items per thread and pass so that
        // their 24 loads are in flight together
        const int items = v.n_free * 27;
        for (int i0 = tid; i0 < items; i0 += 3 * 256) {
            double sum[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = i0 + u * 256;
                sum[u] = 0;
                if (i < items) {
                    const int slot = i / 27, q = i - slot * 27;
                    for (int sp = 0; sp < SPLIT; ++sp) sum[u] += v.partial[((size_t)slot * SPLIT + sp) * PV + q];
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = i0 + u * 256;
                if (i >= items) continue;
                const int slot = i / 27, q = i - slot * 27;
                const double sv = sum[u];
                if (q >= 21) { v.bp[6 * slot + (q - 21)] = sv; v.bp_loc[6 * slot + (q - 21)] = sv; }
                else {
                    int a = 0, rem = q;                    // upper-triangle index q -> (a, c)
                    while (rem >= 6 - a) { rem -= 6 - a; ++a; }
                    const int c = a + rem;
                    double* H = v.Hpp + 36 * (size_t)slot;
                    H[a * 6 + c] = sv; H[c * 6 + a] = sv;
                    if (a == c) { v.hppdiag[6 * slot + a] = sv; v.hppdiag_loc[6 * slot + a] = sv; }
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (mode == 0) {
            st_sc1(&v.scal[6], s_val[0]); st_sc1(&v.scal[7], s_val[2]);        // chi2 and max diag H_pp for the last workgroup of k_ba_point_sum
        } else {
            const double fail = v.scal[5], scale_p = v.scal[3];
            v.scal[1] = s_val[0]; v.scal[2] = s_val[1];
            if (fused) { lm_decide(v, s_val[0], fail, s_val[1], scale_p, fused == 2); ba_sync_words(v)[2] = 0; }      // ([2]: groups of the next Schur launch that have stored their share, ba_band.inl)
        }
    }
}

// ---- linearisation 1/2: the observation side (blocks [0, obs_blocks)) and the pose side (the rest) of the accepted state in one
//      launch: both only read the state, so they run side by side.  Skipped when the previous trial launch already linearised
//      this state on speculation (ctl->spec).
__global__ __launch_bounds__(256) void k_ba_lin(const BaView* __restrict__ views, int robust, int points_fixed)
{
    BA_VIEW(v);
    BA_VIEW_HEAD("s"(v.obs_blocks), "s"(v.pose_blocks), "s"(v.ctl));
    const int obs_blocks = v.obs_blocks;
    if ((int)blockIdx.x >= obs_blocks + v.pose_blocks) return;
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle() || !fl.need_lin || fl.spec) return;
    const int idx = fl.cur;
    if ((int)blockIdx.x < obs_blocks) obs_lin_body(v, blockIdx.x, robust, points_fixed, idx);
    else pose_part_body(v, (int)blockIdx.x - obs_blocks, robust, 0, idx);
}

// ---- chi2 of the trial state per keyframe (blocks [0, trial_blocks)); the last of them totals it and (fused) runs the lambda
//      control.  With spec != 0 the rest of the grid linearises the TRIAL state into the other linearisation set beside it: a
//      trial is accepted far more often than not, and then the next iteration starts with its linearisation done (the sets are
//      double buffered like the states, a rejected trial leaves the accepted state's set untouched).  These workgroups read
//      ctl->cur_launch, not ctl->cur, which the decision may flip while they run.
__global__ __launch_bounds__(256) void k_ba_trial(const BaView* __restrict__ views, int robust, int fused, int points_fixed, int spec, int skip_one_pass)
{
    BA_VIEW(v);
    BA_VIEW_HEAD("s"(v.part_n), "s"(v.pose_blocks), "s"(v.land_blocks), "s"(v.ctl));
    const int part_n = v.part_n, trial_blocks = v.pose_blocks, land_blocks = v.land_blocks;
    if ((int)blockIdx.x >= trial_blocks + (spec ? land_blocks + trial_blocks : 0)) return;
    if (skip_one_pass && upd_takes(v.n_points, v.n_free, v.n_poses)) return;      // that problem's trial ran in k_ba_update
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle()) return;
    const int idx = fl.cur_launch ^ 1;
    int combine = -1;
    if ((int)blockIdx.x >= trial_blocks) {
        // the complete linearisation of the trial state into the other set: observations landmark-major with the landmark sums
        // (land_lin_body), the pose side per keyframe and slice (its partial sums are added up by their readers in k_ba_schur)
        const int bid = (int)blockIdx.x - trial_blocks;
        if (bid < land_blocks) land_lin_body(v, bid, robust, points_fixed, idx);
        else pose_part_body(v, bid - land_blocks, robust, 0, idx);
        return;                          // read by the next launch only (k_ba_schur adds the pose partials up itself): no hand-over
    } else {
        pose_part_body(v, blockIdx.x, robust, 1, idx);
        if (ba_last_block_sc1(v.ctl, trial_blocks)) combine = 1;
    }
    if (combine < 0) return;
    pose_combine_body(v, 1, part_n, fused ? (spec ? 2 : 1) : 0);
}

// partitioned solve: lambda control on the all-reduced quantities
__global__ __launch_bounds__(64) void k_lm_begin(const BaView* __restrict__ views)
{
    BA_VIEW(v);
    if (ba_idle(v.ctl) || !v.ctl->need_lin) return;
    double m = 0;
    for (int i = threadIdx.x; i < v.dim; i += 64) m = fmax(m, fabs(v.hppdiag[i]));
    m = wave_max(m);
    if (threadIdx.x == 0) lm_begin(v, m, v.scal[4], *v.chi_cur);
}
__global__ void k_lm_decide(const BaView* __restrict__ views)
{
    BA_VIEW(v);
    if (ba_idle(v.ctl)) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) lm_decide(v, v.scal[1], v.scal[5], v.scal[2], v.scal[3]);
}

// Y = W (H_ll + lambda I)^-1 of one observation (6x3 times symmetric 3x3), in the operation order every user shares
__device__ __forceinline__ void obs_y_row(const double* h, double w0, double w1, double w2, double& y0, double& y1, double& y2)
{
    y0 = w0 * h[0] + w1 * h[1] + w2 * h[2];
    y1 = w0 * h[1] + w1 * h[3] + w2 * h[4];
    y2 = w0 * h[2] + w1 * h[4] + w2 * h[5];
}

// ---- per trial: Schur complement; Y = W (H_ll + lambda I)^-1 is formed per term from W and the landmark's 3x3 block (a launch
//      and the 18 doubles per observation it wrote and this kernel read back are gone).  One wavefront per pose-block pair (i <= k)
//      or part of one, over the pair's term list (observation of i, observation of k, landmark), 32 terms per round, two lanes per
//      term with 18 accumulators each; partials are summed in lane order through LDS (fixed summation order).
//      Round 6, in-kernel stamps (tools/dev_schur_stamps.py): 3675 of the 5349 workgroups of the first form were surplus parts that
//      left at once, and dispatching them kept the last real ones (the rhs blocks) waiting for 8-12 us of a 24 us launch; a workgroup's
//      own chain was five dependent round trips before its first row and then a round trip per round.  Now the grid holds the parts
//      that exist (table built with the lists, ba_build.inl), a workgroup's item record and the control flags come in one round trip,
//      and a round's 64 rows (144 bytes each) are fetched by the wavefront TOGETHER, nine lanes to a row, sixteen bytes a lane, a round
//      AHEAD of their use (registers -> LDS -> the lanes of the term).  What bounds a workgroup since is its instruction count: ~115
//      FP64 instructions (4 cycles each) + ~100 others per round of 32 terms, two wavefronts to a SIMD (1.3 us per round measured; with
//      every load hitting the L1 still 0.9).  Measured and dropped: four lanes per term with two or three rounds in flight (128- and
//      64-thread workgroups; 28.5 / 31 us: more instructions per term, or more wavefronts than the chip holds), eight parts per list.
//      The rhs of keyframe i, b_p,i - sum Y b_l over its observations, rides on the diagonal block (i, i), whose list has a term per
//      observation: no workgroups of its own and no second pass over W.
//      fused != 0 (single-GPU solve): lambda goes onto the pose diagonal, rhs straight into row `dim` of S and the failure
//      flag / rhs pivot are reset here, so no separate preparation launch is needed.
#ifdef LPSLAM_SCHUR_STAMPS
// development: (start, end, wait start) of every workgroup of the last k_ba_schur launch, 100 MHz clock (tools/dev_schur_stamps.py)
__device__ unsigned long long g_schur_stamps[8 * 8192];
#define SCHUR_STAMP(k) do { if (threadIdx.x == 0 && bx0 < 8192) g_schur_stamps[8 * bx0 + (k)] = wall_clock64(); } while (0)
#else
#define SCHUR_STAMP(k) do {} while (0)
#endif
typedef double dbl2 __attribute__((ext_vector_type(2)));      // (a plain vector type: registers, where the HIP class type ended up in scratch memory)
// a round's rows, landmark blocks and rhs vectors -> registers (term records in `ab`: lane l holds term l & 31 of the round).
// Straight-line code: every lane loads in every slot (the last slots of the landmark blocks fetch a duplicate that is never stored) --
// with a condition around a load the compiler ends each one with a full wait, the eleven lane exchanges and loads ran one after the
// other and a round cost 0.9 us of issue alone.
__device__ __forceinline__ void schur_fetch(const BaView& v, const int4& ab, int lane, bool diag, dbl2 (&wq)[9], dbl2 (&hq)[2], double (&bq)[2], int& same)
{
    const int mine = lane < 32 ? ab.x : ab.y;
    int o[9], lm[2];
#pragma unroll
    for (int j = 0; j < 9; ++j) o[j] = __shfl(mine, (j * 64 + lane) / 9);
#pragma unroll
    for (int j = 0; j < 2; ++j) lm[j] = __shfl(ab.z, min((j * 64 + lane) / 3, 31));
    same = __shfl((int)(ab.x == ab.y), lane >> 1);
    // (byte offsets in 32 bits: base in scalar registers + one vector offset per load instead of a 64-bit multiply-add each)
    GPTR(const char) Wc = reinterpret_cast<GPTR(const char)>(v.W);
    GPTR(const char) Hc = reinterpret_cast<GPTR(const char)>(v.Hll);
    GPTR(const char) Bc = reinterpret_cast<GPTR(const char)>(v.bl);
#pragma unroll
    for (int j = 0; j < 9; ++j) { const int idx = j * 64 + lane, u = idx - 9 * (idx / 9); wq[j] = *reinterpret_cast<GPTR(const dbl2)>(Wc + (144u * (unsigned)o[j] + 16u * (unsigned)u)); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int idx = j * 64 + lane, u = idx - 3 * (idx / 3); hq[j] = *reinterpret_cast<GPTR(const dbl2)>(Hc + (48u * (unsigned)lm[j] + 16u * (unsigned)u)); }
    if (diag) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int idx = j * 64 + lane, u = idx - 3 * (idx / 3); bq[j] = *reinterpret_cast<GPTR(const double)>(Bc + (24u * (unsigned)lm[j] + 8u * (unsigned)u)); }
    }
}
template <int UPD_PB> __device__ __forceinline__ void ba_pose_side_wave(BaView& v, int p, int sp, int robust, int cur, bool publish);
__device__ __forceinline__ void ba_pose_side_wait(const BaView& v);
__global__ __launch_bounds__(64) void k_ba_schur(const BaView* __restrict__ views, int fused, int robust)
{
    BA_VIEW_XCD(v, bx0);
    // workgroups: [pose side of an accepted state's linearisation (ba_update.inl) | further parts, first stretch | part 0 of every
    // block | further parts, the rest]
    const int lead = v.n_poses * SPLIT;
    const int nblk = v.n_blocks, ecap = v.extra_pack >> 12, efirst = v.extra_pack & 4095;
    if (bx0 >= lead + ecap + nblk || v.band_hbw >= 0) return;      // banded windows: k_schur_group / k_schur_band_reduce (ba_band.inl)
    // the item's record and the control block's flags in ONE round trip (both addresses follow from the view alone)
    const int bx = bx0 - lead;
    const bool further = bx >= 0 && (bx < efirst || bx >= efirst + nblk);
    GPTR(const int) ex = v.blk_ticket + nblk;              // [items | block * SCH_MAXP + part ...], written with the lists and constant since
    const int e = bx < efirst ? bx : bx - nblk;
    const int count = ex[0];
    const int item = bx < 0 ? 0 : (further ? ex[1 + min(max(e, 0), max(ecap - 1, 0))] : SCH_MAXP * v.blk_perm[bx - efirst]);
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle()) return;
    const int pending = fused ? ba_sync_words(v)[3] : 0;   // raised by k_ba_update's decision: H_pp, b_p of state `cur` are this launch's to compute
    SCHUR_STAMP(0); SCHUR_STAMP(2);
    if (bx < 0) { if (pending) ba_pose_side_wave<2>(v, bx0 / SPLIT, bx0 % SPLIT, robust, fl.cur, true); SCHUR_STAMP(1); return; }
    if (further && e >= count) return;
    const int blk = item / SCH_MAXP, part_id = item % SCH_MAXP;
    const double lambda = fl.lambda;
    ba_lin_set(v, fl.cur);
    const int lane = threadIdx.x;
    const int n = v.dim_pad;
    const int t_begin = v.blk_start[blk], t_end = v.blk_start[blk + 1];
    const int parts = schur_parts(t_end - t_begin);
    SCHUR_STAMP(3);
    // block -> (i, k), i <= k, blocks numbered row by row: row i starts at i N - i (i - 1) / 2
    int i;
    {
        const int N = v.n_free;
        const float h = (float)(2 * N + 1);
        i = (int)((h - sqrtf(fmaxf(h * h - 8.0f * (float)blk, 0.0f))) * 0.5f);
        i = max(0, min(i, N - 1));
        while (i > 0 && i * N - i * (i - 1) / 2 > blk) --i;
        while (i + 1 < N && (i + 1) * N - (i + 1) * i / 2 <= blk) ++i;
    }
    const int k = i + (blk - (i * v.n_free - i * (i - 1) / 2));
    const bool diag = i == k;
    // LDS: a round's rows [64][18] (0 .. 31 the terms' rows of keyframe i, 32 .. 63 of keyframe k), landmark blocks [32][6] and rhs
    // vectors [32][3]; afterwards the lanes' partial sums [32][43]
    __shared__ __attribute__((aligned(16))) double s_buf[64 * 18 + 2 * 128 + 128];      // (every lane stores in every slot: the blocks' and vectors' sections are rounded up to 128 lanes)
    double* s_h = s_buf + 64 * 18;
    double* s_b = s_h + 2 * 128;
    const int tl = lane >> 1, hp = lane & 1;
    double acc[18], r3[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < 18; ++q) acc[q] = 0;
    const int stride = 32 * parts;
    int t0 = t_begin + part_id * 32;
    // registers of the round being fetched
    dbl2 wq[9], hq[2];
    double bq[2] = {0, 0};
    int4 ab = make_int4(0, 0, 0, 0), ab_next = make_int4(0, 0, 0, 0);
    int same_next = 0;
    if (t0 < t_end) {
        ab = v.blk_terms[min(t0 + (lane & 31), t_end - 1)];
        if (t0 + stride < t_end) ab_next = v.blk_terms[min(t0 + stride + (lane & 31), t_end - 1)];
        schur_fetch(v, ab, lane, diag, wq, hq, bq, same_next);
    }
    SCHUR_STAMP(4);
    bool first_round = true;
    while (t0 < t_end) {
#pragma unroll
        for (int j = 0; j < 9; ++j) *reinterpret_cast<dbl2*>(s_buf + 2 * (j * 64 + lane)) = wq[j];
        *reinterpret_cast<dbl2*>(s_h + 2 * lane) = hq[0];
        *reinterpret_cast<dbl2*>(s_h + 2 * (64 + lane)) = hq[1];
        if (diag) { s_b[lane] = bq[0]; s_b[64 + lane] = bq[1]; }
        const bool valid = t0 + tl < t_end;
        const bool same = same_next != 0;
        __syncthreads();
        if (first_round) { SCHUR_STAMP(5); first_round = false; }
        t0 += stride;
        if (t0 < t_end) {
            ab = ab_next;
            schur_fetch(v, ab, lane, diag, wq, hq, bq, same_next);
            if (t0 + stride < t_end) ab_next = v.blk_terms[min(t0 + stride + (lane & 31), t_end - 1)];
        }
        if (valid) {
            double hraw[6], y[9], w[18];
            const double* ya = s_buf + 18 * tl + 9 * hp;
            const dbl2* wb = reinterpret_cast<const dbl2*>(s_buf + 18 * (32 + tl));
#pragma unroll
            for (int q = 0; q < 6; ++q) hraw[q] = s_h[6 * tl + q];
#pragma unroll
            for (int q = 0; q < 9; ++q) y[q] = ya[q];
#pragma unroll
            for (int q = 0; q < 9; ++q) { const dbl2 b2 = wb[q]; w[2 * q] = b2.x; w[2 * q + 1] = b2.y; }
            double h[6];
            point_hinv(hraw, lambda, h);
#pragma unroll
            for (int r = 0; r < 3; ++r) {                  // Y = W (H_ll + lambda I)^-1, row r of this lane's half
                const double w0 = y[r * 3], w1 = y[r * 3 + 1], w2 = y[r * 3 + 2];
                y[r * 3] = __builtin_fma(w2, h[2], __builtin_fma(w1, h[1], w0 * h[0]));
                y[r * 3 + 1] = __builtin_fma(w2, h[4], __builtin_fma(w1, h[3], w0 * h[1]));
                y[r * 3 + 2] = __builtin_fma(w2, h[5], __builtin_fma(w1, h[4], w0 * h[2]));
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    acc[r * 6 + c] = __builtin_fma(y[r * 3 + 2], w[c * 3 + 2], __builtin_fma(y[r * 3 + 1], w[c * 3 + 1], __builtin_fma(y[r * 3], w[c * 3], acc[r * 6 + c])));      // (fused: the kernel is bound by its vector instructions, 4 cycles each)
            if (diag && same) {                            // the observation's share of the keyframe's rhs (a keyframe that sees a landmark twice has cross terms in its list: not those)
                const double b0 = s_b[3 * tl], b1 = s_b[3 * tl + 1], b2 = s_b[3 * tl + 2];
#pragma unroll
                for (int r = 0; r < 3; ++r) r3[r] = __builtin_fma(y[r * 3 + 2], b2, __builtin_fma(y[r * 3 + 1], b1, __builtin_fma(y[r * 3], b0, r3[r])));
            }
        }
        __syncthreads();
    }
    SCHUR_STAMP(6);
    // the lanes' sums in lane order: per term-lane pair [row half 0: 18 + 3 | row half 1: 18 + 3], stride 43
    double* part = s_buf;
#pragma unroll
    for (int q = 0; q < 18; ++q) part[tl * 43 + 21 * hp + q] = acc[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) part[tl * 43 + 21 * hp + 18 + q] = r3[q];
    __syncthreads();
    // lanes 0 .. 35: entry (r, c) of the 6 x 6 sum; lanes 36 .. 41: row lane - 36 of the keyframe's rhs sum (diagonal blocks)
    const int er = lane < 36 ? lane / 6 : lane - 36, ec = lane < 36 ? lane - 6 * er : 0;
    const int slot = 21 * (er / 3) + (lane < 36 ? 6 * (er % 3) + ec : 18 + er % 3);
    double sum = 0;
    if (lane < SCH_PV) for (int l = 0; l < 32; ++l) sum += part[l * 43 + slot];
    SCHUR_STAMP(7);
    if (parts > 1) {
        // hand-over without cache maintenance: the partial sums are stored write-through (sc1) and read back L1-bypassing (sc1),
        // the ticket is a relaxed agent-scope add made after this (single) wavefront's stores have drained -- no buffer_wbl2 /
        // buffer_inv, which cost more than the part they guard (MI355X_MICROARCH: valid forms, one unsharded counter)
        double* mine = v.blk_part + (size_t)(SCH_MAXP * blk + part_id) * SCH_PV;
        if (lane < SCH_PV) __hip_atomic_store(&mine[lane], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const int tk = __hip_atomic_fetch_add(&v.blk_ticket[blk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (tk == parts - 1);
            if (s_last) v.blk_ticket[blk] = 0;
        }
        __syncthreads();
        if (!s_last) { SCHUR_STAMP(1); return; }
        sum = 0;
        if (lane < SCH_PV) for (int p = 0; p < parts; ++p) sum += __hip_atomic_load(&v.blk_part[(size_t)(SCH_MAXP * blk + p) * SCH_PV + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    SCHUR_STAMP(2);
    if (diag && pending) ba_pose_side_wait(v);             // the leading workgroups of this launch have stored the pose side (write-through)
    SCHUR_STAMP(1);
    if (lane < 36) {
        const int r = er, c = ec;
        if (diag) {
            const int ra = min(r, c), rc = max(r, c);
            const int q = 6 * ra - ra * (ra - 1) / 2 + (rc - ra);      // H_pp(r, c) in the row-major upper triangle of the pose partials
            double hpp = 0;
            for (int sp = 0; sp < SPLIT; ++sp) { const double* pr = v.partial + ((size_t)i * SPLIT + sp) * PV + q; hpp += pending ? ld_sc1(pr) : *pr; }
            double val = hpp - sum;
            if (fused && r == c) val += lambda;
            v.S[(size_t)(6 * i + r) * n + 6 * i + c] = val;
        } else {
            v.S[(size_t)(6 * i + r) * n + 6 * k + c] = -sum;
            v.S[(size_t)(6 * k + c) * n + 6 * i + r] = -sum;
        }
    } else if (diag && lane < SCH_PV) {
        // b_p and diag H_pp of keyframe i: the SPLIT partial sums of the set's pose-side linearisation, added in order (what
        // pose_combine_body does after an explicit linearisation; a linearisation made beside a trial has no combine of its own)
        const int q = er;
        const int qd = 6 * q - q * (q - 1) / 2;              // (q, q) in the row-major upper triangle
        double bsum = 0, dsum = 0;
        for (int sp = 0; sp < SPLIT; ++sp) { const double* pr = v.partial + ((size_t)i * SPLIT + sp) * PV; bsum += pending ? ld_sc1(pr + 21 + q) : pr[21 + q]; dsum += pending ? ld_sc1(pr + qd) : pr[qd]; }
        const double val = bsum - sum;
        v.rhs[6 * i + q] = val;
        if (fused) v.S[(size_t)v.dim * n + 6 * i + q] = val;
        v.bp[6 * i + q] = bsum; v.hppdiag[6 * i + q] = dsum;      // the solved set's sums (partitioned: fresh partials for the all-reduce)
    } else if (diag && i == 0) {
        if (!fused && lane == 62) *v.chi_cur = *v.chi_loc;
        if (fused && lane == 63) { v.S[(size_t)v.dim * n + v.dim] = 1e200; v.scal[5] = 0.0; }
    }
}

// preparation for the partitioned (all-reduced) solve: lambda on the diagonal, rhs row, flag / pivot reset
__global__ __launch_bounds__(256) void k_chol_prep(const BaView* __restrict__ views)
{
    BA_VIEW(v);
    if (ba_idle(v.ctl)) return;
    const double lambda = v.ctl->lambda;
    const int n = v.dim_pad, dim = v.dim;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { v.scal[5] = 0.0; v.S[(size_t)dim * n + dim] = 1e200; }
    if (i < dim) { v.S[(size_t)i * n + i] += lambda; v.S[(size_t)dim * n + i] = v.rhs[i]; }
    else if (i < n && i > dim) v.S[(size_t)i * n + i] = 1.0;      // the all-reduce summed the identity padding
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


// ---- blocked Cholesky of S (+ lambda I), 32-wide panel columns, two per launch ------------------------------------------------
//   Two kinds of extra rows ride along so that no triangular substitution is ever run:
//     row `dim` of S carries the rhs            -> after the last panel it holds y = L^-1 rhs,
//     Minv starts as the identity (synthesised) -> its row blocks become L^-T; block (e, .) only exists from column e on.
//   A panel is factored by ONE wavefront in two 16-column strips (chol_panel_dpp below): the diagonal block replicated in every
//   16-lane DPP row, the rows that ride along one per lane; the same rank-1 updates factor D and solve B L_jj^T = B.  The
//   diagonal block is factored redundantly by every panel workgroup, which removes every dependence between workgroups of a launch.
//   All working workgroups run on one XCD: the grid is launched 8x oversized and only every 8th workgroup works (workgroups go
//   round-robin to the 8 XCDs), so the panel written by one launch is an L2 hit for the next (measured: -1 us per launch).
// Explicit FMAs are used here (the contraction pragma only governs implicit fusing); the factorisation is not a parity
// quantity, the solve's effect on chi2 / poses is (tolerances in DESIGN.md).
// Launch m factors panel columns j = 2m and j1 = 2m + 1 and applies the previous pair (kb0 = 2m - 2, kb1 = 2m - 1) to the rest of
// the trailing matrix: one kernel boundary per two panels on the critical path.  A panel workgroup (row block i, or
// Minv row block e) keeps its five blocks  D_j, X = A_j1,j, D_j1, B0 = A_i,j, B1 = A_i,j1  in LDS:
//   1. lookahead with the previous pair (K = 64, matrix cores): D_j, X, B0 on all four wavefronts;
//   2. wavefront 0 factors [D_j; X] -> L_jj, L_j1,j and wavefront 1 [D_j; B0] -> L_i,j in registers, while
//      wavefronts 2 and 3 finish the lookahead of D_j1 and B1;
//   3. D_j1 -= L_j1,j L_j1,j^T, B1 -= L_i,j L_j1,j^T (matrix cores);
//   4. wavefront 0 factors [D_j1; B1] -> L_j1,j1, L_i,j1.
// D_j, X and D_j1 are factored redundantly by every panel workgroup; nobody overwrites them in S during the launch (the
// diagonal workgroup publishes L_jj, L_j1,j1 into Ldiag; L_j1,j is needed by no later launch).  With an odd number of panels
// the last launch is `single`: only column j.
constexpr int CP_BLK = NB * (NB + 1);                  // one padded 32x32 block in LDS
constexpr int CP_LDS_BYTES = (11 * CP_BLK + 2 * (64 * 17 + 64 * 17)) * (int)sizeof(double);      // 11 blocks + the factor scratch of two wavefronts

constexpr int CH_SCR = 64 * 17 + 64 * 17;              // doubles of scratch per factoring wavefront: L1 [64][17] and U [64][17]
// ---- 16-column strip of a panel, rank-1 multipliers by DPP ---------------------------------------------------------------
// The diagonal block D (16 x 16) is held REPLICATED: lane l keeps row l & 15 in d[0..15], so every 16-lane DPP row owns a full
// copy, and x[0..15] is the lane's own row of whatever rides along below D (64 rows per wavefront).  The rank-1 update of column
// jj, a[c] -= l[c] * l[row], then is ONE instruction per column and register set: v_fmac_f64_dpp with row_newbcast:c fetches
// l[c] from lane c of the lane's own DPP row (the DP ALU only knows this DPP control, gfx90a+).  The readlane form it replaces
// costs three issue slots per column (two v_readlane_b32 + the fma) and keeps the multipliers in SGPRs, which spilled; on
// MI355X a strip with 64 riding rows takes 2.4k cycles against 3.4k (micro-benchmark) and 7.4k inside k_chol_pair.
// Hazard: a VALU write of the DPP source needs two wait states before the DPP read and the compiler does not look into inline
// assembly, so every block starts with s_nop 1.
#define LP_DPPF(c) "v_fmac_f64_dpp %" #c ", %16, -%17 row_newbcast:" #c " row_mask:0xf bank_mask:0xf\n\t"
#define LP_F15 LP_DPPF(15)
#define LP_F14 LP_DPPF(14) LP_F15
#define LP_F13 LP_DPPF(13) LP_F14
#define LP_F12 LP_DPPF(12) LP_F13
#define LP_F11 LP_DPPF(11) LP_F12
#define LP_F10 LP_DPPF(10) LP_F11
#define LP_F9 LP_DPPF(9) LP_F10
#define LP_F8 LP_DPPF(8) LP_F9
#define LP_F7 LP_DPPF(7) LP_F8
#define LP_F6 LP_DPPF(6) LP_F7
#define LP_F5 LP_DPPF(5) LP_F6
#define LP_F4 LP_DPPF(4) LP_F5
#define LP_F3 LP_DPPF(3) LP_F4
#define LP_F2 LP_DPPF(2) LP_F3
#define LP_F1 LP_DPPF(1) LP_F2
#define LP_ACC16(a) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
// a[c] -= (src of lane c of this DPP row) * mul   for c = FROM .. 15
template <int FROM>
__device__ __forceinline__ void dpp_rank1(double (&a)[16], double src, double mul)
{
#define LP_CASE(k, S) if constexpr (FROM == k) asm("s_nop 1\n\t" S : LP_ACC16(a) : "v"(src), "v"(mul));
    LP_CASE(1, LP_F1) LP_CASE(2, LP_F2) LP_CASE(3, LP_F3) LP_CASE(4, LP_F4) LP_CASE(5, LP_F5) LP_CASE(6, LP_F6) LP_CASE(7, LP_F7) LP_CASE(8, LP_F8)
    LP_CASE(9, LP_F9) LP_CASE(10, LP_F10) LP_CASE(11, LP_F11) LP_CASE(12, LP_F12) LP_CASE(13, LP_F13) LP_CASE(14, LP_F14) LP_CASE(15, LP_F15)
#undef LP_CASE
}
template <int L>
__device__ __forceinline__ double dpp_bcast(double v)          // the value of lane L of this lane's DPP row
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
    return r;
}
// One pivot of a strip.  The strip is bound by its instruction COUNT (measured, tools/dev/strip_bench2.hip: 240 DPP updates at
// 5.5 cycles + the per-pivot instructions at 4 to 5 cycles each; placing the next pivot's root between the updates by hand changes
// nothing), so a pivot is written with as few instructions as the arithmetic allows:
//   * 1 / sqrt(d) = y0 (1 + e / 2 + 3 e^2 / 8), e = 1 - d y0^2, y0 = v_rsq_f64 (2^-24, measured): one cubic step, five operations,
//     1.4e-16 relative error over 4M samples (two Goldschmidt steps: eight operations, 2.1e-16);
//   * the positivity guard sits BEFORE the root (compare + one 64-bit select; a failed pivot continues on 1.0: harmless finite
//     numbers, the factorisation is flagged and its result discarded);
//   * the next pivot is read back from the updated column with one DPP broadcast (a recurrence a(jj+1,jj+1) - l(jj+1)^2 on values
//     broadcast beforehand shortens the dependent chain but costs four more instructions: 3.15 k against 2.84 k cycles per strip).
template <int JJ>
__device__ __forceinline__ void strip_step(double (&d)[16], double (&x)[16], double piv, bool& fail)
{
    const bool ok = piv > 0.0;
    fail |= !ok;
    const double pg = ok ? piv : 1.0;
    const double y0 = __builtin_amdgcn_rsq(pg);
    const double t = pg * y0;
    const double e = fma(-t, y0, 1.0);
    const double pp = fma(0.375, e, 0.5), ye = y0 * e;
    const double rs = fma(ye, pp, y0);
    const double l = d[JJ] * rs, lx = x[JJ] * rs;
    d[JJ] = l; x[JJ] = lx;
    if constexpr (JJ < 15) {
        dpp_rank1<JJ + 1>(d, l, l);
        const double next = dpp_bcast<JJ + 1>(d[JJ + 1]);
        dpp_rank1<JJ + 1>(x, l, lx);
        strip_step<JJ + 1>(d, x, next, fail);
    }
}
// d <- chol(D) (lower part; rows above the diagonal of a column hold values nobody reads), x <- x chol(D)^-T
__device__ __forceinline__ bool strip_factor(double (&d)[16], double (&x)[16])
{
    bool fail = false;
    strip_step<0>(d, x, dpp_bcast<0>(d[0]), fail);
    return fail;
}
// Register Cholesky of a 32-column panel [D; B] from two padded LDS blocks (B may be absent): two strips with the block product
// between them on the matrix cores (through this wavefront's scratch).  Lane l carries row l & 15 of the replicated diagonal
// block of each strip, and in x / y its own row of the stack  [D rows 16..31 (lanes 0-15); B rows 0..31 (lanes 16-47)]:
//   dA = L00, x = [L10; L_B,0], dB = L11, y = [L11; L_B,1]  (lanes 48-63 idle along with zeros)
struct PanelRegs { double dA[16], x[16], dB[16], y[16]; };
// the two strips of a loaded panel with the block product between them; ROWS: lanes that carry rows (64, or 48 where the
// scratch is short: 2 * ROWS * 17 doubles)
template <int ROWS>
__device__ __forceinline__ bool chol_panel_core(PanelRegs& p, int lane, double* scr)
{
    const int r = lane & 15;
    bool fail = strip_factor(p.dA, p.x);
    double* Ls = scr; double* Us = scr + ROWS * 17;
    if (ROWS == 64 || lane < ROWS) {
#pragma unroll
        for (int k = 0; k < 16; ++k) Ls[lane * 17 + k] = p.x[k];
    }
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int t = 0; t < 3; ++t) {                        // U[16t.., :] = X[16t.., :] (X[0..15, :])^T, K = 16 (rows 48.. are zero)
        f64x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int s4 = 0; s4 < 16; s4 += 4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[(16 * t + lr) * 17 + s4 + lk], Ls[lr * 17 + s4 + lk], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) Us[(16 * t + lk + 4 * q) * 17 + lr] = acc[q];
    }
    if (lane < 48) {
#pragma unroll
        for (int c = 0; c < 16; ++c) p.y[c] -= Us[lane * 17 + c];
    }
    // the second diagonal block, updated, goes back out to every DPP row
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) Ls[lane * 17 + c] = p.y[c];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) p.dB[c] = Ls[r * 17 + c];
    fail |= strip_factor(p.dB, p.y);
    return fail;
}
__device__ __forceinline__ bool chol_panel_dpp(PanelRegs& p, const double* Dblk, const double* Bblk, int lane, double* scr)
{
    const int r = lane & 15;
    const double* own = lane < 16 ? Dblk + (16 + lane) * (NB + 1) : (Bblk != nullptr && lane < 48 ? Bblk + (lane - 16) * (NB + 1) : nullptr);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        p.dA[c] = Dblk[r * (NB + 1) + c];
        p.x[c] = own ? own[c] : 0.0;
        p.y[c] = own ? own[16 + c] : 0.0;
    }
    return chol_panel_core<64>(p, lane, scr);
}
// acc += A[tr.., :] B[tc.., :]^T over one 32-wide k-block (16x16 tile, 8 x v_mfma_f64_16x16x4)
__device__ __forceinline__ f64x4 mfma_tile32(const double* A, const double* B, int tr, int tc, int lr, int lk, f64x4 acc)
{
#pragma unroll
    for (int s4 = 0; s4 < NB; s4 += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(tr + lr) * (NB + 1) + s4 + lk], B[(tc + lr) * (NB + 1) + s4 + lk], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ void tile_sub(double* D, int tr, int tc, int lr, int lk, f64x4 acc)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) D[(tr + lk + 4 * q) * (NB + 1) + tc + lr] -= acc[q];
}

__global__ __launch_bounds__(256) void k_chol_pair(const BaView* __restrict__ views, int m, int pin, int skip_small)
{
    if (pin && (blockIdx.x & 7)) return;                 // small launches: XCD 0 only (see above); large ones use the whole chip
    // Everything this kernel reads of its view, fetched in ONE round trip: left to itself the compiler loads each member where it
    // is first used, and the prologue becomes a chain of dependent round trips (measured: 3 us of an 18 us launch went into ten
    // of them -- arguments, sizes, control block, pointers, four passes of block loads).
    const BaView& vw = views[blockIdx.y];
    const int dim = vw.dim, n = vw.dim_pad, band = vw.band_hbw;
    GPTR(double) S = vw.S; GPTR(double) M = vw.Minv; GPTR(double) Ldiag = vw.Ldiag; GPTR(double) Lsub = vw.Lsub; GPTR(double) scal = vw.scal;
    GPTR(BaCtl) ctl = vw.ctl;
    asm volatile("" :: "s"(dim), "s"(n), "s"(S), "s"(M), "s"(Ldiag), "s"(Lsub), "s"(scal), "s"(ctl), "s"(band));
    const int nb = n / NB;
    if (2 * m >= nb || dim == 0 || band >= 0 || (skip_small && cw_fits(dim))) return;   // a batch runs the panel pairs of its largest system; small systems may be k_chol_wg's
    const int bid = pin ? blockIdx.x >> 3 : blockIdx.x;
    {
        const int ncol0 = (2 * m + 1 < nb) ? 2 : 1, T0 = nb - 2 * m - ncol0;
        if (bid >= 1 + T0 + 2 * m + ncol0 + (m > 0 ? T0 * (T0 + 1) / 2 + 2 * m * T0 : 0)) return;
    }
    // the control block travels with the block loads below (no short circuit: one round trip, awaited after they are issued)
    const int c_stopped = ctl->stopped, c_done = ctl->outer_done, c_max = ctl->max_outer;
    extern __shared__ double cp_lds[];
    const int j = 2 * m, j1 = j + 1;
    const bool single = j1 >= nb;
    const bool prev = m > 0;
    const int kb0 = j - 2, kb1 = j - 1;
    const int ncol = single ? 1 : 2;
    const int n_rows = nb - j - ncol;                    // square row blocks below the pair
    const int n_extra = j + ncol;                        // Minv row blocks 0 .. j (j1)
    const int n_panel = 1 + n_rows + n_extra;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tr = (wave >> 1) * 16, tc = (wave & 1) * 16, lr = tid & 15, lk = lane >> 4;
    // A block of the matrix on its way into LDS: mode 0 copy from memory, 1 zero, 2 identity.  Every block is LOADED, whatever
    // its mode (blocks that are not read point at the pair's own diagonal block, which always exists): with the loads
    // unconditional all of them -- 44 per thread in a panel workgroup -- are in flight together, one round trip.
    struct Blk { const double* p; int mode; };
    auto blk = [&](const double* src, size_t r0, size_t c0, int mode) -> Blk {
        return mode == 0 ? Blk{src + r0 * n + c0, 0} : Blk{S + (size_t)j * NB * n + (size_t)j * NB, mode};
    };
    auto fill = [&](int mode, double ld, int r, int c) -> double { return mode == 0 ? ld : (mode == 2 && r == c ? 1.0 : 0.0); };
    if (bid >= n_panel) {
        // ---- trailing update of block (i2, j2), j2 >= j + ncol, with the previous pair
        const int T = nb - j - ncol;
        int u = bid - n_panel;
        const int n_sq_upd = T * (T + 1) / 2;
        double* dst; const double* srcI; size_t ri, rj; bool overwrite = false; int e = -1;
        if (u < n_sq_upd) {
            int bi = 0;
            while (u > bi) { u -= bi + 1; ++bi; }
            ri = (size_t)(j + ncol + bi) * NB; rj = (size_t)(j + ncol + u) * NB; dst = S; srcI = S;
        } else {
            u -= n_sq_upd;
            e = u / T;
            ri = (size_t)e * NB; rj = (size_t)(j + ncol + (u - e * T)) * NB; dst = M; srcI = M;
            overwrite = e >= kb0;                        // first contribution to this block: it holds no value yet
        }
        double* Li0 = cp_lds; double* Li1 = cp_lds + CP_BLK; double* Lj0 = cp_lds + 2 * CP_BLK; double* Lj1 = cp_lds + 3 * CP_BLK;
        // Minv block (e, k) is structurally zero for k < e
        const Blk bk[4] = {blk(srcI, ri, (size_t)kb0 * NB, e > kb0 ? 1 : 0), blk(srcI, ri, (size_t)kb1 * NB, e > kb1 ? 1 : 0),
                           blk(S, rj, (size_t)kb0 * NB, 0), blk(S, rj, (size_t)kb1 * NB, 0)};
        f64x2 ld[2][4];                                  // two adjacent columns per lane: 16-byte loads, 16 lanes to a 256-byte row
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = 2 * tid + it * 512, r = t / NB, c = t % NB;
#pragma unroll
            for (int q = 0; q < 4; ++q) ld[it][q] = *reinterpret_cast<const f64x2*>(bk[q].p + (size_t)r * n + c);
        }
        double old[4];
        if (!overwrite) {
#pragma unroll
            for (int q = 0; q < 4; ++q) old[q] = dst[(ri + tr + lk + 4 * q) * n + rj + tc + lr];
        }
        if (c_stopped | (c_done >= c_max)) return;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = 2 * tid + it * 512, r = t / NB, c = t % NB, o = r * (NB + 1) + c;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Li0[o + h] = fill(bk[0].mode, ld[it][0][h], r, c + h); Li1[o + h] = fill(bk[1].mode, ld[it][1][h], r, c + h);
                Lj0[o + h] = ld[it][2][h]; Lj1[o + h] = ld[it][3][h];
            }
        }
        __syncthreads();
        f64x4 acc = {0, 0, 0, 0};
        acc = mfma_tile32(Li0, Lj0, tr, tc, lr, lk, acc);
        acc = mfma_tile32(Li1, Lj1, tr, tc, lr, lk, acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[(ri + tr + lk + 4 * q) * n + rj + tc + lr] = overwrite ? -acc[q] : old[q] - acc[q];
        return;
    }
    // ---- panel workgroup
    const bool diag = bid == 0;
    const bool extra = bid > n_rows;
    const int i = diag ? j : (extra ? bid - 1 - n_rows : j + ncol + bid - 1);       // row block (extra: Minv row e)
    const size_t rj = (size_t)j * NB, rj1 = (size_t)j1 * NB, ri = (size_t)i * NB;
    const bool has_b = !diag;
    const double* Bsrc = extra ? M : S;
    double* Dj = cp_lds;               double* X = cp_lds + CP_BLK;        double* Dj1 = cp_lds + 2 * CP_BLK;
    double* B0 = cp_lds + 3 * CP_BLK;  double* B1 = cp_lds + 4 * CP_BLK;
    double* Ljk0 = cp_lds + 5 * CP_BLK; double* Ljk1 = cp_lds + 6 * CP_BLK;      // row j of the previous pair (later: L_j1,j and L_i,j)
    double* Lj1k0 = cp_lds + 7 * CP_BLK; double* Lj1k1 = cp_lds + 8 * CP_BLK;
    double* Lik0 = cp_lds + 9 * CP_BLK; double* Lik1 = cp_lds + 10 * CP_BLK;
    // Minv row e: block (e, e) starts as the identity, blocks left of it are zero, blocks right of it hold a value only once a
    // trailing update of an earlier launch has written them (e < kb0)
    const int mode0 = !has_b ? 1 : (!extra ? 0 : (i == j ? 2 : (i > j || i >= kb0 ? 1 : 0)));
    const int mode1 = (!has_b || single) ? 1 : (!extra ? 0 : (i == j1 ? 2 : (i >= kb0 ? 1 : 0)));
    const int ms = single ? 1 : 0;                                             // second-column blocks do not exist
    const int mp = prev ? 0 : 1, mps = (prev && !single) ? 0 : 1;
    const int mi0 = (!prev || !has_b || (extra && i > kb0)) ? 1 : 0, mi1 = (!prev || !has_b || (extra && i > kb1)) ? 1 : 0;
    const size_t c0 = (size_t)(prev ? kb0 : 0) * NB, c1 = (size_t)(prev ? kb1 : 0) * NB;
    // in the order of the LDS blocks: Dj X Dj1 B0 B1 Ljk0 Ljk1 Lj1k0 Lj1k1 Lik0 Lik1
    const Blk bk[11] = {blk(S, rj, rj, 0), blk(S, rj1, rj, ms), blk(S, rj1, rj1, ms), blk(Bsrc, ri, rj, mode0), blk(Bsrc, ri, rj1, mode1),
                        blk(S, rj, c0, mp), blk(S, rj, c1, mp), blk(S, rj1, c0, mps), blk(S, rj1, c1, mps), blk(Bsrc, ri, c0, mi0), blk(Bsrc, ri, c1, mi1)};
    {
        f64x2 ld[2][11];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = 2 * tid + it * 512, r = t / NB, c = t % NB;
#pragma unroll
            for (int q = 0; q < 11; ++q) ld[it][q] = *reinterpret_cast<const f64x2*>(bk[q].p + (size_t)r * n + c);
        }
        if (c_stopped | (c_done >= c_max)) return;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = 2 * tid + it * 512, r = t / NB, c = t % NB, o = r * (NB + 1) + c;
#pragma unroll
            for (int q = 0; q < 11; ++q) {
                cp_lds[q * CP_BLK + o] = fill(bk[q].mode, ld[it][q][0], r, c);
                cp_lds[q * CP_BLK + o + 1] = fill(bk[q].mode, ld[it][q][1], r, c + 1);
            }
        }
    }
    __syncthreads();
    if (prev) {                                          // 1. lookahead of the blocks the first factorisations need
        f64x4 acc = {0, 0, 0, 0};
        acc = mfma_tile32(Ljk0, Ljk0, tr, tc, lr, lk, acc); acc = mfma_tile32(Ljk1, Ljk1, tr, tc, lr, lk, acc);
        tile_sub(Dj, tr, tc, lr, lk, acc);
        if (!single) {
            f64x4 ax = {0, 0, 0, 0};
            ax = mfma_tile32(Lj1k0, Ljk0, tr, tc, lr, lk, ax); ax = mfma_tile32(Lj1k1, Ljk1, tr, tc, lr, lk, ax);
            tile_sub(X, tr, tc, lr, lk, ax);
        }
        if (has_b) {
            f64x4 ab = {0, 0, 0, 0};
            ab = mfma_tile32(Lik0, Ljk0, tr, tc, lr, lk, ab); ab = mfma_tile32(Lik1, Ljk1, tr, tc, lr, lk, ab);
            tile_sub(B0, tr, tc, lr, lk, ab);
        }
    }
    __syncthreads();
    bool fail = false;
    // The factored blocks leave the registers through LDS (every lane owns a ROW there): the workgroup then stores them to memory
    // 32 lanes to a row, where a lane-per-row store touched one cache line per lane (2.4k cycles of the factoring wavefront).
    // L_jj goes into X's place (wavefront 0 alone reads X, at the start of 2a), L_j1,j1 into D_j1's, L_i,j1 into B1's.
    auto panel_to_lds = [&](const PanelRegs& p, double* Ld, double* Lb) {
        if (lane < 16) {
            if (Ld) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    Ld[lane * (NB + 1) + c] = c <= lane ? p.dA[c] : 0.0;        Ld[lane * (NB + 1) + 16 + c] = 0.0;
                    Ld[(16 + lane) * (NB + 1) + c] = p.x[c];                     Ld[(16 + lane) * (NB + 1) + 16 + c] = c <= lane ? p.dB[c] : 0.0;
                }
            }
        } else if (lane < 48 && Lb) {
#pragma unroll
            for (int c = 0; c < 16; ++c) { Lb[(lane - 16) * (NB + 1) + c] = p.x[c]; Lb[(lane - 16) * (NB + 1) + 16 + c] = p.y[c]; }
        }
    };
    auto block_to_memory = [&](const double* L, double* dst, size_t pitch) {       // all four wavefronts
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int t = tid + it * 256, r = t / NB, c = t % NB;
            dst[(size_t)r * pitch + c] = L[r * (NB + 1) + c];
        }
    };
    if (wave == 0 && (!single || diag)) {                // 2a. [D_j; X] -> L_jj, L_j1,j   (single: the diagonal workgroup only needs L_jj)
        PanelRegs p;
        fail = chol_panel_dpp(p, Dj, single ? nullptr : X, lane, cp_lds + 11 * CP_BLK);
        panel_to_lds(p, diag ? X : nullptr, single ? nullptr : Ljk0);                  // L_j1,j for step 3
        if (diag && fail && lane == 0) scal[5] = 1.0;
    } else if (wave == 1 && has_b) {                     // 2b. [D_j; B0] -> L_i,j
        PanelRegs p;
        chol_panel_dpp(p, Dj, B0, lane, cp_lds + 11 * CP_BLK + CH_SCR);
        panel_to_lds(p, nullptr, Ljk1);
    } else if (prev && !single && wave >= 2) {           // 2c. the rest of the lookahead, beside the factorisations
        if (wave == 2) {
            for (int t4 = 0; t4 < 4; ++t4) {
                const int r2 = (t4 >> 1) * 16, c2 = (t4 & 1) * 16;
                f64x4 acc = {0, 0, 0, 0};
                acc = mfma_tile32(Lj1k0, Lj1k0, r2, c2, lr, lk, acc); acc = mfma_tile32(Lj1k1, Lj1k1, r2, c2, lr, lk, acc);
                tile_sub(Dj1, r2, c2, lr, lk, acc);
            }
        } else if (has_b) {
            for (int t4 = 0; t4 < 4; ++t4) {
                const int r2 = (t4 >> 1) * 16, c2 = (t4 & 1) * 16;
                f64x4 acc = {0, 0, 0, 0};
                acc = mfma_tile32(Lik0, Lj1k0, r2, c2, lr, lk, acc); acc = mfma_tile32(Lik1, Lj1k1, r2, c2, lr, lk, acc);
                tile_sub(B1, r2, c2, lr, lk, acc);
            }
        }
    }
    __syncthreads();
    if (diag) {
        block_to_memory(X, Ldiag + (size_t)j * NB * NB, NB);
        // L_j1,j for k_chol_xsolve when the rhs row lives in block j1.  Not into S: the other panel workgroups read A_j1,j.
        if (!single) block_to_memory(Ljk0, Lsub + (size_t)j1 * NB * NB, NB);
    } else {
        block_to_memory(Ljk1, (extra ? M : S) + ri * n + rj, n);
    }
    if (single) return;
    {                                                    // 3. the second column sees the first: K = 32
        f64x4 acc = {0, 0, 0, 0};
        acc = mfma_tile32(Ljk0, Ljk0, tr, tc, lr, lk, acc);
        tile_sub(Dj1, tr, tc, lr, lk, acc);
        if (has_b) {
            f64x4 ab = {0, 0, 0, 0};
            ab = mfma_tile32(Ljk1, Ljk0, tr, tc, lr, lk, ab);
            tile_sub(B1, tr, tc, lr, lk, ab);
        }
    }
    __syncthreads();
    if (wave == 0) {                                     // 4. [D_j1; B1] -> L_j1,j1, L_i,j1
        PanelRegs p;
        fail = chol_panel_dpp(p, Dj1, has_b ? B1 : nullptr, lane, cp_lds + 11 * CP_BLK);
        panel_to_lds(p, diag ? Dj1 : nullptr, has_b ? B1 : nullptr);
        if (diag && fail && lane == 0) scal[5] = 1.0;
    }
    __syncthreads();
    if (diag) block_to_memory(Dj1, Ldiag + (size_t)j1 * NB * NB, NB);
    else block_to_memory(B1, (extra ? M : S) + ri * n + rj1, n);
}

// the whole factorisation + L^-T rows: ceil(nb / 2) launches
// nb = panels of the (largest) system, count = problems (grid.y; every problem reads its own size from its view)
void enqueue_cholesky(hipStream_t s, const BaView* d_views, int count, int nb, int skip_small = 0, bool spread = false)
{
    // the attribute belongs to the (function, device) pair: once per device this process uses
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev].load()) {
        (void)hipFuncSetAttribute((const void*)k_chol_pair, hipFuncAttributeMaxDynamicSharedMemorySize, CP_LDS_BYTES);
        attr_set[dev].store(true);
    }
    for (int m = 0; 2 * m < nb; ++m) {
        const int j = 2 * m, ncol = (j + 1 < nb) ? 2 : 1;
        const int T = nb - j - ncol;
        const int n_panel = 1 + (nb - j - ncol) + (j + ncol);
        const int n_update = m > 0 ? T * (T + 1) / 2 + j * T : 0;
        // 32 CUs of one XCD hold a latency-bound launch of one problem; a throughput-bound one (or a batch) needs all 256
        // (spread: the front end keeps a few CUs of EVERY XCD free for this chain -- lpslam_hip_set_mapping_reserve -- so its workgroups go there)
        const int pin = (!spread && count == 1 && (n_panel + n_update) <= 96) ? 1 : 0;
        hipLaunchKernelGGL(k_chol_pair, dim3((n_panel + n_update) * (pin ? 8 : 1), count), dim3(256), CP_LDS_BYTES, s, d_views, m, pin, skip_small);
    }
}
// x_p = L^-T y with y = L[dim][0..dim): one wavefront per row of the (upper triangular) L^-T, butterfly sum
__global__ __launch_bounds__(256) void k_chol_xsolve(const BaView* __restrict__ views, int pin, int skip_small)
{
    if (pin && (blockIdx.x & 7)) return;  // XCD 0 only, like k_chol_pair: its inputs sit in that L2
    BA_VIEW(v);
    BA_VIEW_HEAD("s"(v.dim), "s"(v.dim_pad), "s"(v.ctl), "s"(v.S), "s"(v.Ldiag), "s"(v.Lsub), "s"(v.Minv), "s"(v.xp), "s"(v.band_hbw));
    if (v.band_hbw >= 0 || ba_flags(v.ctl).idle() || (skip_small && cw_fits(v.dim))) return;
    const int lane = threadIdx.x & 63;
    const int i = (pin ? blockIdx.x >> 3 : blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (i >= v.dim) return;
    const int n = v.dim_pad;
    // y = L^-1 rhs is row `dim` of L: off-diagonal blocks live in S, the part inside the row's own diagonal block in Ldiag
    const int yb = v.dim / NB;
    const double* y = v.S + (size_t)v.dim * n;
    const double* yd = v.Ldiag + ((size_t)yb * NB + (v.dim - yb * NB)) * NB - (size_t)yb * NB;
    // an odd block is the second column of a panel pair: its block left of the diagonal went to Lsub (see k_chol_pair)
    const int ys0 = (yb & 1) ? (yb - 1) * NB : yb * NB;
    const double* ys = v.Lsub + ((size_t)yb * NB + (v.dim - yb * NB)) * NB - (size_t)ys0;
    const double* m = v.Minv + (size_t)i * n;
    double acc = 0;
    for (int c = (i / NB) * NB + lane; c < v.dim; c += 64) acc += m[c] * (c < ys0 ? y[c] : (c < yb * NB ? ys[c] : yd[c]));     // blocks left of the diagonal are empty
    acc = wave_sum(acc);
    if (lane == 0) v.xp[i] = acc;
}

void enqueue_xsolve(hipStream_t s, const BaView* d_views, int count, int dim, int skip_small = 0, bool spread = false)
{
    const int pin = (!spread && count == 1) ? 1 : 0;
    hipLaunchKernelGGL(k_chol_xsolve, dim3((dim + 3) / 4 * (pin ? 8 : 1), count), dim3(256), 0, s, d_views, pin, skip_small);
}

#include "ba_solve.inl"
#include "ba_band.inl"

// factorisation + solve of `count` reduced systems.  `wg`: the systems that fit one compute unit go to k_chol_wg (one workgroup
// each, one launch), the others through the panel-pair chain and k_chol_xsolve (each kernel skips the problems of the other
// kind).  Measured (MI355X, 295 x 295, 10 LM iterations per problem, round 3): batches of 4 / 16 / 32 / 48 / 64 / 96 windows take
// 1.78 / 4.10 / 7.96 / 11.8 / 15.2 / 22.4 ms through the chain and 2.65 / 4.97 / 7.99 / 10.9 / 13.8 / 19.6 ms through k_chol_wg: the
// chain spreads one problem's trailing updates and L^-T rows over 20-60 workgroups, which only stops paying once the batch alone
// fills the chip.  (Factoring k_chol_wg's diagonal blocks with the DPP strips of the chain instead of the readlane form -- 2 x 2.8 k
// cycles against 2 x 10 k -- changed none of these numbers: that wavefront works beside the seven that update tiles.)
constexpr int CW_MIN_BATCH = 40;
// LPSLAM_HIP_CW_MIN_BATCH overrides the threshold (measurements)
int cw_min_batch()
{
    static const int v = [] { const char* e = getenv("LPSLAM_HIP_CW_MIN_BATCH"); return e ? atoi(e) : CW_MIN_BATCH; }();
    return v;
}
void enqueue_factor_solve(hipStream_t s, const BaView* d_views, int count, int nb_max, int dim_max, bool wg, bool any_small, bool any_large, bool spread = false)
{
    if (!wg) { any_large = any_large || any_small; any_small = false; }
    if (any_small) {
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && !attr_set[dev].load()) {
            (void)hipFuncSetAttribute((const void*)k_chol_wg, hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS_BYTES);
            attr_set[dev].store(true);
        }
        hipLaunchKernelGGL(k_chol_wg, dim3(1, count), dim3(CW_THREADS), CW_LDS_BYTES, s, d_views);
    }
    if (any_large) {
        enqueue_cholesky(s, d_views, count, nb_max, any_small ? 1 : 0, spread);
        enqueue_xsolve(s, d_views, count, dim_max, any_small ? 1 : 0, spread);
    }
}

// ---- landmark back substitution and update (4 lanes per landmark); the last block applies x_p to the poses -------------------
__global__ __launch_bounds__(256) void k_ba_backsub(const BaView* __restrict__ views, int skip_one_pass)
{
    BA_VIEW(v);
    BA_VIEW_HEAD("s"(v.part_n), "s"(v.ctl));
    const int point_blocks = v.part_n;
    if ((int)blockIdx.x > point_blocks) return;
    if (skip_one_pass && upd_takes(v.n_points, v.n_free, v.n_poses)) return;      // that problem's update ran in k_ba_update
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle()) return;
    const double lambda = fl.lambda;
    ba_select_idx(v, fl.cur); ba_lin_set(v, fl.cur);
    GPTR(double) poses_out = sel2(v.poses_buf[0], v.poses_buf[1], fl.cur ^ 1);
    GPTR(double) points_out = sel2(v.points_buf[0], v.points_buf[1], fl.cur ^ 1);
    if ((int)blockIdx.x == point_blocks) {
        // trial poses = exp(x_p) * poses; scal[3] = sum x_p (lambda x_p + b_p) (fixed order, one wavefront)
        if (threadIdx.x == 0) v.ctl->cur_launch = fl.cur;          // what the trial launch reads while the decision flips `cur`
        if (threadIdx.x >= 64) return;
        double sc = 0;
        for (int p = threadIdx.x; p < v.n_poses; p += 64) {
            const int slot = v.pose_slot[p];
            if (slot < 0) { for (int i = 0; i < 7; ++i) poses_out[7 * p + i] = v.poses[7 * p + i]; continue; }
            pose_oplus(v.poses + 7 * p, v.xp + 6 * slot, poses_out + 7 * p);
            for (int a = 0; a < 6; ++a) { const double x = v.xp[6 * slot + a]; sc += x * (lambda * x + v.bp[6 * slot + a]); }
        }
        sc = wave_sum(sc);
        if (threadIdx.x == 0) v.scal[3] = sc;
        return;
    }
    const int g = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
    double sc = 0;
    double r[3] = {0, 0, 0};
    if (g < v.n_points) {
        // two observations of the lane at a time, their index / slot / W / x_p load chains side by side (same order of the sums)
        const int s_end = v.pt_start[g + 1];
        for (int s = v.pt_start[g] + sub; s < s_end; s += 8) {
            int kk[2], slot[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) kk[u] = v.pt_obs[min(s + 4 * u, s_end - 1)];
#pragma unroll
            for (int u = 0; u < 2; ++u) slot[u] = v.pose_slot[v.o_pose[kk[u]]];
            double wv[2][18], xv[2][6];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const double* Wk = v.W + 18 * (size_t)kk[u];
                const double* x = v.xp + 6 * (size_t)max(slot[u], 0);
#pragma unroll
                for (int i = 0; i < 18; ++i) wv[u][i] = Wk[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) xv[u][i] = x[i];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (s + 4 * u < s_end && slot[u] >= 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int rr = 0; rr < 6; ++rr) r[c] += wv[u][rr * 3 + c] * xv[u][rr];
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double a1 = __shfl_xor(r[c], 1);
        const double lo = (sub & 1) ? a1 + r[c] : r[c] + a1;      // (s0 + s1) and (s2 + s3), same order in both lanes
        const double a2 = __shfl_xor(lo, 2);
        r[c] = (sub & 2) ? a2 + lo : lo + a2;                     // (s0 + s1) + (s2 + s3)
    }
    if (g < v.n_points && sub == 0) {
        const double b0 = v.bl[3 * (size_t)g], b1 = v.bl[3 * (size_t)g + 1], b2 = v.bl[3 * (size_t)g + 2];
        const double q0 = b0 - r[0], q1 = b1 - r[1], q2 = b2 - r[2];
        double h[6];
        point_hinv(v.Hll + 6 * (size_t)g, lambda, h);
        const double x0 = h[0] * q0 + h[1] * q1 + h[2] * q2;
        const double x1 = h[1] * q0 + h[3] * q1 + h[4] * q2;
        const double x2 = h[2] * q0 + h[4] * q1 + h[5] * q2;
        points_out[3 * (size_t)g] = v.points[3 * (size_t)g] + x0;
        points_out[3 * (size_t)g + 1] = v.points[3 * (size_t)g + 1] + x1;
        points_out[3 * (size_t)g + 2] = v.points[3 * (size_t)g + 2] + x2;
        sc = x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
    }
    __shared__ double sm[4];
    sc = wave_sum(sc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = sc;
    __syncthreads();
    if (threadIdx.x == 0) v.part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// per-observation chi2 (non robust) and depth sign of the accepted state
__global__ __launch_bounds__(256) void k_ba_obs_chi2(const BaView* __restrict__ views, double* chi2, uint8_t* depth_pos)
{
    BA_VIEW(v);
    ba_select(v, 0);
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= v.n_obs) return;
    const int p = v.o_pose[k], j = v.o_point[k];
    double R[9], e[3], pc[3];
    quat_to_rot(v.poses + 7 * p, R);
    const double X[3] = {v.points[3 * j], v.points[3 * j + 1], v.points[3 * j + 2]};
    const int D = ba_residual(v, k, R, v.poses + 7 * p + 4, X, e, pc);
    const int ko = v.o_orig[k];                 // the caller's observation index
    chi2[ko] = v.o_w[k] * (e[0] * e[0] + e[1] * e[1] + (D == 3 ? e[2] * e[2] : 0.0));
    depth_pos[ko] = pc[2] > 0 ? 1 : 0;
}

// ---- motion-only pose optimisation, the whole flow in one workgroup ---------------------------------------------------------
// [UPSTREAM] optimize::pose_optimizer: one SE3 vertex, unary reprojection edges to fixed landmarks, 4 rounds of 10 Levenberg
// iterations (g2o lambda control), after every round the observations with chi2 > 5.991 (mono) / 7.815 (stereo) become
// outliers (and may come back), Huber is dropped after the third round, the flow stops when fewer than 5 inliers remain.
// The tracker needs this once per frame: instead of ~600 launches through the general BA machinery the 6x6 system lives in
// LDS and one launch returns the pose (one workgroup; several frames / candidates could share a launch, one workgroup each).
struct PoShared {
    double pose[7];
    double red[8];                     // the wavefronts' partial sums of po_block_sum
    double sums[28];                   // a pass's 28 sums (upper triangle of H, b, chi2)
};

// The kernel is one workgroup and a chain of 50 - 65 dependent Levenberg trials; a trial is ~3 us of latency, not of arithmetic
// (round 4, in-kernel cycle stamps: pass over the observations ~2000 cycles, the 28-value reduction ~1700, decision + 6x6 solve +
// pose update ~3500).  Four wavefronts, one per SIMD: the reduction's register step costs every SIMD half of what it costs with
// eight, the passes are issue-bound either way, and the serial section between two passes is executed by EVERY thread on
// replicated registers (a SIMD runs one lane as fast as 64), so nothing is published and no barrier follows it.  In the
// per-observation arithmetic, reciprocals and reciprocal square roots come from v_rcp_f64 / v_rsq_f64 plus one cubic correction
// step (1.4e-16 relative error, measured) where IEEE division and sqrt cost ~30 instructions each.  Products and sums are contracted
// to fused multiply-adds in these functions (the rest of the file is compiled without contraction): the Cholesky solve alone went
// from 107 multiplications + 73 additions to half as many instructions on the serial section's critical path.
#ifndef LPSLAM_PO_T
#define LPSLAM_PO_T 256
#endif
constexpr int PO_T = LPSLAM_PO_T;
static_assert(PO_T >= 128 && PO_T % 64 == 0, "k_pose_optimize: the 27-value reduction needs at least two wavefronts (one wavefront faulted on the device, round 4)");
constexpr int PO_W = PO_T / 64;
__device__ __forceinline__ double po_rcp(double d) { return fast_rcp(d); }
__device__ __forceinline__ double po_rsqrt(double d) { return fast_rsqrt(d); }
__device__ __forceinline__ double po_block_sum(double v, PoShared& sh)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = sh.red[0];
#pragma unroll
    for (int w = 1; w < PO_W; ++w) s += sh.red[w];
    return s;
}
__device__ __forceinline__ void po_quat_to_rot(const double* q, double* R)
{
#pragma clang fp contract(fast)
    const double rn = po_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] * rn, x = q[1] * rn, y = q[2] * rn, z = q[3] * rn;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
// the same for a quaternion of unit length (what po_oplus returns): no normalisation on the chain between two passes
__device__ __forceinline__ void po_unit_quat_to_rot(const double* q, double* R)
{
#pragma clang fp contract(fast)
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ void po_huber(double e2, double delta, double* rho0, double* rho1)
{
#pragma clang fp contract(fast)
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; }
    else { const double rs = po_rsqrt(e2); *rho0 = 2 * (e2 * rs) * delta - dsqr; *rho1 = delta * rs; }
}
// sin and cos of a half angle up to 0.5 rad (every Levenberg step of a tracked frame) from their Taylor polynomials -- two
// interleaved Horner chains of eight terms, truncation below 1e-18 -- instead of the library's ~130 instructions of argument
// reduction; larger angles take the library call
__device__ __forceinline__ void po_sincos_half(double h, double* sn, double* cs)
{
    if (h <= 0.5) {
        const double z = h * h;
        double ps = 1.0 / 355687428096000.0, pc = 1.0 / 20922789888000.0;     // 1 / 17!, 1 / 16!
        ps = fma(ps, -z, 1.0 / 1307674368000.0);  pc = fma(pc, -z, 1.0 / 87178291200.0);       // 1 / 15!, 1 / 14!
        ps = fma(ps, -z, 1.0 / 6227020800.0);     pc = fma(pc, -z, 1.0 / 479001600.0);         // 1 / 13!, 1 / 12!
        ps = fma(ps, -z, 1.0 / 39916800.0);       pc = fma(pc, -z, 1.0 / 3628800.0);           // 1 / 11!, 1 / 10!
        ps = fma(ps, -z, 1.0 / 362880.0);         pc = fma(pc, -z, 1.0 / 40320.0);             // 1 / 9!, 1 / 8!
        ps = fma(ps, -z, 1.0 / 5040.0);           pc = fma(pc, -z, 1.0 / 720.0);               // 1 / 7!, 1 / 6!
        ps = fma(ps, -z, 1.0 / 120.0);            pc = fma(pc, -z, 1.0 / 24.0);                // 1 / 5!, 1 / 4!
        ps = fma(ps, -z, 1.0 / 6.0);              pc = fma(pc, -z, 0.5);                       // 1 / 3!, 1 / 2!
        *sn = fma(h * z, -ps, h);                                                              // h - h^3 (1/3! - ...)
        *cs = fma(z, -pc, 1.0);                                                                // 1 - h^2 (1/2! - ...)
    } else {
        sincos(h, sn, cs);
    }
}
// pose_oplus: exp(d) * pose with one sincos of the half angle and the fast reciprocals.  The increment's rotation and its V matrix
// are applied as Rodrigues sums (v + a w x v + b w x (w x v)), not as 3x3 matrices: 40 instructions where forming W^2, R and V took
// 80 (this runs between two passes, on every thread's own registers).
__device__ __forceinline__ void po_oplus(const double* pose, const double* d, double* out)
{
#pragma clang fp contract(fast)
    const double wx = d[0], wy = d[1], wz = d[2];
    const double theta2 = wx * wx + wy * wy + wz * wz;
    double a, b, c, qe[4];
    if (theta2 < 1e-10) {
        a = 1.0; b = 0.5; c = 1.0 / 6.0;
        qe[0] = 1.0; qe[1] = 0.5 * wx; qe[2] = 0.5 * wy; qe[3] = 0.5 * wz;
    } else {
        const double rt = po_rsqrt(theta2), theta = theta2 * rt, rt2 = rt * rt;
        double sh2, ch2;
        po_sincos_half(0.5 * theta, &sh2, &ch2);
        const double st = 2.0 * sh2 * ch2, omc = 2.0 * sh2 * sh2;        // sin(theta), 1 - cos(theta)
        a = st * rt;
        b = omc * rt2;
        c = (theta - st) * (rt2 * rt);
        const double shq = sh2 * rt;
        qe[0] = ch2; qe[1] = shq * wx; qe[2] = shq * wy; qe[3] = shq * wz;
    }
    const double* t = pose + 4;
    // w x t, w x (w x t), w x u, w x (w x u) with u the translation part of the increment
    const double c1[3] = {wy * t[2] - wz * t[1], wz * t[0] - wx * t[2], wx * t[1] - wy * t[0]};
    const double c2[3] = {wy * c1[2] - wz * c1[1], wz * c1[0] - wx * c1[2], wx * c1[1] - wy * c1[0]};
    const double u1[3] = {wy * d[5] - wz * d[4], wz * d[3] - wx * d[5], wx * d[4] - wy * d[3]};
    const double u2[3] = {wy * u1[2] - wz * u1[1], wz * u1[0] - wx * u1[2], wx * u1[1] - wy * u1[0]};
    double tn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) tn[i] = fma(c, u2[i], fma(b, u1[i], d[3 + i])) + fma(b, c2[i], fma(a, c1[i], t[i]));
    const double* q = pose;
    double qn[4];
    qn[0] = qe[0] * q[0] - qe[1] * q[1] - qe[2] * q[2] - qe[3] * q[3];
    qn[1] = qe[0] * q[1] + qe[1] * q[0] + qe[2] * q[3] - qe[3] * q[2];
    qn[2] = qe[0] * q[2] - qe[1] * q[3] + qe[2] * q[0] + qe[3] * q[1];
    qn[3] = qe[0] * q[3] + qe[1] * q[2] - qe[2] * q[1] + qe[3] * q[0];
    const double rn = po_rsqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = qn[i] * rn;
#pragma unroll
    for (int i = 0; i < 3; ++i) out[4 + i] = tn[i];
}
// pose half of ba_jacobians (the landmark is a constant here), with one reciprocal
__device__ __forceinline__ void po_jacobian(const BaCam& c, const double* pc, double iz, int D, double B[3][6])
{
#pragma clang fp contract(fast)
    const double x = pc[0], y = pc[1], iz2 = iz * iz;
    B[0][0] = x * y * iz2 * c.fx;          B[0][1] = -(1.0 + (x * x * iz2)) * c.fx; B[0][2] = y * iz * c.fx;
    B[0][3] = -iz * c.fx;                  B[0][4] = 0.0;                            B[0][5] = x * iz2 * c.fx;
    B[1][0] = (1.0 + y * y * iz2) * c.fy;  B[1][1] = -x * y * iz2 * c.fy;            B[1][2] = -x * iz * c.fy;
    B[1][3] = 0.0;                         B[1][4] = -iz * c.fy;                     B[1][5] = y * iz2 * c.fy;
    B[2][0] = B[0][0] - c.fxb * y * iz2;   B[2][1] = B[0][1] + c.fxb * x * iz2;      B[2][2] = B[0][2];
    B[2][3] = B[0][3];                     B[2][4] = 0.0;                            B[2][5] = B[0][5] - c.fxb * iz2;
    if (D == 2) {
#pragma unroll
        for (int k = 0; k < 6; ++k) B[2][k] = 0.0;
    }
}
// residual of observation k at pose p7; returns the dimension (2 / 3)
__device__ __forceinline__ int po_residual(const BaCam& cam, const double* R, const double* t, const double* X, const lpslam_hip_ba_obs& o, double* e, double* pc, double* iz_out = nullptr)
{
#pragma clang fp contract(fast)
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2] + t[i];
    const double iz = po_rcp(pc[2]);
    if (iz_out) *iz_out = iz;
    const double u = cam.fx * pc[0] * iz + cam.cx, vv = cam.fy * pc[1] * iz + cam.cy;
    e[0] = o.u - u; e[1] = o.v - vv;
    if (o.ur < 0) { e[2] = 0; return 2; }
    e[2] = o.ur - (u - cam.fxb * iz);
    return 3;
}
// One observation as the kernel keeps it in LDS: measurement, weight and the landmark it sees (56 bytes); the first `cache_n`
// observations live there, the rest (only very large n) is read from global memory like before.
struct PoObs { double u, v, ur, w, X[3]; };

template <bool ALL_CACHED>
struct PoData {
    const double* pts; const lpslam_hip_ba_obs* obs; const PoObs* cache; const uint8_t* act; int n, cache_n;
    __device__ __forceinline__ void get(int k, lpslam_hip_ba_obs& o, double* X) const
    {
        // (ALL_CACHED is a compile-time fact on purpose: with both sources in one function the compiler merges them into generic
        // pointers and every LDS read becomes a flat load)
        if (ALL_CACHED || k < cache_n) {
            const PoObs c = cache[k];
            o.pose = 0; o.point = 0; o.u = c.u; o.v = c.v; o.ur = c.ur; o.inv_sigma2 = c.w;
            X[0] = c.X[0]; X[1] = c.X[1]; X[2] = c.X[2];
        } else {
            o = obs[k];
            const double* p = pts + 3 * (size_t)o.point;
            X[0] = p[0]; X[1] = p[1]; X[2] = p[2];
        }
    }
};

// The 28 sums of a pass (upper triangle of H, b, chi2) over the PO_T threads: first over each quad of lanes in registers (two DPP
// exchanges per value), then one lane of four stores its 28 partials transposed into LDS, PO_T / 32 lanes per value add eight of
// them each (interleaved: neighbouring lanes read neighbouring words) and finish inside their DPP row.
template <int CTRL>
__device__ __forceinline__ double quad_swap(double v)     // DPP exchange inside a row (a shuffle would go through the LDS crossbar)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int PO_NV = 28;                  // values per pass
constexpr int PO_Q = PO_T / 4;             // partials per value after the quad step
constexpr int PO_TR = PO_Q + 8;            // padded row of the transposed reduction buffer [PO_NV][PO_Q]: the second step's lanes read (value q, partial j + 8 i) --
                                           // q 72 + j puts the eight values of a wavefront's read on different banks (with + 1 they met four to a bank)
constexpr int PO_LPV = PO_T / 32;          // lanes per value in the second step (16 = one DPP row at 512 threads)
static_assert(PO_NV * PO_LPV <= PO_T && PO_Q == 8 * PO_LPV && (PO_LPV == 16 || PO_LPV == 8 || PO_LPV == 4), "po_reduce28 layout");
__device__ __forceinline__ void po_reduce28(double (&acc)[PO_NV], double* tr, double* out, int lane_stride)
{
    const int tid = threadIdx.x;
    // (lane_stride: observations sit in every lane / every second / every fourth -- po_pass: the register step shrinks with them)
    if (lane_stride == 1) {
#pragma unroll
        for (int q = 0; q < PO_NV; ++q) acc[q] += quad_swap<0xB1>(acc[q]);           // lanes 0<->1, 2<->3
    }
    if (lane_stride <= 2) {
#pragma unroll
        for (int q = 0; q < PO_NV; ++q) acc[q] += quad_swap<0x4E>(acc[q]);           // lanes 0<->2, 1<->3
    }
    if ((tid & 3) == 0) {
#pragma unroll
        for (int q = 0; q < PO_NV; ++q) tr[q * PO_TR + (tid >> 2)] = acc[q];
    }
    __syncthreads();
    const int q = tid / PO_LPV, j = tid % PO_LPV;
    double s = 0;
    if (q < PO_NV) {
        const double* row = tr + q * PO_TR + j;
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = row[i * PO_LPV];
        s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    s += quad_swap<0xB1>(s);
    s += quad_swap<0x4E>(s);
    if (PO_LPV >= 8) s += quad_swap<0x141>(s);           // row_half_mirror: the other quad of the eight
    if (PO_LPV >= 16) s += quad_swap<0x140>(s);          // row_mirror: the other half of the row
    if (q < PO_NV && j == 0) out[q] = s;
    __syncthreads();
}

#ifdef LPSLAM_PO_STAMPS
__device__ double g_po_stamps[16];
#define PO_STAMP(k) do { if (threadIdx.x == 0) { const double now_ = (double)clock64(); po_acc[k] += now_ - po_last; po_last = now_; } } while (0)
#define PO_ST_PARAM , double (&po_acc)[16], double& po_last
#define PO_ST_ARG , po_acc, po_last
#else
#define PO_STAMP(k) do {} while (0)
#define PO_ST_PARAM
#define PO_ST_ARG
#endif

// One observation's share of the 28 sums (upper triangle of H, b, robustified chi2) at the pose (R, t)
__device__ __forceinline__ void po_accumulate(const BaCam& cam, const double (&R)[9], const double (&t)[3], const lpslam_hip_ba_obs& o, const double (&X)[3], int robust, double (&acc)[PO_NV])
{
#pragma clang fp contract(fast)
    double e[3], pc[3], B[3][6], iz;
    const int D = po_residual(cam, R, t, X, o, e, pc, &iz);
    const double om = o.inv_sigma2;
    const double chi = om * (e[0] * e[0] + e[1] * e[1] + (D == 3 ? e[2] * e[2] : 0.0));
    const double delta = D == 3 ? cam.hub_stereo : cam.hub_mono;
    double w = om, c = chi;
    if (robust && delta > 0) { double r0, r1; po_huber(chi, delta, &r0, &r1); w *= r1; c = r0; }
    acc[27] += c;
    po_jacobian(cam, pc, iz, D, B);
    // w B once (18 products), then every entry is three fused multiply-adds onto its running sum; the columns that
    // are structurally zero (B[0][4], B[1][3], B[2][4]) are skipped by hand -- the compiler may not drop x * 0
    double wB[3][6], we[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        we[r] = -w * e[r];
#pragma unroll
        for (int a = 0; a < 6; ++a) wB[r][a] = w * B[r][a];
    }
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c2 = a; c2 < 6; ++c2) {
            double s2 = acc[idx];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const bool zero = (r == 0 && (a == 4 || c2 == 4)) || (r == 1 && (a == 3 || c2 == 3)) || (r == 2 && (a == 4 || c2 == 4));
                if (!zero) s2 = fma(wB[r][a], B[r][c2], s2);
            }
            acc[idx++] = s2;
        }
        double s3 = acc[21 + a];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const bool zero = (r == 0 && a == 4) || (r == 1 && a == 3) || (r == 2 && a == 4);
            if (!zero) s3 = fma(B[r][a], we[r], s3);
        }
        acc[21 + a] = s3;
    }
}

// One pass over the active observations at pose p7: the 27 sums of the linearised system (upper triangle of H, then b) and the
// (robustified) chi2 as the 28th, into sh.sums (valid until the next pass).  A trial's chi2 and the NEXT iteration's linearisation are the same
// pass: the trial is accepted nearly always, and then its pose is the pose to linearise at (one reduction less per Levenberg
// iteration; a rejected trial wastes the 27 sums).
template <bool ALL_CACHED>
__device__ __forceinline__ void po_pass(const BaCam& cam, const double (&p7)[7], const PoData<ALL_CACHED>& d, int robust, double* tr, PoShared& sh PO_ST_PARAM)
{
#pragma clang fp contract(fast)
    const int tid = threadIdx.x;
    double R[9];
    po_quat_to_rot(p7, R);
    const double t[3] = {p7[4], p7[5], p7[6]};
    double acc[PO_NV];
#pragma unroll
    for (int q = 0; q < PO_NV; ++q) acc[q] = 0;
    // few observations are spread out, one per quad of lanes (up to PO_T / 4) or one per pair: the reduction's first step adds the
    // four lanes of a quad in registers, 168 instructions when every lane carries sums -- none when only one lane of four does
    const int lane_stride = d.n <= PO_T / 4 ? 4 : d.n <= PO_T / 2 ? 2 : 1;
    for (int k = (tid % lane_stride) ? d.n : tid / lane_stride; k < d.n; k += PO_T / lane_stride) {
        if (!d.act[k]) continue;
        double X[3];
        lpslam_hip_ba_obs o;
        d.get(k, o, X);
        po_accumulate(cam, R, t, o, X, robust, acc);
    }
    PO_STAMP(1);
#ifdef LPSLAM_PO_DUP_REDUCE
    {   // development: the reduction twice (the first one's result is discarded) -- the difference in time per pass is its cost
        double acc2[PO_NV];
#pragma unroll
        for (int q = 0; q < PO_NV; ++q) { acc2[q] = acc[q]; asm volatile("" : "+v"(acc2[q])); }
        po_reduce28(acc2, tr, sh.sums, lane_stride);
    }
#endif
    po_reduce28(acc, tr, sh.sums, lane_stride);
    PO_STAMP(2);
#ifdef LPSLAM_PO_STAMPS
    if (tid == 0) po_acc[15] += 1;
#endif
}

// (H + lambda I) x = b by a 6x6 Cholesky factorisation and two substitutions, fully unrolled with constant indices (the matrix
// stays in registers), then the trial pose; every thread computes its own copy.  Returns 0 when the matrix is not positive definite
// (the trial is then the pose itself and x is not meaningful).
__device__ __forceinline__ int po_solve_trial(const double (&sys)[PO_NV], double lam, const double (&pose)[7], double (&x)[6], double (&trial)[7] PO_ST_PARAM)
{
#pragma clang fp contract(fast)
    double A[36];
    {
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = a; c < 6; ++c, ++idx) { A[a * 6 + c] = sys[idx]; A[c * 6 + a] = sys[idx]; }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) A[j * 7] += lam;
    int ok = 1;
    double inv[6];                                     // 1 / L_jj: six divisions per solve instead of twenty-seven
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d2 = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d2 -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d2 > 0.0)) { ok = 0; d2 = 1.0; }         // keep going on harmless numbers; the result is discarded
        inv[j] = po_rsqrt(d2);
        A[j * 6 + j] = d2 * inv[j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s2 = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s2 -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s2 * inv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s2 = sys[21 + i];
#pragma unroll
        for (int k = 0; k < i; ++k) s2 -= A[i * 6 + k] * x[k];
        x[i] = s2 * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s2 = x[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s2 -= A[k * 6 + i] * x[k];
        x[i] = s2 * inv[i];
    }
    PO_STAMP(4);
    double moved[7];
    po_oplus(pose, x, moved);
#pragma unroll
    for (int i = 0; i < 7; ++i) trial[i] = ok ? moved[i] : pose[i];
    PO_STAMP(5);
    return ok;
}

// ALL_CACHED (every tracker-sized call): pose7 / packed / outlier / n_inliers / done_flag are page-locked HOST memory -- the workgroup
// reads its inputs over PCIe in the prologue and writes the results straight back, then releases `seq` into *done_flag, which the
// calling thread polls (no copy-engine packet on either side, no wait for the end-of-kernel cache flush).
template <bool ALL_CACHED>
__global__ __launch_bounds__(PO_T) void k_pose_optimize(double* pose7, const double* pts, const lpslam_hip_ba_obs* obs, const PoObs* packed, int n, BaCam cam,
                                                       uint8_t* outlier, int* n_inliers, int cache_n, int* done_flag, int seq)
{
#pragma clang fp contract(fast)
    __shared__ PoShared sh;
    extern __shared__ double po_dyn[];
    double* tr = po_dyn;
    PoObs* cache = reinterpret_cast<PoObs*>(po_dyn + PO_NV * PO_TR);
    uint8_t* active = reinterpret_cast<uint8_t*>(cache + cache_n);
    const int tid = threadIdx.x;
#ifdef LPSLAM_PO_STAMPS
    double po_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, po_last = (double)clock64();
#endif
    if (tid < 7) sh.pose[tid] = pose7[tid];
    if (ALL_CACHED) {
        static_assert(sizeof(PoObs) == 7 * sizeof(double), "PoObs is copied as doubles");
        for (int i = tid; i < 7 * n; i += PO_T) reinterpret_cast<double*>(cache)[i] = reinterpret_cast<const double*>(packed)[i];
    }
    for (int k = tid; k < n; k += PO_T) {
        active[k] = 1;
        if (!ALL_CACHED && k < cache_n) {
            const lpslam_hip_ba_obs o = obs[k];
            const double* p = pts + 3 * (size_t)o.point;
            PoObs c; c.u = o.u; c.v = o.v; c.ur = o.ur; c.w = o.inv_sigma2; c.X[0] = p[0]; c.X[1] = p[1]; c.X[2] = p[2];
            cache[k] = c;
        }
    }
    __syncthreads();
    const PoData<ALL_CACHED> d{pts, obs, cache, active, n, cache_n};
    // from here on every thread holds the optimiser's state (pose, trial, system, lambda control) in its own registers
    double pose[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = sh.pose[i];
    int robust = 1, n_bad_last = 0, passes = 0;
    PO_STAMP(0);
    for (int round = 0; round < 4; ++round) {
        // chi2 and linearisation at the round's starting pose (the kernel / the active set may have changed), then trial after
        // trial: every pass evaluates the trial in flight AND linearises at it; g2o's Levenberg control runs between the passes:
        // up to ten iterations, each with up to ten trials of growing lambda.
        double sys[PO_NV], x[6], trial[7];
        po_pass(cam, pose, d, robust, tr, sh PO_ST_ARG);
#pragma unroll
        for (int q = 0; q < PO_NV; ++q) sys[q] = sh.sums[q];
        ++passes;
        double lambda = 1e-5 * fmax(fmax(fmax(fabs(sys[0]), fabs(sys[6])), fmax(fabs(sys[11]), fabs(sys[15]))), fmax(fabs(sys[18]), fabs(sys[20])));
        double ni = 2, current_chi = sys[27];
        int it = 0, qmax = 1;
        PO_STAMP(3);
        int ok = po_solve_trial(sys, lambda, pose, x, trial PO_ST_ARG);
        for (;;) {
            po_pass(cam, trial, d, robust, tr, sh PO_ST_ARG);
            ++passes;
            const double temp = ok ? sh.sums[27] : DBL_MAX;
            double rho = current_chi - temp, scale = 0;
            if (ok) {
#pragma unroll
                for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + sys[21 + j]);
            }
            scale += 1e-3;
            rho *= po_rcp(scale);
            if (rho > 0 && isfinite(temp)) {
                const double t3 = 2 * rho - 1;
                double alpha = 1. - t3 * t3 * t3;
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2;
                current_chi = temp;
#pragma unroll
                for (int i = 0; i < 7; ++i) pose[i] = trial[i];
#pragma unroll
                for (int q = 0; q < PO_NV - 1; ++q) sys[q] = sh.sums[q];     // the pass just made linearised at the accepted pose
            } else {
                lambda *= ni; ni *= 2;
            }
            if (rho < 0 && qmax < 10) ++qmax;                                // another trial of this iteration
            else {
                ++it;
                if (qmax == 10 || rho == 0 || it == 10) break;
                qmax = 1;
            }
            PO_STAMP(3);
#ifdef LPSLAM_PO_DUP_SERIAL
            {   // development: solve + pose update twice
                double lam2 = lambda; asm volatile("" : "+v"(lam2));
                double x2[6], trial2[7];
                const int ok2 = po_solve_trial(sys, lam2, pose, x2, trial2 PO_ST_ARG);
                asm volatile("" :: "v"(trial2[0]), "v"(trial2[6]), "v"(x2[0]), "v"(ok2));
            }
#endif
            ok = po_solve_trial(sys, lambda, pose, x, trial PO_ST_ARG);
        }
        // classification with the plain chi2 of this round's pose
        double R[9];
        po_quat_to_rot(pose, R);
        int bad = 0;
        for (int k = tid; k < n; k += PO_T) {
            double e[3], pc[3], X[3];
            lpslam_hip_ba_obs o;
            d.get(k, o, X);
            const int D = po_residual(cam, R, pose + 4, X, o, e, pc);
            const double chi = o.inv_sigma2 * (e[0] * e[0] + e[1] * e[1] + (D == 3 ? e[2] * e[2] : 0.0));
            const double thr = D == 3 ? 7.81473 : 5.99146;
            const int out = thr < chi ? 1 : 0;
            active[k] = (uint8_t)!out;
            bad += out;
        }
        n_bad_last = (int)po_block_sum((double)bad, sh);
        if (round == 2) robust = 0;
        __syncthreads();
        PO_STAMP(8);
        if (n - n_bad_last < 5) break;
    }
#ifdef LPSLAM_PO_STAMPS
    if (tid == 0) for (int k = 0; k < 16; ++k) g_po_stamps[k] = po_acc[k];
#endif
    for (int k = tid; k < n; k += PO_T) outlier[k] = active[k] ? 0 : 1;       // the last classification made
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) pose7[i] = pose[i];
        n_inliers[0] = n - n_bad_last;
        n_inliers[1] = passes;
    }
    if (done_flag) {
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- the same flow with ONE observation per lane, for up to 256 observations (what a tracked frame has) ---------------------------------
// k_pose_optimize's trial is 2.45 us, of which the 28-value reduction across its wavefronts is 0.8-1.0 (quad step in registers, LDS
// transposition, row reductions).  Here W = 1, 2 or 4 wavefronts carry one observation per lane (n <= 64 W): a pass is bound by the
// FP64 issue rate of a SIMD -- ~350 instructions per observation AND LANE, 1650 cycles whether 1 or 64 lanes are busy -- so a second
// observation per lane costs a second 1650 cycles (round 5's first version: 2.05 us per pass at <= 64 observations, 2.6 at 65-128),
// a second wavefront on the next SIMD costs two workgroup barriers.  Every lane writes its 28 partial sums into a [28][66 W] LDS
// array (consecutive lanes, consecutive words), 2 W lanes per value add 32 partials each in a fixed tree and meet their partners by DPP
// exchanges inside a row, the 28 totals come back to every lane by broadcast reads.  The serial section (lambda control, 6 x 6 solve,
// pose update) is the four-wavefront kernel's code on replicated registers, identical in every wavefront.  The sums' order differs from
// the four-wavefront kernel's (results move in the last bits); which kernel runs depends on the observation count alone.
#define PO_WN_ROW(W) (66 * (W))
template <int W>
__device__ __forceinline__ void po_reduce28_wn(const double (&acc)[PO_NV], double* tr, double* out, double (&sums)[PO_NV])
{
    // column c of a row sits at word c + c / 32 and a row is 66 W words: the 2 W lanes of a value read their 32 partials from chunks that
    // start 33 words apart and neighbouring values 2 W words (mod 32) apart, so the 16 lanes the LDS serves together touch 16 different
    // bank pairs (with rows of 64 W + 1 words the lanes of a value met on ONE bank: W = 2 / 4 measured 2.5 / 3.9 us per pass against 2.1)
    constexpr int ROW = PO_WN_ROW(W);
    constexpr int LPV = 2 * W;                             // lanes per value: neighbours inside a DPP row
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < PO_NV; ++q) tr[q * ROW + tid + (tid >> 5)] = acc[q];
    __syncthreads();                                       // (W = 1: an ordering point for the compiler and the LDS queue, not a wait)
    const int q = tid / LPV, j = tid % LPV;
    double s = 0;
    if (q < PO_NV) {
        const double* row = tr + q * ROW + 33 * j;
        double v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = row[i];
#pragma unroll
        for (int w = 16; w >= 1; w >>= 1)
#pragma unroll
            for (int i = 0; i < w; ++i) v[i] = v[2 * i] + v[2 * i + 1];
        s = v[0];
    }
    s += quad_swap<0xB1>(s);                               // lanes 0<->1, 2<->3
    if (LPV >= 4) s += quad_swap<0x4E>(s);                 // lanes 0<->2, 1<->3
    if (LPV >= 8) s += quad_swap<0x141>(s);                // row_half_mirror: the other quad of the eight
    if (q < PO_NV && j == 0) out[q] = s;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PO_NV; ++i) sums[i] = out[i];
}

// (a device function: the launch is k_pose_optimize_req below -- one workgroup per request of a batch, the same code whether the batch
// holds one tracker's frame or the pending frames of every session of the process, so shared and unshared results are the same bits)
template <int W>
__device__ __forceinline__ void po_wn_body(double* pose7, const PoObs* packed, int n, const BaCam& cam, uint8_t* outlier, int* n_inliers, int* done_flag, int seq,
                                           double* tr, double* out28, PoObs* cache, int (*s_bad)[4])
{
#pragma clang fp contract(fast)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    static_assert(sizeof(PoObs) == 7 * sizeof(double), "PoObs is copied as doubles");
    for (int i = tid; i < 7 * n; i += 64 * W) reinterpret_cast<double*>(cache)[i] = reinterpret_cast<const double*>(packed)[i];      // page-locked host memory, over PCIe
    double pose[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = pose7[i];
    {   // unit quaternion from here on: every pose the passes see is this one or a po_oplus result (normalised there)
        const double rn = po_rsqrt(pose[0] * pose[0] + pose[1] * pose[1] + pose[2] * pose[2] + pose[3] * pose[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pose[i] *= rn;
    }
    __syncthreads();
    bool act = tid < n;                                    // this lane's observation is an inlier of the last classification
    // the lane's observation stays in registers for the whole call (the compiler cannot keep it there itself: the reduction writes LDS
    // between two passes)
    lpslam_hip_ba_obs o;
    double X[3];
    {
        const PoObs c = cache[tid < n ? tid : 0];
        o.pose = 0; o.point = 0; o.u = c.u; o.v = c.v; o.ur = c.ur; o.inv_sigma2 = c.w;
        X[0] = c.X[0]; X[1] = c.X[1]; X[2] = c.X[2];
    }
#ifdef LPSLAM_PO_STAMPS
    double po_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, po_last = (double)clock64();
#endif
    auto pass = [&](const double (&p7)[7], int robust, double (&sums)[PO_NV]) __attribute__((always_inline)) {
        double R[9];
        po_unit_quat_to_rot(p7, R);
        const double t[3] = {p7[4], p7[5], p7[6]};
        double acc[PO_NV];
#pragma unroll
        for (int q = 0; q < PO_NV; ++q) acc[q] = 0;
        if (act) po_accumulate(cam, R, t, o, X, robust, acc);
        PO_STAMP(1);
        po_reduce28_wn<W>(acc, tr, out28, sums);
        PO_STAMP(2);
#ifdef LPSLAM_PO_STAMPS
        if (tid == 0) po_acc[15] += 1;
#endif
    };
    int robust = 1, n_bad_last = 0, passes = 0;
    double sys[PO_NV];
    double chi_kept = 0;                                   // the (robustified) chi2 at `pose` the previous round ended with
    bool reuse = false;                                    // the round starts from the sums the previous one ended with
    for (int round = 0; round < 4; ++round) {
        // (the control flow of k_pose_optimize: g2o's Levenberg between the passes, up to ten iterations of up to ten trials; every
        // wavefront takes the same decisions from the same sums, so the barriers inside the passes match)
        double got[PO_NV], x[6], trial[7];
        // A round opens with chi2 and the linearisation at its starting pose.  When the classification changed nothing and the kernel is
        // the same, that is what the previous round ended with -- sys holds the sums at `pose` (an accepted trial's pass, or the ones the
        // rejected trials left standing), chi_kept its chi2: the same numbers a pass would produce, so the pass is not made.
        if (reuse) sys[27] = chi_kept;
        else { pass(pose, robust, sys); ++passes; }
        double lambda = 1e-5 * fmax(fmax(fmax(fabs(sys[0]), fabs(sys[6])), fmax(fabs(sys[11]), fabs(sys[15]))), fmax(fabs(sys[18]), fabs(sys[20])));
        double ni = 2, current_chi = sys[27];
        int it = 0, qmax = 1;
        PO_STAMP(3);
        int ok = po_solve_trial(sys, lambda, pose, x, trial PO_ST_ARG);
        for (;;) {
            pass(trial, robust, got);
            ++passes;
            const double temp = ok ? got[27] : DBL_MAX;
            double rho = current_chi - temp, scale = 0;
            if (ok) {
#pragma unroll
                for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + sys[21 + j]);
            }
            scale += 1e-3;
            rho *= po_rcp(scale);
            if (rho > 0 && isfinite(temp)) {
                const double t3 = 2 * rho - 1;
                double alpha = 1. - t3 * t3 * t3;
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2;
                current_chi = temp;
#pragma unroll
                for (int i = 0; i < 7; ++i) pose[i] = trial[i];
#pragma unroll
                for (int q = 0; q < PO_NV - 1; ++q) sys[q] = got[q];         // the pass just made linearised at the accepted pose
            } else {
                lambda *= ni; ni *= 2;
            }
            if (rho < 0 && qmax < 10) ++qmax;                                // another trial of this iteration
            else {
                ++it;
                if (qmax == 10 || rho == 0 || it == 10) break;
                qmax = 1;
            }
            PO_STAMP(3);
            ok = po_solve_trial(sys, lambda, pose, x, trial PO_ST_ARG);
        }
        PO_STAMP(3);
        // classification with the plain chi2 of this round's pose
        double R[9];
        po_unit_quat_to_rot(pose, R);
        int is_out = 0;
        if (tid < n) {
            double e[3], pc[3];
            const int D = po_residual(cam, R, pose + 4, X, o, e, pc);
            const double chi = o.inv_sigma2 * (e[0] * e[0] + e[1] * e[1] + (D == 3 ? e[2] * e[2] : 0.0));
            const double thr = D == 3 ? 7.81473 : 5.99146;
            is_out = thr < chi ? 1 : 0;
        }
        const bool act_new = tid < n && !is_out;
        int bad = __popcll(__ballot(is_out)) | (__ballot(act_new != act) ? 1 << 16 : 0);       // outliers | "some lane changed sides"
        act = act_new;
        if (W > 1) {
            if (lane == 0) s_bad[round & 1][wave] = bad;
            __syncthreads();
            bad = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) { const int b2 = s_bad[round & 1][w]; bad = ((bad & 0xffff) + (b2 & 0xffff)) | ((bad | b2) & (1 << 16)); }
        }
        n_bad_last = bad & 0xffff;
        chi_kept = current_chi;
        reuse = !(bad >> 16) && round != 2;                // (after round 2 the kernel changes: plain chi2)
        if (round == 2) robust = 0;
        PO_STAMP(8);
        if (n - n_bad_last < 5) break;
    }
#ifdef LPSLAM_PO_STAMPS
    if (tid == 0) for (int k = 0; k < 16; ++k) g_po_stamps[k] = po_acc[k];
#endif
    if (tid < n) outlier[tid] = act ? 0 : 1;               // the last classification made
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) pose7[i] = pose[i];
        n_inliers[0] = n - n_bad_last;
        n_inliers[1] = passes;
    }
    if (done_flag) {
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One workgroup per request.  A request is a page-locked block of the caller: pose (in / out) at 0, inlier count and passes at 56 / 60,
// the done flag at 64, the camera at 72, the packed observations at 128, the outlier bytes behind them (lpslam_hip_pose_optimize lays
// it out).  Block pointer, observation count and sequence number come by value; W = 1, 2 or 4 wavefronts work on a request
// (n <= 64 W), the others of the workgroup leave at once -- a barrier counts the wavefronts that are left.
constexpr int PO_MAX_BATCH = 32;
struct PoBatch { uint8_t* blk[PO_MAX_BATCH]; int n[PO_MAX_BATCH]; int seq[PO_MAX_BATCH]; };
constexpr size_t PO_BLK_INLIERS = 56, PO_BLK_FLAG = 64, PO_BLK_CAM = 72, PO_BLK_PACKED = 128;
static_assert(PO_BLK_CAM + sizeof(BaCam) <= PO_BLK_PACKED, "pose-optimiser block header");
__global__ __launch_bounds__(256) void k_pose_optimize_req(PoBatch b)
{
    // LDS by the launch (po_lds_bytes of the largest request of the batch): the transposition array [28][66 W], the 28 totals, the
    // observations, the wavefronts' outlier counts.  A workgroup that asks for the four-wavefront layout (74 KB) whatever it needs found
    // no compute unit to start on while an extraction kernel's workgroups held theirs (8 x 13 KB of 160): requests of <= 128
    // observations -- 92 % of a tracker's -- now ask for 37 KB or 19.
    extern __shared__ __attribute__((aligned(16))) double po_lds[];
    const int wmax = (int)blockDim.x >> 6;
    double* tr = po_lds;
    double* out28 = tr + PO_NV * 66 * wmax;
    PoObs* cache = reinterpret_cast<PoObs*>(out28 + 32);
    int (*s_bad)[4] = reinterpret_cast<int (*)[4]>(cache + 64 * wmax);      // outliers per wavefront, double-buffered over the rounds
    const int r = blockIdx.x, n = b.n[r];
    uint8_t* blk = b.blk[r];
    const int W = n <= 64 ? 1 : (n <= 128 ? 2 : 4);
    if ((int)threadIdx.x >= 64 * W) return;
    BaCam cam;
    {
        const double* cp = reinterpret_cast<const double*>(blk + PO_BLK_CAM);
        cam.fx = cp[0]; cam.fy = cp[1]; cam.cx = cp[2]; cam.cy = cp[3]; cam.fxb = cp[4]; cam.hub_mono = cp[5]; cam.hub_stereo = cp[6];
    }
    double* pose7 = reinterpret_cast<double*>(blk);
    const PoObs* packed = reinterpret_cast<const PoObs*>(blk + PO_BLK_PACKED);
    uint8_t* outlier = blk + PO_BLK_PACKED + (size_t)(n > 0 ? n : 1) * sizeof(PoObs);
    int* n_inliers = reinterpret_cast<int*>(blk + PO_BLK_INLIERS);
    int* flag = reinterpret_cast<int*>(blk + PO_BLK_FLAG);
    if (W == 1) po_wn_body<1>(pose7, packed, n, cam, outlier, n_inliers, flag, b.seq[r], tr, out28, cache, s_bad);
    else if (W == 2) po_wn_body<2>(pose7, packed, n, cam, outlier, n_inliers, flag, b.seq[r], tr, out28, cache, s_bad);
    else po_wn_body<4>(pose7, packed, n, cam, outlier, n_inliers, flag, b.seq[r], tr, out28, cache, s_bad);
}

#include "ba_update.inl"
#include "ba_build.inl"

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct lpslam_hip_ba {
    lpslam_hip_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;                       // orders this problem's stream before a batch that runs on another one
    int n_poses = 0, n_points = 0, n_obs = 0, n_free = 0, dim = 0, dim_pad = 0, n_blocks = 0;
    // one block of the context's cache holds everything (carved at creation); the members below point into it
    void* block = nullptr; size_t block_cap = 0;
    BaView h_view{};                               // host copy of the device-resident view
    BaView* d_view = nullptr;
    double *d_poses[2] = {nullptr, nullptr}, *d_points[2] = {nullptr, nullptr};
    uint8_t* d_o_active = nullptr; uint8_t* d_act_in = nullptr; int* d_o_orig = nullptr;
    double* d_red = nullptr; int64_t red_n = 0;
    double* d_scal = nullptr;
    double* d_chi_obs = nullptr; uint8_t* d_depth = nullptr;
    BaCtl* d_ctl = nullptr; lpslam_hip_ba_iter_log* d_log = nullptr;
    std::vector<double> h_ur;                      // mono/stereo classification for the outlier thresholds
    BaCtl h_ctl{};                                 // last control block read back
    struct Pinned { BaCtl ctl; lpslam_hip_ba_iter_log log[MAX_LOG]; };
    Pinned* pin = nullptr;                         // page-locked: control block and iteration log come back in one round trip
    void* stage = nullptr; size_t stage_cap = 0;   // page-locked staging of the creation inputs, handed back at the first synchronisation
    // page-locked, device-mapped exchange block [poses in | points in | poses out | points out]: the small per-keyframe transfers
    // (set_state, get, control block) are done by KERNELS that read / write host memory over PCIe, not by the DMA engines -- a
    // hipMemcpyAsync of a few KB queues behind whatever the engine is busy with (the 0.9 MB image uploads of the front end:
    // +0.08 ms per keyframe, measured) and costs a packet round trip of its own even on an idle engine
    uint8_t* xfer = nullptr; size_t xfer_cap = 0;
    hipEvent_t xfer_in_read = nullptr; bool xfer_in_pending = false;      // the kernel that reads the "in" half has been enqueued
    int robust = 1, points_fixed = 0;
    int pending_iters = -1;                            // >= 0 between optimize_begin and optimize_end
    int band_hbw_structure = -1;                       // block half-bandwidth of the window when the band path can take it (creation), else -1
    int band_gmax = 0;                                 // landmarks per group at most (LDS of k_schur_group)
    int faults_band = 0, faults_update = 0;            // timed-out hand-overs seen so far (report_faults)
    bool quiesced = false;                             // everything enqueued for this problem is known to be complete (a shared batch waited for it): destroy need not wait for its stream again
    bool built = false;                                // the structure build has been enqueued (lpslam_hip_ba_build_batch); prepare alone leaves the block untouched
    void* build_desc = nullptr;                        // BuildDesc of this problem (host copy), ba_build.inl
    size_t o_descs = 0;                                // offset of the descriptor array (device: in the block; host: in the staging block)
    hipEvent_t ev_built = nullptr;                     // a build enqueued on another problem's stream: this problem's stream waits for it
};

namespace {

// what a launch chain needs to know: the view array, how many problems it holds and the launch extents (maxima over them)
struct BaLaunch {
    const BaView* d_views = nullptr; int count = 0; hipStream_t s = nullptr; lpslam_hip_ctx* ctx = nullptr;
    int obs_blocks = 0, pose_blocks = 0, point_blocks = 0, part_n = 0, n_free = 0, n_blocks = 0, dim = 0, nb = 0, land_blocks = 0, n_poses = 0, schur_items = 0;
    int robust = 1, points_fixed = 0;
    bool any_small = false, any_large = false;          // systems for k_chol_wg / for the panel-pair chain
    bool any_band = false, any_dense = false;           // banded windows (ba_band.inl) / pair lists + dense factorisation
    int band_groups = 0, band_blocks = 0, band_gmax = 0; // extents of k_schur_group / k_schur_band_reduce, landmarks per group
    bool spread = false;                                // the context reserves CUs of every XCD for the solves: no XCD pinning
    bool any_one_pass = false, any_two_launch = false;  // problems that take k_ba_update behind the fused solve / that keep k_ba_backsub + k_ba_trial (upd_takes)
    // profiled run (lpslam_hip_ba_optimize_profiled): an event after every launch, tagged with the kernel it closes
    std::vector<std::pair<hipEvent_t, int>>* marks = nullptr;
    void mark(int kernel) const
    {
        if (!marks) return;
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, s);
        marks->emplace_back(e, kernel);
    }
    void add(const lpslam_hip_ba* b)
    {
        const BaView& v = b->h_view;
        obs_blocks = std::max(obs_blocks, v.obs_blocks); pose_blocks = std::max(pose_blocks, v.pose_blocks);
        point_blocks = std::max(point_blocks, v.point_blocks); part_n = std::max(part_n, v.part_n); land_blocks = std::max(land_blocks, v.land_blocks);
        n_free = std::max(n_free, v.n_free); n_poses = std::max(n_poses, v.n_poses);
        if (upd_takes(v.n_points, v.n_free, v.n_poses)) any_one_pass = true; else any_two_launch = true;
        if (v.band_hbw >= 0) {
            any_band = true;
            band_groups = std::max(band_groups, v.band_groups); band_blocks = std::max(band_blocks, v.n_free * (v.band_hbw + 2));
            band_gmax = std::max(band_gmax, b->band_gmax);
        } else {
            if (v.n_free) any_dense = true;
            n_blocks = std::max(n_blocks, v.n_blocks); dim = std::max(dim, v.dim); schur_items = std::max(schur_items, v.n_poses * SPLIT + (v.extra_pack >> 12) + v.n_blocks);
            nb = std::max(nb, v.dim_pad / NB);
            if (v.dim > 0) { if (cw_fits(v.dim)) any_small = true; else any_large = true; }
        }
        ++count;
    }
};
BaLaunch single_launch(lpslam_hip_ba* b)
{
    BaLaunch L;
    L.d_views = b->d_view; L.s = b->stream; L.ctx = b->ctx; L.robust = b->robust; L.points_fixed = b->points_fixed;
    L.add(b);
    // a small reserve (<= 8 CUs of every XCD) cannot hold the pinned chain's workgroups on ONE XCD: spread them over all XCDs then (4 CUs:
    // 4256 against 4145 frames/s pinned); from 12 on the pinned chain is the better one again (12: 4392 against 4336, 16: 4428 against 4357)
    L.spread = b->ctx && b->ctx->reserve_cus > 0 && b->ctx->reserve_cus <= 8;
    static const int spread_env = [] { const char* e = getenv("LPSLAM_HIP_BA_SPREAD"); return e ? atoi(e) : -1; }();      // measurements: 0 / 1 force the placement
    if (spread_env >= 0) L.spread = spread_env != 0;
    return L;
}

// linearisation of the accepted state (skipped on the device when the previous trial was rejected)
int enqueue_linearize(const BaLaunch& L, int fused, bool explicit_lin = true)
{
    // fused solve: only the first unit of an optimize() call linearises here; every later state is linearised COMPLETELY beside its
    // trial (k_ba_trial: observation side with the landmark sums, pose side, combine), so later units start at the Schur complement
    if (!explicit_lin) return LPSLAM_HIP_OK;
    hipLaunchKernelGGL(k_ba_lin, dim3(L.obs_blocks + L.pose_blocks, L.count), dim3(256), 0, L.s, L.d_views, L.robust, L.points_fixed); L.mark(LPSLAM_HIP_BA_K_LIN);
    hipLaunchKernelGGL(k_ba_point_sum, dim3(L.point_blocks + 1, L.count), dim3(256), 0, L.s, L.d_views, fused);      // + the workgroup that combines the pose partials
    L.mark(LPSLAM_HIP_BA_K_POINT_SUM);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

// Schur complement for the device's current lambda into the reduced buffer
int enqueue_reduce(const BaLaunch& L, int fused)
{
    if (L.any_dense) hipLaunchKernelGGL(k_ba_schur, dim3(L.schur_items, L.count), dim3(64), 0, L.s, L.d_views, fused, L.robust);
    if (L.any_band) {
        if (bd_set_attributes() != hipSuccess) return LPSLAM_HIP_ERR_DEVICE;
        // The band reduction as trailing workgroups of the group launch (LPSLAM_HIP_BA_REDUCE_IN_SCHUR=1) was measured and is OFF: the
        // groups' shares (3.1 MB per trial) then cross from workgroup to workgroup inside one launch, which on this part means
        // write-through stores and L2-bypassing loads -- 38.2 us for the one launch against 14.7 + 9.3 us for the two (MI355X, config 3).
        static const bool reduce_in_schur_env = [] { const char* e = getenv("LPSLAM_HIP_BA_REDUCE_IN_SCHUR"); return e && atoi(e) != 0; }();
        const int reduce_here = (fused && reduce_in_schur_env) ? 1 : 0;
        // a batch fills the chip with group workgroups: the variant that fits two of them on a compute unit (128 registers; the pose side
        // spills a few values there -- it is off the path) took a batch of 16 contiguous windows from 2.87 to 2.65 ms per 10 iterations; a single window keeps the
        // variant without spills (its pose-side workgroups are its longest).  Same arithmetic, same bytes.
        const dim3 sg_grid(L.n_poses + std::max(L.band_groups, 1) + (reduce_here ? (L.band_blocks + 1) / 2 : 0), L.count);
        const size_t sg_lds = std::max(bd_lds_bytes(L.band_gmax), (size_t)4096);
        if (L.count >= 4) hipLaunchKernelGGL(k_schur_group<4>, sg_grid, dim3(BD_THREADS), sg_lds, L.s, L.d_views, fused, L.robust, reduce_here);
        else hipLaunchKernelGGL(k_schur_group<1>, sg_grid, dim3(BD_THREADS), sg_lds, L.s, L.d_views, fused, L.robust, reduce_here);
        L.mark(LPSLAM_HIP_BA_K_SCHUR);
        if (!reduce_here) {
            hipLaunchKernelGGL(k_schur_band_reduce, dim3(L.band_blocks, L.count), dim3(256), 0, L.s, L.d_views, fused);
            L.mark(LPSLAM_HIP_BA_K_BAND_REDUCE);
        }
    } else if (L.any_dense) L.mark(LPSLAM_HIP_BA_K_SCHUR);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

// factor + solve, update into the trial state, trial chi2 and scale terms (+ the lambda control when fused)
int enqueue_solve(const BaLaunch& L, int fused)
{
    hipStream_t s = L.s;
    if (L.any_band) {
        if (bd_set_attributes() != hipSuccess) return LPSLAM_HIP_ERR_DEVICE;
        hipLaunchKernelGGL(k_chol_band, dim3(2, L.count), dim3(BC_THREADS), BC_LDS_BYTES, s, L.d_views);      // workgroup 0: the bottom-up helper of a twisted factorisation
        if (!L.any_dense) L.mark(LPSLAM_HIP_BA_K_CHOL);
    }
    if (L.dim > 0) {
        if (!fused) {
            hipLaunchKernelGGL(k_lm_begin, dim3(1, L.count), dim3(64), 0, s, L.d_views);
            hipLaunchKernelGGL(k_chol_prep, dim3((L.nb * NB + 255) / 256, L.count), dim3(256), 0, s, L.d_views);
        }
        const bool wg = L.count >= cw_min_batch();
        if (wg && L.any_small && L.ctx) L.ctx->ba_wg_launches.fetch_add(1);
        if (L.marks && !(wg && L.any_small)) {              // profiled run through the panel-pair chain: factorisation and solve timed apart
            enqueue_cholesky(s, L.d_views, L.count, L.nb, 0, L.spread);
            L.mark(LPSLAM_HIP_BA_K_CHOL);
            enqueue_xsolve(s, L.d_views, L.count, L.dim, 0, L.spread);
            L.mark(LPSLAM_HIP_BA_K_XSOLVE);
        } else {
            enqueue_factor_solve(s, L.d_views, L.count, L.nb, L.dim, wg, L.any_small, L.any_large, L.spread);
            L.mark(LPSLAM_HIP_BA_K_CHOL);
        }
    } else if (!fused) {
        hipLaunchKernelGGL(k_lm_begin, dim3(1, L.count), dim3(64), 0, s, L.d_views);
    }
    static const bool two_launch_env = [] { const char* e = getenv("LPSLAM_HIP_BA_TWO_LAUNCH_UPDATE"); return e && atoi(e) != 0; }();      // measurements: the round-4 form
    const bool one_pass = fused && !two_launch_env;
    if (one_pass && L.any_one_pass) {
        // back substitution, trial state, its chi2 and complete linearisation, the lambda control: one launch (ba_update.inl)
        hipLaunchKernelGGL(k_ba_update, dim3(L.land_blocks, L.count), dim3(256), 0, s, L.d_views, L.robust, L.points_fixed);
        L.mark(LPSLAM_HIP_BA_K_TRIAL);
        LP_HIP(hipGetLastError());
        if (!L.any_two_launch) return LPSLAM_HIP_OK;
    }
    // (the problems k_ba_update does not take -- no landmarks, no free keyframe, more keyframes than its LDS holds -- and the partitioned solve)
    hipLaunchKernelGGL(k_ba_backsub, dim3(L.part_n + 1, L.count), dim3(256), 0, s, L.d_views, one_pass ? 1 : 0);
    L.mark(LPSLAM_HIP_BA_K_BACKSUB);
    {
        // fused solve: the trial launch also linearises the trial state on speculation (observation side + pose side)
        const int spec = fused ? 1 : 0;
        hipLaunchKernelGGL(k_ba_trial, dim3(L.pose_blocks + (spec ? L.land_blocks + L.pose_blocks : 0), L.count), dim3(256), 0, s, L.d_views, L.robust, fused, L.points_fixed, spec, one_pass ? 1 : 0);
        L.mark(LPSLAM_HIP_BA_K_TRIAL);
    }
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

// ---- control block on the device: armed, reset and collected by kernels (one launch for any number of problems) -------------
// arms the control block for an optimize() call of `iters` outer iterations (g2o: lambda_0 is recomputed per call)
__global__ __launch_bounds__(64) void k_ba_arm(const BaView* __restrict__ views, int iters)
{
    BA_VIEW(v);
    if (threadIdx.x != 0) return;
    BaCtl c = *v.ctl;
    c.max_outer = iters; c.outer_done = 0; c.need_lin = 1; c.first = 1; c.qmax = 0; c.stopped = 0; c.ni = 2; c.rho = 0; c.last_accepted = 0;
    c.ticket = 0; c.spec = 0; c.cur_launch = c.cur;
    *v.ctl = c;
    ba_sync_words(v)[3] = 0;              // the call starts with an explicit linearisation (pose side included)
    ba_sync_words(v)[2] = 0; ba_sync_words(v)[4] = 0;      // counts of the Schur launch (groups / pose-side wavefronts that have stored)
}
// state given at creation back into buffer 0, every observation active, LM state cleared
__global__ __launch_bounds__(256) void k_ba_reset(const BaView* __restrict__ views)
{
    BA_VIEW(v);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 7 * v.n_poses) v.poses_buf[0][i] = v.poses0[i];
    if (i < 3 * v.n_points) v.points_buf[0][i] = v.points0[i];
    if (i < v.n_obs) v.o_active[i] = 1;
    if (i == 0) {
        BaCtl c{};
        c.ni = 2; c.need_lin = 1; c.first = 1;
        *v.ctl = c;
    }
}
// control block + iteration log of every problem of a batch into one contiguous buffer (one copy to the host)
constexpr int COLLECT_STRIDE = (int)((sizeof(BaCtl) + MAX_LOG * sizeof(lpslam_hip_ba_iter_log) + 15) / 16 * 16);
__global__ __launch_bounds__(64) void k_ba_collect(const BaView* __restrict__ views, uint8_t* out, int n_log)
{
    BA_VIEW(v);
    uint8_t* dst = out + (size_t)blockIdx.y * COLLECT_STRIDE;
    const int* src_c = reinterpret_cast<const int*>((const BaCtl*)v.ctl);
    int* dst_c = reinterpret_cast<int*>(dst);
    const int* sw = ba_sync_words(v);
    for (int i = threadIdx.x; i < (int)(sizeof(BaCtl) / 4); i += 64)
        dst_c[i] = i == (int)(offsetof(BaCtl, faults_band) / 4) ? sw[0] : (i == (int)(offsetof(BaCtl, faults_update) / 4) ? sw[1] : src_c[i]);
    const int* src_l = reinterpret_cast<const int*>((const lpslam_hip_ba_iter_log*)v.log);
    int* dst_l = reinterpret_cast<int*>(dst + sizeof(BaCtl));
    const int words = min(n_log, MAX_LOG) * (int)(sizeof(lpslam_hip_ba_iter_log) / 4);
    for (int i = threadIdx.x; i < words; i += 64) dst_l[i] = src_l[i];
}

// host (page-locked, device-mapped) -> device by load / store: 16 bytes per lane and round, grid-stride
__global__ __launch_bounds__(256) void k_copy_from_host(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// current state -> page-locked host memory (kernel stores over PCIe: no DMA packet, no engine queue)
__global__ __launch_bounds__(256) void k_ba_state_to_host(const BaView* __restrict__ views, double* poses, double* points)
{
    BA_VIEW(v);
    const int cur = v.ctl->cur;
    // two doubles per thread: 16-byte stores (a PCIe write per 8 bytes made this kernel 14.7 us for 123 KB)
    const int i = 2 * (blockIdx.x * 256 + threadIdx.x);
    const double* sp = sel2(v.poses_buf[0], v.poses_buf[1], cur); const double* sx = sel2(v.points_buf[0], v.points_buf[1], cur);
    const int np = 7 * v.n_poses, nx = 3 * v.n_points;
    if (poses && i + 1 < np) *reinterpret_cast<f64x2*>(poses + i) = f64x2{sp[i], sp[i + 1]}; else if (poses && i < np) poses[i] = sp[i];
    if (points && i + 1 < nx) *reinterpret_cast<f64x2*>(points + i) = f64x2{sx[i], sx[i + 1]}; else if (points && i < nx) points[i] = sx[i];
}
// k_ba_reset with new creation-time values read from page-locked host memory (lpslam_hip_ba_set_state)
__global__ __launch_bounds__(256) void k_ba_reset_from_host(const BaView* __restrict__ views, const double* poses, const double* points)
{
    BA_VIEW(v);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 7 * v.n_poses) { const double x = poses ? poses[i] : v.poses0[i]; if (poses) const_cast<double*>((const double*)v.poses0)[i] = x; v.poses_buf[0][i] = x; }
    if (i < 3 * v.n_points) { const double x = points ? points[i] : v.points0[i]; if (points) const_cast<double*>((const double*)v.points0)[i] = x; v.points_buf[0][i] = x; }
    if (i < v.n_obs) v.o_active[i] = 1;
    if (i == 0) {
        BaCtl c{};
        c.ni = 2; c.need_lin = 1; c.first = 1;
        *v.ctl = c;
    }
}

uint8_t* ensure_xfer(lpslam_hip_ba* b)
{
    if (b->xfer) return b->xfer;
    const size_t bytes = (2 * (7 * (size_t)b->n_poses + 3 * (size_t)std::max(b->n_points, 1)) + 4) * sizeof(double);
    b->xfer = (uint8_t*)lp_pin_big_alloc(b->ctx, bytes, &b->xfer_cap);
    if (b->xfer && hipEventCreateWithFlags(&b->xfer_in_read, hipEventDisableTiming) != hipSuccess) { lp_pin_big_free(b->ctx, b->xfer, b->xfer_cap); b->xfer = nullptr; }
    return b->xfer;
}

void release_stage(lpslam_hip_ba* b)
{
    if (b->stage) { lp_pin_big_free(b->ctx, b->stage, b->stage_cap); b->stage = nullptr; b->stage_cap = 0; }
}
// control block (and, with log_entries > 0, that many entries of the iteration log) to the host: one synchronisation
int read_ctl(lpslam_hip_ba* b, int log_entries = 0)
{
    if (b->pin) {
        // one small kernel stores the control block and the log entries straight into the page-locked block
        static_assert(offsetof(lpslam_hip_ba::Pinned, log) == sizeof(BaCtl), "k_ba_collect writes the log right behind the control block");
        hipLaunchKernelGGL(k_ba_collect, dim3(1, 1), dim3(64), 0, b->stream, b->d_view, (uint8_t*)b->pin, std::max(log_entries, 0));
        LP_HIP(hipGetLastError());
        LP_HIP(hipStreamSynchronize(b->stream));
        b->h_ctl = b->pin->ctl;
        release_stage(b);
        return LPSLAM_HIP_OK;
    }
    int sw[2] = {0, 0};
    LP_HIP(hipMemcpyAsync(&b->h_ctl, b->d_ctl, sizeof(BaCtl), hipMemcpyDeviceToHost, b->stream));
    LP_HIP(hipMemcpyAsync(sw, b->d_scal + 8, sizeof(sw), hipMemcpyDeviceToHost, b->stream));
    LP_HIP(hipStreamSynchronize(b->stream));
    b->h_ctl.faults_band = sw[0]; b->h_ctl.faults_update = sw[1];
    release_stage(b);
    return LPSLAM_HIP_OK;
}

// A hand-over between workgroups that timed out -- the two chains of the twisted band factorisation, the keyframe blocks of
// k_ba_update -- leaves a result that must not be used: the call that sees new time-outs in the collected control block fails with
// the reason.  The late chain may have left the band factorisation's flags set: they are cleared and the problem solves dense from
// here on (a stale flag would let the next launch merge blocks that are not there yet).
int report_faults(lpslam_hip_ba* b)
{
    const int nb = b->h_ctl.faults_band - b->faults_band, nu = b->h_ctl.faults_update - b->faults_update;
    if (nb <= 0 && nu <= 0) return LPSLAM_HIP_OK;
    b->faults_band = b->h_ctl.faults_band; b->faults_update = b->h_ctl.faults_update;
    if (b->ctx) { b->ctx->ba_timeouts_band.fetch_add(std::max(nb, 0)); b->ctx->ba_timeouts_update.fetch_add(std::max(nu, 0)); }
    if (nb > 0) {
        (void)hipMemsetAsync((void*)b->h_view.blk_ticket, 0, 2 * sizeof(int), b->stream);
        if (b->h_view.band_hbw >= 0) (void)lpslam_hip_ba_set_solver(b, LPSLAM_HIP_BA_SOLVER_DENSE);
    }
    set_error("bundle adjustment: %d band-factorisation and %d update hand-over(s) between workgroups timed out; this call's result is not valid", std::max(nb, 0), std::max(nu, 0));
    return LPSLAM_HIP_ERR_DEVICE;
}

int begin_optimize(lpslam_hip_ba* b, int robust, int iters)
{
    b->robust = robust;
    hipLaunchKernelGGL(k_ba_arm, dim3(1, 1), dim3(64), 0, b->stream, b->d_view, iters);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

// Host-side plan of the band path (ba_band.inl), made at creation from the caller's observation list.
struct BandPlan {
    int hbw = -1;                               // block half-bandwidth; -1: the window does not qualify
    std::vector<int> order, qinfo, bstart;      // landmarks with free observations by (first slot, id); (f0 << 8 | index in group); entry offsets
    std::vector<int> groups, glo, ghi;          // BD_REC ints per group; per free slot: first / last group that can touch it
    void build(const lpslam_hip_ba_obs* obs, int n_obs, int n_points, const int* slot, int n_free, int dim, const int* deg, int gmax)
    {
        if (n_free < 1 || n_obs < 1 || !bc_fits(dim)) return;
        std::vector<int> fmin((size_t)n_points, INT32_MAX), fmax((size_t)n_points, -1);
        for (int k = 0; k < n_obs; ++k) {
            const int sl = slot[obs[k].pose], j = obs[k].point;
            if (sl < 0) continue;
            fmin[j] = std::min(fmin[j], sl); fmax[j] = std::max(fmax[j], sl);
        }
        int h = 0;
        std::vector<int> first_count((size_t)n_free + 1, 0);
        for (int j = 0; j < n_points; ++j) if (fmax[j] >= 0) { h = std::max(h, fmax[j] - fmin[j]); first_count[(size_t)fmin[j] + 1]++; }
        if (h > BD_MAXHBW) return;
        for (int i = 0; i < n_free; ++i) first_count[(size_t)i + 1] += first_count[(size_t)i];
        order.resize((size_t)first_count[(size_t)n_free]);
        {
            std::vector<int> at(first_count.begin(), first_count.end() - 1);
            for (int j = 0; j < n_points; ++j) if (fmax[j] >= 0) order[(size_t)at[(size_t)fmin[j]]++] = j;      // counting sort: ties stay in landmark order
        }
        const int n_ord = (int)order.size();
        qinfo.resize((size_t)n_ord); bstart.resize((size_t)n_ord + 1);
        int e = 0;
        for (int q = 0; q < n_ord;) {
            const int f0 = fmin[order[(size_t)q]];
            int last = f0, cnt = 0, e0 = e;
            while (q + cnt < n_ord && cnt < gmax) {
                const int j = order[(size_t)(q + cnt)];
                const int l2 = std::max(last, fmax[j]);
                if (l2 - f0 + 1 > BD_MAXKF) break;
                last = l2;
                qinfo[(size_t)(q + cnt)] = (f0 << 8) | cnt; bstart[(size_t)(q + cnt)] = e;
                e += deg[j]; ++cnt;
            }
            const int rec[BD_REC] = {e0, e, f0, cnt, 6 * (last - f0 + 1), 0, 0, 0};
            groups.insert(groups.end(), rec, rec + BD_REC);
            q += cnt;
        }
        bstart[(size_t)n_ord] = e;
        // a group touches slots f0 .. f0 + rows / 6 - 1; groups are sorted by f0, so the candidates of a block (i, k), k <= i, are those
        // with f0 > i - BD_MAXKF (first: glo[i]) and f0 <= k (last: ghi[k]); the kernel tests the cover itself
        const int n_grp = (int)groups.size() / BD_REC;
        glo.assign((size_t)n_free, n_grp); ghi.assign((size_t)n_free, -1);
        for (int i = 0, g = 0; i < n_free; ++i) { while (g < n_grp && groups[(size_t)BD_REC * g + 2] <= i - BD_MAXKF) ++g; glo[(size_t)i] = g; }
        for (int i = 0, g = -1; i < n_free; ++i) { while (g + 1 < n_grp && groups[(size_t)BD_REC * (g + 1) + 2] <= i) ++g; ghi[(size_t)i] = g; }
        hbw = h;
    }
};

struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; }
};

}  // namespace

extern "C" {

// Creation in two halves.  lpslam_hip_ba_prepare is the HOST half: validation, the window's shape (band plan, landmark blocks), one block of
// the context's cache carved, the inputs copied into a page-locked staging block -- no kernel, no stream operation, safe to call from
// several threads at once (a server's sessions prepare their windows side by side).  lpslam_hip_ba_build_batch is the DEVICE half for
// any number of prepared problems: ~16 launches in all (blockIdx.y = problem) on the first problem's stream.  lpslam_hip_ba_create =
// prepare + build_batch of one.
int lpslam_hip_ba_prepare(lpslam_hip_ctx* ctx, const double* poses, const uint8_t* fixed, int32_t n_poses, const double* points,
                          int32_t n_points, const lpslam_hip_ba_obs* obs, int32_t n_obs, const lpslam_hip_ba_camera* cam,
                          lpslam_hip_ba** out)
{
    if (!ctx || !poses || !points || !obs || !cam || !out || n_poses < 1 || n_points < 0 || n_obs < 0) {
        set_error("invalid bundle-adjustment arguments"); return LPSLAM_HIP_ERR_INVALID;
    }
    *out = nullptr;
    // range check + landmark degrees (bound of the pair lists: every ordered pair of observations of a landmark)
    std::vector<int> deg((size_t)std::max(n_points, 1), 0);
    for (int k = 0; k < n_obs; ++k) {
        if (obs[k].pose < 0 || obs[k].pose >= n_poses || obs[k].point < 0 || obs[k].point >= n_points) {
            set_error("observation %d references pose %d / point %d out of range", k, obs[k].pose, obs[k].point);
            return LPSLAM_HIP_ERR_INVALID;
        }
        deg[obs[k].point]++;
    }
    size_t terms_cap = 0;
    for (int j = 0; j < n_points; ++j) { terms_cap += (size_t)deg[j] * deg[j]; if (deg[j] > 0xFFFF) { set_error("landmark %d has more than 65535 observations", j); return LPSLAM_HIP_ERR_INVALID; } }
    LP_HIP(hipSetDevice(ctx->cfg.device));
    lpslam_hip_ba* b = new lpslam_hip_ba();
    b->ctx = ctx;
    static_assert(sizeof(lpslam_hip_ba::Pinned) <= 8192, "pinned block size");
    b->pin = static_cast<lpslam_hip_ba::Pinned*>(lp_pin_alloc(ctx));      // nullptr: the pageable path stays
    // own stream (from the context's cache): a bundle adjustment runs beside the front end of later frames
    b->stream = lp_stream_acquire(ctx);
    if (!b->stream) { lpslam_hip_ba_destroy(b); set_error("hipStreamCreate failed"); return LPSLAM_HIP_ERR_DEVICE; }
    b->n_poses = n_poses; b->n_points = n_points; b->n_obs = n_obs;
    std::vector<int> slot(n_poses), free_pose;
    for (int i = 0; i < n_poses; ++i) { if (fixed && fixed[i]) slot[i] = -1; else { slot[i] = (int)free_pose.size(); free_pose.push_back(i); } }
    b->n_free = (int)free_pose.size();
    b->dim = 6 * b->n_free;
    b->dim_pad = ((b->dim + 1 + NB - 1) / NB) * NB;           // room for the rhs row
    b->n_blocks = b->n_free * (b->n_free + 1) / 2;
    b->h_ur.resize((size_t)n_obs);
    std::vector<int> kf_obs((size_t)n_poses, 0);
    for (int k = 0; k < n_obs; ++k) { b->h_ur[k] = obs[k].ur; ++kf_obs[(size_t)obs[k].pose]; }   // caller order (host-side outlier thresholds)

    // ---- shape of the window (ba_band.inl): first / last FREE keyframe slot of every landmark.  When no landmark spans more than
    //      BD_MAXHBW slots the reduced system is block-banded and the problem takes the band path: landmarks ordered by their first
    //      slot, cut into groups of <= band_gmax whose observations fall into <= BD_MAXKF neighbouring keyframes.
    BandPlan plan;
    {
        static const int solver_env = [] { const char* e = getenv("LPSLAM_HIP_BA_SOLVER"); return !e ? 0 : (!strcmp(e, "dense") ? 1 : 0); }();
        static const int group_env = [] { const char* e = getenv("LPSLAM_HIP_BA_GROUP"); const int g = e ? atoi(e) : 0; return g >= 4 && g <= BD_GMAX ? g : 32; }();
        if (solver_env != 1) plan.build(obs, n_obs, n_points, slot.data(), b->n_free, b->dim, deg.data(), group_env);
        if (plan.hbw >= 0 && bd_set_attributes() != hipSuccess) { lpslam_hip_ba_destroy(b); return LPSLAM_HIP_ERR_DEVICE; }      // (the reason is in lpslam_hip_last_error)
        b->band_hbw_structure = plan.hbw;
        b->band_gmax = plan.hbw >= 0 ? group_env : 0;
        static const bool trace = getenv("LPSLAM_HIP_BA_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[lpslam_hip_ba_create] %d poses (%d free) %d points %d obs: block half-bandwidth %d, %zu groups -> %s\n", n_poses, b->n_free, n_points,
                           n_obs, plan.hbw, plan.groups.size() / BD_REC, plan.hbw >= 0 ? "band" : "dense");
    }

    // ---- landmark blocks of the landmark-major passes (k_ba_update, land_lin_body): consecutive landmarks, at most LAND_B of them and --
    //      unless a single landmark has more -- at most 256 CSR entries, so that a block's entries are one per thread
    std::vector<int> land_start;
    {
        int cnt = 0, ent = 0, first_entry = 0;
        land_start.push_back(0); land_start.push_back(0);
        for (int j = 0; j < n_points; ++j) {
            if (cnt == LAND_B || (cnt > 0 && ent + deg[j] > 256)) { first_entry += ent; land_start.push_back(j); land_start.push_back(first_entry); cnt = 0; ent = 0; }
            ++cnt; ent += deg[j];
        }
        if (n_points > 0) { land_start.push_back(n_points); land_start.push_back(first_entry + ent); }
    }
    const int land_blocks = (int)land_start.size() / 2 - 1;

    // ---- k_ba_schur's further parts (ba_build.inl): the table holds at most terms / SCH_PART items (a list of n > 256 terms has at most n / SCH_PART
    //      further parts); what the host can foresee are the diagonal blocks' (one term per observation of the keyframe) + some slack
    int extra_cap = 0, extra_first = 0;
    {
        extra_cap = (int)std::min<size_t>(std::min<size_t>((SCH_MAXP - 1) * (size_t)b->n_blocks, terms_cap / SCH_PART), (size_t)1 << 18);
        for (int i = 0; i < b->n_free; ++i) extra_first += schur_parts(kf_obs[(size_t)free_pose[(size_t)i]]) - 1;
        extra_first = std::min(std::min(extra_first + 32, extra_cap), 4095);
    }
    // ---- one block: [view | inputs as staged | zero-initialised part | the rest]
    const size_t np = (size_t)n_poses, npt = (size_t)std::max(n_points, 1), no = (size_t)std::max(n_obs, 1), n = (size_t)b->dim_pad;
    const size_t nblk = (size_t)std::max(b->n_blocks, 1), nfree = (size_t)std::max(b->n_free, 1);
    b->red_n = (int64_t)(n * n + 3 * n + 8);
    const int part_n = std::max((n_points + 63) / 64, 1);
    Carve cv;
    const size_t o_view = cv.take(sizeof(BaView));
    const size_t o_poses0 = cv.take(7 * np * 8), o_points0 = cv.take(3 * npt * 8), o_slot = cv.take(np * 4), o_free = cv.take(nfree * 4);
    const size_t o_obs_in = cv.take(no * sizeof(lpslam_hip_ba_obs));
    const size_t n_ord = plan.order.size(), n_grp = plan.groups.size() / BD_REC;
    const size_t o_land_start = cv.take(land_start.size() * 4);
    const size_t o_blk_perm = cv.take(nblk * 4);
    const size_t o_band_tab = cv.take((BD_REC * n_grp + 2 * nfree) * 4), o_band_order = cv.take(n_ord * 4), o_band_qinfo = cv.take(n_ord * 4), o_band_bstart = cv.take((n_ord + 1) * 4);
    const size_t staged_bytes = cv.off;                 // what the copy kernel moves: [0, staged_bytes)
    const size_t o_descs = cv.take(BUILD_MAX_BATCH * sizeof(BuildDesc));      // descriptors of a batched build led by this problem (device: here; host: same offset of the staging block)
    const size_t stage_alloc = cv.off;
    const size_t z_begin = cv.off;
    const size_t o_A = cv.take(np * npt * 4), o_ptcount = cv.take(npt * 4);
    const SetOff so = set_offsets(n_poses, n_points, n_obs, b->n_free, b->dim_pad);
    const size_t o_setz0 = cv.take(so.z_total * 8), o_setz1 = cv.take(so.z_total * 8);
    const size_t o_red = cv.take((size_t)b->red_n * 8), o_minv = cv.take(std::max(n * n, 64 * n) * 8) /* L^-T rows, or the band path's M blocks: 1024 doubles per 16 columns */, o_xp = cv.take(n * 8);
    const size_t o_scal = cv.take(16 * 8) /* 8 scalars + the fault words (ba_update.inl) */, o_ctl = cv.take(sizeof(BaCtl)), o_ticket = cv.take((nblk + 2 + (size_t)extra_cap) * 4) /* tickets | count | items (+ one word: a workgroup reads an item slot before it looks at the count) */, o_log = cv.take(MAX_LOG * sizeof(lpslam_hip_ba_iter_log));
    const size_t z_end = cv.off;
    const size_t o_R = cv.take(np * npt * 4), o_pscount = cv.take(np * 4), o_slotof = cv.take(no * 4);
    const size_t o_ps_start = cv.take((np + 1) * 4), o_pt_start = cv.take((npt + 1) * 4), o_pt_obs = cv.take(no * 4), o_orig = cv.take(no * 4);
    const size_t o_opose = cv.take(no * 4), o_opoint = cv.take(no * 4), o_u = cv.take(no * 8), o_v = cv.take(no * 8), o_ur = cv.take(no * 8), o_w = cv.take(no * 8);
    const size_t o_active = cv.take(no), o_actin = cv.take(no);
    const size_t o_poses_a = cv.take(7 * np * 8), o_poses_b = cv.take(7 * np * 8), o_points_a = cv.take(3 * npt * 8), o_points_b = cv.take(3 * npt * 8);
    const size_t o_setd0 = cv.take(so.d_total * 8), o_setd1 = cv.take(so.d_total * 8), o_ptrial = cv.take(np * SPLIT * 8);
    const size_t cst = csr_stride(n_obs), o_csr = cv.take(6 * cst * 8);          // u, v, ur, w (doubles) + pose, point (ints) + pose slot (int)
    const size_t o_ldiag = cv.take(n * NB * 8), o_lsub = cv.take(n * NB * 8), o_chipose = cv.take(np * 8), o_part = cv.take((size_t)std::max(part_n, 2 * std::max(land_blocks, 1)) * 8) /* k_ba_backsub: part_n; k_ba_update: scale term and chi2 per landmark block */;
    const size_t o_chiobs = cv.take(no * 8), o_depth = cv.take(no);
    const size_t o_blk_count = cv.take(nblk * 4), o_blk_start = cv.take((nblk + 1) * 4), o_blk_part = cv.take(nblk * SCH_MAXP * SCH_PV * 8);
    const size_t o_terms = cv.take(std::max<size_t>(terms_cap, 1) * sizeof(int4));
    const size_t o_band_ent = cv.take(plan.hbw >= 0 ? no * sizeof(int4) : 0), o_band_part = cv.take((n_grp + 1) * BD_PART * 8) /* + the exchange block of the twisted band factorisation */;
    {
        const int rc = lp_pool_alloc(ctx, cv.off, &b->block, &b->block_cap);
        if (rc) { lpslam_hip_ba_destroy(b); return rc; }
    }
    uint8_t* base = (uint8_t*)b->block;
    auto fail = [&](int code) { lpslam_hip_ba_destroy(b); return code; };
#define BA_HIP(x) do { if ((x) != hipSuccess) { set_error("HIP call failed: %s", #x); return fail(LPSLAM_HIP_ERR_DEVICE); } } while (0)
    // ---- the view
    BaView& v = b->h_view;
    v = BaView{};
    v.n_poses = n_poses; v.n_points = n_points; v.n_obs = n_obs; v.n_free = b->n_free; v.dim = b->dim; v.dim_pad = b->dim_pad;
    v.obs_blocks = (n_obs + 255) / 256; v.pose_blocks = (n_poses * SPLIT + 3) / 4; v.point_blocks = (n_points + 255) / 256; v.part_n = part_n;
    v.n_blocks = b->n_blocks; v.land_blocks = land_blocks;
    v.extra_pack = (extra_cap << 12) | extra_first;
    vset(v.land_start, (const int*)(base + o_land_start));
    b->d_poses[0] = (double*)(base + o_poses_a); b->d_poses[1] = (double*)(base + o_poses_b);
    b->d_points[0] = (double*)(base + o_points_a); b->d_points[1] = (double*)(base + o_points_b);
    for (int s2 = 0; s2 < 2; ++s2) { vset(v.poses_buf[s2], b->d_poses[s2]); vset(v.points_buf[s2], b->d_points[s2]); }
    vset(v.poses0, (const double*)(base + o_poses0)); vset(v.points0, (const double*)(base + o_points0));
    vset(v.pose_slot, (const int*)(base + o_slot)); vset(v.free_pose, (const int*)(base + o_free));
    vset(v.o_pose, (const int*)(base + o_opose)); vset(v.o_point, (const int*)(base + o_opoint));
    vset(v.o_u, (const double*)(base + o_u)); vset(v.o_v, (const double*)(base + o_v)); vset(v.o_ur, (const double*)(base + o_ur)); vset(v.o_w, (const double*)(base + o_w));
    b->d_o_active = base + o_active; b->d_act_in = base + o_actin; b->d_o_orig = (int*)(base + o_orig);
    vset(v.o_active, b->d_o_active);
    vset(v.pt_start, (const int*)(base + o_pt_start)); vset(v.pt_obs, (const int*)(base + o_pt_obs)); vset(v.ps_start, (const int*)(base + o_ps_start));
    vset(v.o_orig, (const int*)b->d_o_orig);
    vset(v.set_z[0], (double*)(base + o_setz0)); vset(v.set_z[1], (double*)(base + o_setz1));
    vset(v.set_d[0], (double*)(base + o_setd0)); vset(v.set_d[1], (double*)(base + o_setd1));
    vset(v.csr, (const double*)(base + o_csr));
    {   // the selected set's pointers (kernels set them with ba_lin_set before use): set 0
        double* z = (double*)(base + o_setz0); double* d = (double*)(base + o_setd0);
        vset(v.partial, z); vset(v.bp_loc, z + so.loc); vset(v.hppdiag_loc, z + so.loc + n); vset(v.chi_loc, z + so.loc + 2 * n);
        vset(v.Hll, d); vset(v.bl, d + so.bl); vset(v.Hpp, d + so.Hpp); vset(v.W, d + so.W); vset(v.hl_obs, d + so.hl);
    }
    vset(v.partial_trial, (double*)(base + o_ptrial));
    b->d_red = (double*)(base + o_red);
    vset(v.S, b->d_red); vset(v.rhs, b->d_red + n * n); vset(v.bp, b->d_red + n * n + n); vset(v.hppdiag, b->d_red + n * n + 2 * n); vset(v.chi_cur, b->d_red + n * n + 3 * n);
    vset(v.Minv, (double*)(base + o_minv)); vset(v.Ldiag, (double*)(base + o_ldiag)); vset(v.Lsub, (double*)(base + o_lsub));
    b->d_scal = (double*)(base + o_scal);
    vset(v.xp, (double*)(base + o_xp)); vset(v.chi_pose, (double*)(base + o_chipose)); vset(v.part, (double*)(base + o_part)); vset(v.scal, b->d_scal);
    vset(v.blk_start, (const int*)(base + o_blk_start)); vset(v.blk_terms, (const int4*)(base + o_terms));
    vset(v.blk_part, (double*)(base + o_blk_part)); vset(v.blk_ticket, (int*)(base + o_ticket));
    vset(v.blk_perm, (const int*)(base + o_blk_perm));
    b->d_ctl = (BaCtl*)(base + o_ctl); b->d_log = (lpslam_hip_ba_iter_log*)(base + o_log);
    vset(v.ctl, b->d_ctl); vset(v.log, b->d_log);
    v.cam = BaCam{cam->fx, cam->fy, cam->cx, cam->cy, cam->focal_x_baseline, cam->huber_mono, cam->huber_stereo};
    v.band_hbw = plan.hbw; v.band_groups = (int)n_grp; v.band_groups_cap = (int)n_grp;
    vset(v.band_tab, (const int*)(base + o_band_tab)); vset(v.band_ent, (const int*)(base + o_band_ent)); vset(v.band_part, (double*)(base + o_band_part));
    b->d_view = (BaView*)(base + o_view);
    b->d_chi_obs = (double*)(base + o_chiobs); b->d_depth = base + o_depth;
    // ---- inputs through one page-locked staging block, one copy
    b->stage = lp_pin_big_alloc(ctx, stage_alloc, &b->stage_cap);
    if (!b->stage) { set_error("page-locked staging of %zu bytes failed", stage_alloc); return fail(LPSLAM_HIP_ERR_DEVICE); }
    b->o_descs = o_descs;
    uint8_t* hs = (uint8_t*)b->stage;
    memcpy(hs + o_view, &v, sizeof(BaView));
    memcpy(hs + o_poses0, poses, 7 * np * 8);
    if (n_points) memcpy(hs + o_points0, points, 3 * (size_t)n_points * 8);
    memcpy(hs + o_slot, slot.data(), np * 4);
    if (b->n_free) memcpy(hs + o_free, free_pose.data(), (size_t)b->n_free * 4);
    if (n_obs) memcpy(hs + o_obs_in, obs, (size_t)n_obs * sizeof(lpslam_hip_ba_obs));
    memcpy(hs + o_land_start, land_start.data(), land_start.size() * 4);
    {
        // k_ba_schur's work item w (part 0 of a pose-block pair) runs on XCD (lead + extra_first + w) mod 8 -- workgroups go round the XCDs -- and every
        // XCD has an L2 of its own: with the pairs in row-major order each L2 fetched all of W (50.7 MB per launch for 5.76 MB of W,
        // profiles/r05c_pmc.json).  The free keyframes are cut into four groups and the ten group pairs dealt to the eight XCDs (six
        // off-diagonal tiles one each, the four diagonal tiles two to an XCD): an XCD's pairs then touch the W blocks of two groups, half
        // of the window.  Any assignment is valid (every pair is taken once); batched launches place problems, not pairs, on XCDs.
        std::vector<int> perm((size_t)nblk, 0);
        const int N = b->n_free, nb_ = b->n_blocks;
        static const bool tiles = [] { const char* e = getenv("LPSLAM_HIP_BA_SCHUR_TILES"); return !e || atoi(e) != 0; }();
        if (nb_ > 0) {
            std::vector<std::vector<int>> of_xcd(8);
            auto grp = [N](int i) { return std::min(3, i * 4 / std::max(N, 1)); };
            static const int tile_xcd[4][4] = {{6, 0, 1, 2}, {0, 6, 3, 4}, {1, 3, 7, 5}, {2, 4, 5, 7}};
            int blk = 0;
            // (every XCD's diagonal blocks first: theirs are the longest lists -- a term per observation -- and their epilogue waits for the pose side)
            for (int pass = 0; pass < 2; ++pass) {
                blk = 0;
                for (int i = 0; i < N; ++i) for (int k = i; k < N; ++k, ++blk) if ((i == k) == (pass == 0)) of_xcd[tiles && N >= 16 ? (size_t)tile_xcd[grp(i)][grp(k)] : (size_t)(blk & 7)].push_back(blk);
            }
            const int lead = n_poses * SPLIT;
            std::vector<size_t> at(8, 0);
            for (int w = 0; w < nb_; ++w) {
                size_t x = (size_t)((lead + extra_first + w) & 7);
                if (at[x] >= of_xcd[x].size()) { size_t best = 0, left = 0; for (size_t y = 0; y < 8; ++y) if (of_xcd[y].size() - at[y] > left) { left = of_xcd[y].size() - at[y]; best = y; } x = best; }      // its own tile is used up: from the fullest
                perm[(size_t)w] = of_xcd[x][at[x]++];
            }
        }
        memcpy(hs + o_blk_perm, perm.data(), nblk * 4);
    }
    if (plan.hbw >= 0) {
        memcpy(hs + o_band_tab, plan.groups.data(), plan.groups.size() * 4);
        memcpy(hs + o_band_tab + BD_REC * n_grp * 4, plan.glo.data(), plan.glo.size() * 4);
        memcpy(hs + o_band_tab + (BD_REC * n_grp + nfree) * 4, plan.ghi.data(), plan.ghi.size() * 4);
        memcpy(hs + o_band_order, plan.order.data(), n_ord * 4); memcpy(hs + o_band_qinfo, plan.qinfo.data(), n_ord * 4);
        memcpy(hs + o_band_bstart, plan.bstart.data(), (n_ord + 1) * 4);
    }
    // ---- what the device half will need (ba_build.inl)
    {
        BuildDesc* d = new BuildDesc();
        b->build_desc = d;
        d->n_poses = n_poses; d->n_points = n_points; d->n_obs = n_obs; d->n_free = b->n_free; d->n_blocks = b->n_blocks; d->dim = b->dim; d->dim_pad = b->dim_pad;
        d->n_ord = plan.hbw >= 0 ? (int)n_ord : 0;
        d->extra_cap = extra_cap;
        d->obs = (const lpslam_hip_ba_obs*)(base + o_obs_in);
        d->A = (int*)(base + o_A); d->R = (int*)(base + o_R); d->pt_count = (int*)(base + o_ptcount); d->ps_count = (int*)(base + o_pscount);
        d->ps_start = (int*)(base + o_ps_start); d->pt_start = (int*)(base + o_pt_start); d->slot_of = (int*)(base + o_slotof); d->pt_obs = (int*)(base + o_pt_obs);
        d->o_orig = b->d_o_orig; d->o_pose = (int*)(base + o_opose); d->o_point = (int*)(base + o_opoint);
        d->o_u = (double*)(base + o_u); d->o_v = (double*)(base + o_v); d->o_ur = (double*)(base + o_ur); d->o_w = (double*)(base + o_w);
        d->o_active = b->d_o_active; d->act_in = b->d_act_in;
        d->pose_slot = (const int*)(base + o_slot); d->free_pose = (const int*)(base + o_free);
        d->c_pose = (int*)(base + o_csr + 4 * cst * 8); d->c_point = d->c_pose + cst; d->c_slot = (int*)(base + o_csr + 5 * cst * 8);
        d->c_u = (double*)(base + o_csr); d->c_v = d->c_u + cst; d->c_ur = d->c_u + 2 * cst; d->c_w = d->c_u + 3 * cst;
        d->blk_count = (int*)(base + o_blk_count); d->blk_start = (int*)(base + o_blk_start); d->blk_ticket = (int*)(base + o_ticket); d->blk_terms = (int4*)(base + o_terms);
        d->band_order = (const int*)(base + o_band_order); d->band_qinfo = (const int*)(base + o_band_qinfo); d->band_bstart = (const int*)(base + o_band_bstart);
        d->band_ent = (int4*)(base + o_band_ent);
        d->S = b->d_red;
        d->copy_dst = (uint4*)base; d->copy_src = (const uint4*)hs; d->copy_n16 = (staged_bytes + 15) / 16;
        d->zero_dst = (uint4*)(base + z_begin); d->zero_n16 = (z_end - z_begin + 15) / 16;      // (carved in multiples of 256 bytes)
        d->view = b->d_view;
    }
#undef BA_HIP
    b->h_ctl = BaCtl{};
    b->h_ctl.ni = 2; b->h_ctl.need_lin = 1; b->h_ctl.first = 1;
    *out = b;
    return LPSLAM_HIP_OK;
}

// The device half of creation for n prepared problems: the structure phase (= g2o buildStructure of every window) as ONE launch chain,
// blockIdx.y = problem, on the first problem's stream; the other problems' streams wait for it on the device.  Asynchronous.
int lpslam_hip_ba_build_batch(lpslam_hip_ba* const* ps, int32_t n)
{
    if (!ps || n < 1) { set_error("empty batch"); return LPSLAM_HIP_ERR_INVALID; }
    for (int i = 0; i < n; ++i) {
        if (!ps[i] || !ps[i]->build_desc) { set_error("build_batch: entry %d is not a prepared problem", i); return LPSLAM_HIP_ERR_INVALID; }
        if (ps[i]->built) { set_error("build_batch: entry %d has been built already", i); return LPSLAM_HIP_ERR_INVALID; }
        if (ps[i]->ctx->cfg.device != ps[0]->ctx->cfg.device) { set_error("batch spans devices (entry %d)", i); return LPSLAM_HIP_ERR_INVALID; }
        for (int k = 0; k < i; ++k) if (ps[k] == ps[i]) { set_error("problem listed twice in a batch (entries %d, %d)", k, i); return LPSLAM_HIP_ERR_INVALID; }
    }
    LP_HIP(hipSetDevice(ps[0]->ctx->cfg.device));
    for (int c0 = 0; c0 < n; c0 += BUILD_MAX_BATCH) {
        const int m = std::min(n - c0, (int)BUILD_MAX_BATCH);
        lpslam_hip_ba* lead = ps[c0];
        hipStream_t s = lead->stream;
        BuildDesc* h_descs = (BuildDesc*)((uint8_t*)lead->stage + lead->o_descs);
        const BuildDesc* d_descs = (const BuildDesc*)((uint8_t*)lead->block + lead->o_descs);
        int mx_obs = 0, mx_poses = 0, mx_points = 0, mx_blocks = 0, mx_ord = 0, mx_pad = 0;
        size_t mx_copy = 0, mx_zero = 0;
        bool any_band = false;
        for (int i = 0; i < m; ++i) {
            const BuildDesc& d = *(const BuildDesc*)ps[c0 + i]->build_desc;
            h_descs[i] = d;
            mx_obs = std::max(mx_obs, d.n_obs); mx_poses = std::max(mx_poses, d.n_poses); mx_points = std::max(mx_points, d.n_points);
            mx_blocks = std::max(mx_blocks, d.n_blocks); mx_ord = std::max(mx_ord, d.n_ord); mx_pad = std::max(mx_pad, d.dim_pad - d.dim - 1);
            mx_copy = std::max(mx_copy, d.copy_n16); mx_zero = std::max(mx_zero, d.zero_n16);
            any_band = any_band || d.n_ord > 0;
        }
        const dim3 B256(256), BS(BS_THREADS);
        auto blocks = [](long x) { return (unsigned)std::max<long>((x + 255) / 256, 1); };
        hipLaunchKernelGGL(k_bs_descs_in, dim3(1), B256, 0, s, (uint4*)d_descs, (const uint4*)h_descs, (int)(((size_t)m * sizeof(BuildDesc) + 15) / 16));
        hipLaunchKernelGGL(k_bs_copy_in, dim3(std::min<unsigned>(64, blocks((long)mx_copy)), m), B256, 0, s, d_descs);
        hipLaunchKernelGGL(k_bs_zero, dim3(std::min<unsigned>(256, blocks((long)mx_zero)), m), B256, 0, s, d_descs);
        hipLaunchKernelGGL(k_bs_count, dim3(blocks(std::max(mx_obs, mx_pad)), m), B256, 0, s, d_descs);
        hipLaunchKernelGGL(k_bs_rowscan, dim3(mx_poses, m), BS, 0, s, d_descs);
        hipLaunchKernelGGL(k_bs_starts, dim3(2, m), BS, 0, s, d_descs);
        if (mx_obs) {
            hipLaunchKernelGGL(k_bs_scatter, dim3(blocks(mx_obs), m), B256, 0, s, d_descs);
            hipLaunchKernelGGL(k_bs_gather, dim3(blocks(mx_obs), m), B256, 0, s, d_descs);
            hipLaunchKernelGGL(k_bs_ptfill, dim3(blocks(4L * mx_points), m), B256, 0, s, d_descs);      // four lanes per landmark
            hipLaunchKernelGGL(k_bs_csrcopy, dim3(blocks(mx_obs), m), B256, 0, s, d_descs);
        }
        hipLaunchKernelGGL(k_bs_paircount, dim3((unsigned)std::max((mx_blocks + 3) / 4, 1), m), B256, 0, s, d_descs);
        hipLaunchKernelGGL(k_bs_blkscan, dim3(1, m), BS, 0, s, d_descs);
        if (mx_blocks) hipLaunchKernelGGL(k_bs_pairfill, dim3((unsigned)((mx_blocks + 3) / 4), m), B256, 0, s, d_descs);
        if (any_band && mx_ord) hipLaunchKernelGGL(k_bs_band_entries, dim3(blocks(mx_ord), m), B256, 0, s, d_descs);
        hipLaunchKernelGGL(k_bs_reset, dim3(blocks(std::max<long>(std::max<long>(7L * mx_poses, 3L * mx_points), mx_obs)), m), B256, 0, s, d_descs);
        LP_HIP(hipGetLastError());
        for (int i = 0; i < m; ++i) ps[c0 + i]->built = true;
        if (m > 1) {
            // the other problems' own streams (their solves, reads and set_state calls) are ordered behind the build on the device
            if (!lead->ev_built) LP_HIP(hipEventCreateWithFlags(&lead->ev_built, hipEventDisableTiming));
            LP_HIP(hipEventRecord(lead->ev_built, s));
            for (int i = 1; i < m; ++i) if (ps[c0 + i]->stream != s) LP_HIP(hipStreamWaitEvent(ps[c0 + i]->stream, lead->ev_built, 0));
        }
    }
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_create(lpslam_hip_ctx* ctx, const double* poses, const uint8_t* fixed, int32_t n_poses, const double* points,
                         int32_t n_points, const lpslam_hip_ba_obs* obs, int32_t n_obs, const lpslam_hip_ba_camera* cam,
                         lpslam_hip_ba** out)
{
    int rc = lpslam_hip_ba_prepare(ctx, poses, fixed, n_poses, points, n_points, obs, n_obs, cam, out);
    if (rc) return rc;
    if ((rc = lpslam_hip_ba_build_batch(out, 1))) { lpslam_hip_ba_destroy(*out); *out = nullptr; }
    return rc;
}

void lpslam_hip_ba_destroy(lpslam_hip_ba* b)
{
    if (!b) return;
    if (b->stream && !b->quiesced) (void)hipStreamSynchronize(b->stream);
    if (b->block) lp_pool_free(b->ctx, b->block, b->block_cap);
    release_stage(b);
    if (b->pin) lp_pin_free(b->ctx, b->pin);
    if (b->xfer) lp_pin_big_free(b->ctx, b->xfer, b->xfer_cap);
    if (b->xfer_in_read) (void)hipEventDestroy(b->xfer_in_read);
    if (b->ev) (void)hipEventDestroy(b->ev);
    if (b->ev_built) (void)hipEventDestroy(b->ev_built);
    if (b->stream) lp_stream_release(b->ctx, b->stream);
    delete (BuildDesc*)b->build_desc;
    delete b;
}

// caller-order activity flags -> storage order
__global__ __launch_bounds__(256) void k_ba_gather_active(const uint8_t* in, const int* o_orig, uint8_t* out, int n)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = in[o_orig[k]];
}

int lpslam_hip_ba_set_active(lpslam_hip_ba* b, const uint8_t* active)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b->n_obs) return LPSLAM_HIP_OK;
    if (active) {
        LP_HIP(hipMemcpyAsync(b->d_act_in, active, b->n_obs, hipMemcpyHostToDevice, b->stream));
        hipLaunchKernelGGL(k_ba_gather_active, dim3((b->n_obs + 255) / 256), dim3(256), 0, b->stream, b->d_act_in, b->d_o_orig, b->d_o_active, b->n_obs);
        LP_HIP(hipGetLastError());
    } else LP_HIP(hipMemsetAsync(b->d_o_active, 1, b->n_obs, b->stream));
    LP_HIP(hipStreamSynchronize(b->stream));
    release_stage(b);
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_set_points_fixed(lpslam_hip_ba* b, int32_t points_fixed)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    b->points_fixed = points_fixed ? 1 : 0;
    return LPSLAM_HIP_OK;
}

// One unit = one LM trial.  Without rejected steps `iters` units finish the call with a single look at the control block;
// every rejected trial costs one more unit, enqueued after that look.
static const hipGraphExec_t kGraphFailed = (hipGraphExec_t)(uintptr_t)1;      // cache sentinel: capture / instantiate failed for this signature
// Graph replay is OFF in a process that runs under a rocprofiler-sdk tool (rocprofv3, rocprof-compute).  A replay rings the doorbell once
// for its ~125 AQL packets; with a tool attached the HSA runtime routes every queue through its InterceptQueue, which hands the tool's
// packet interceptor (pointer into the ring, packet count) WITHOUT splitting a batch that wraps the end of the 1 MB ring, and the tool
// reads the packets as a linear array: the first replay that straddles the ring's end (after ~16 k packets on the queue) faults in the
// tool's packet loop (gpurun_out/tm2.log of round 5, resolved in DESIGN.md 13.1).  One packet per doorbell -- a direct launch -- never
// wraps.  LPSLAM_HIP_BA_GRAPH=1 forces replay (to profile it on short runs), =0 switches it off anywhere.
static bool ba_graphs_enabled()
{
    static const bool on = [] {
        if (const char* e = getenv("LPSLAM_HIP_BA_GRAPH")) return atoi(e) != 0;
        const char* tool = getenv("ROCP_TOOL_LIBRARIES"); const char* pre = getenv("LD_PRELOAD");
        return !((tool && *tool) || (pre && strstr(pre, "rocprofiler-sdk")));
    }();
    return on;
}
// capture + instantiate happen once per (stream, signature): serialised over the whole process, so that two mapping threads (two managers)
// never build graphs at the same time -- the replay itself, the hot path, takes no lock
static std::mutex g_ba_capture_mutex;
static int enqueue_batch(const BaLaunch& L, int units, bool first_batch)
{
    for (int u = 0; u < units; ++u) {
        int rc;
        if ((rc = enqueue_linearize(L, 1, first_batch && u == 0))) return rc;
        if ((rc = enqueue_reduce(L, 1))) return rc;
        if ((rc = enqueue_solve(L, 1))) return rc;
    }
    return LPSLAM_HIP_OK;
}

// optimize() in two halves: begin enqueues the whole first batch on the problem's stream and returns (the mapping side of the
// reference runs beside tracking: the caller can enqueue front-end work of the next frames meanwhile), end waits, handles
// rejected trials and fetches the log.
int lpslam_hip_ba_optimize_begin(lpslam_hip_ba* b, int32_t robust, int32_t iters)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (iters < 0 || iters > MAX_LOG) { set_error("iterations must be in [0,%d]", MAX_LOG); return LPSLAM_HIP_ERR_INVALID; }
    if (b->pending_iters >= 0) { set_error("optimize_begin: the previous optimize_begin has not been ended"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(b->ctx->cfg.device));
    b->robust = robust;
    // The first batch of a call -- the arming of the control block and `iters` units, the first with its explicit linearisation --
    // has a fixed launch sequence for a given (launch extents, robust, iters, points_fixed).  The second time a stream of this context
    // sees a signature the sequence is captured into a hipGraph whose kernels read their view from the STREAM's slot, and from then
    // on every problem with that signature on that stream -- the same window solved again, or the next keyframe's new window --
    // copies its view into the slot and replays the graph with one hipGraphLaunch.  Extents are rounded up (surplus workgroups
    // exit on the view's own extents, as in a batch), so windows of slightly different size share a graph.
    const int units = iters;
    bool launched = false;
    int rc;
    if (units > 0) {
        BaLaunch L = single_launch(b);
        auto up = [](int x, int m) { return (x + m - 1) / m * m; };
        L.obs_blocks = up(L.obs_blocks, 8); L.pose_blocks = up(L.pose_blocks, 4); L.point_blocks = up(L.point_blocks, 4); L.part_n = up(L.part_n, 8); L.land_blocks = up(L.land_blocks, 8);
        L.band_groups = up(L.band_groups, 8); L.band_blocks = up(L.band_blocks, 8); L.schur_items = up(L.schur_items, 128);
        const std::array<int, 24> sig = {L.schur_items, units, robust ? 1 : 0, b->points_fixed ? 1 : 0, L.obs_blocks, L.pose_blocks, L.point_blocks, L.part_n, L.n_free,
                                         L.n_blocks, L.dim, L.nb, L.any_small ? 1 : 0, L.any_large ? 1 : 0, L.land_blocks, L.spread ? 1 : 0,
                                         L.any_band ? 1 : 0, L.any_dense ? 1 : 0, L.band_groups, L.band_blocks, L.band_gmax, L.any_one_pass ? 1 : 0, L.any_two_launch ? 1 : 0, L.n_poses};
        lpslam_hip_ctx* c = b->ctx;
        hipGraphExec_t exec = nullptr;
        void* slot = nullptr;
        bool capture = false;
        if (ba_graphs_enabled() && b->stream != c->role_solve) {      // (several sessions' windows run on the solves' role stream: no capture on a stream other threads launch on)
            std::lock_guard<std::mutex> lock(c->pool_mutex);
            auto key = std::make_pair(b->stream, sig);
            auto it = c->ba_graphs.find(key);
            if (it == c->ba_graphs.end()) { if (c->ba_graphs.size() < 256) c->ba_graphs.emplace(key, nullptr); }   // seen once: run directly (also sets function attributes); the cache is bounded, never evicted (an entry may be in flight on its stream)
            else { exec = it->second; capture = exec == nullptr; if (exec == kGraphFailed) exec = nullptr; }      // a signature whose capture failed once runs direct from then on
            auto sl = c->ba_view_slot.find(b->stream);
            if (sl != c->ba_view_slot.end()) slot = sl->second;
        }
        if ((exec || capture) && !slot) {
            if (hipMalloc(&slot, sizeof(BaView)) == hipSuccess) { std::lock_guard<std::mutex> lock(c->pool_mutex); c->ba_view_slot[b->stream] = slot; }
            else { slot = nullptr; (void)hipGetLastError(); }
        }
        if (capture && slot) {
            std::lock_guard<std::mutex> capture_lock(g_ba_capture_mutex);
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                L.d_views = (const BaView*)slot;
                hipLaunchKernelGGL(k_ba_arm, dim3(1, 1), dim3(64), 0, b->stream, (const BaView*)slot, iters);
                int r2 = enqueue_batch(L, units, true);
                const hipError_t e2 = hipStreamEndCapture(b->stream, &graph);
                if (r2 == LPSLAM_HIP_OK && e2 == hipSuccess && graph) {
                    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                        std::lock_guard<std::mutex> lock(c->pool_mutex);
                        c->ba_graphs[std::make_pair(b->stream, sig)] = exec;
                        c->ba_graphs_built.fetch_add(1);
                    } else exec = nullptr;
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (!exec) { std::lock_guard<std::mutex> lock(c->pool_mutex); c->ba_graphs[std::make_pair(b->stream, sig)] = kGraphFailed; }
            (void)hipGetLastError();
        }
        if (exec && slot) {
            LP_HIP(hipMemcpyAsync(slot, b->d_view, sizeof(BaView), hipMemcpyDeviceToDevice, b->stream));
            LP_HIP(hipGraphLaunch(exec, b->stream));
            launched = true; c->ba_graph_replays.fetch_add(1);
        }
    }
    if (!launched) {
        if ((rc = begin_optimize(b, robust, iters))) return rc;
        if (units > 0 && (rc = enqueue_batch(single_launch(b), units, true))) return rc;
    }
    b->pending_iters = iters;
    return LPSLAM_HIP_OK;
}

int64_t lpslam_hip_ba_graph_replays(lpslam_hip_ctx* c) { return c ? (int64_t)c->ba_graph_replays.load() : 0; }
int lpslam_hip_ba_counters(lpslam_hip_ctx* c, int64_t* out, int32_t n)
{
    if (!c || !out || n < 0) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    int64_t v[LPSLAM_HIP_BA_COUNTERS] = {0};
    {
        std::lock_guard<std::mutex> lock(c->pool_mutex);
        v[LPSLAM_HIP_BA_COUNTER_SIGNATURES] = (int64_t)c->ba_graphs.size();
    }
    v[LPSLAM_HIP_BA_COUNTER_GRAPHS] = (int64_t)c->ba_graphs_built.load();
    v[LPSLAM_HIP_BA_COUNTER_REPLAYS] = (int64_t)c->ba_graph_replays.load();
    v[LPSLAM_HIP_BA_COUNTER_TIMEOUTS_BAND] = (int64_t)c->ba_timeouts_band.load();
    v[LPSLAM_HIP_BA_COUNTER_TIMEOUTS_UPDATE] = (int64_t)c->ba_timeouts_update.load();
    for (int i = 0; i < n && i < LPSLAM_HIP_BA_COUNTERS; ++i) out[i] = v[i];
    return LPSLAM_HIP_OK;
}
int64_t lpslam_hip_ba_wg_factorisations(lpslam_hip_ctx* c) { return c ? (int64_t)c->ba_wg_launches.load() : 0; }
int32_t lpslam_hip_pose_optimize_passes(lpslam_hip_ctx* c) { return c ? c->po_passes : 0; }

int lpslam_hip_ba_optimize_end(lpslam_hip_ba* b, lpslam_hip_ba_iter_log* log, int32_t* done_out)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (b->pending_iters < 0) { set_error("optimize_end without optimize_begin"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(b->ctx->cfg.device));
    const int iters = b->pending_iters;
    b->pending_iters = -1;
    int rc;
    const int want_log = (log && b->pin) ? iters : 0;
    if ((rc = read_ctl(b, want_log))) return rc;
    if (iters > 0) {
        int guard = 0;
        while (!b->h_ctl.stopped && b->h_ctl.outer_done < iters && guard++ < 16 * MAX_LOG) {
            if ((rc = enqueue_batch(single_launch(b), iters - b->h_ctl.outer_done, false))) return rc;
            if ((rc = read_ctl(b, want_log))) return rc;
        }
    }
    if ((rc = report_faults(b))) return rc;
    const int done = b->h_ctl.outer_done;
    if (log && done) {
        if (want_log) memcpy(log, b->pin->log, (size_t)std::min(done, MAX_LOG) * sizeof(lpslam_hip_ba_iter_log));     // came with the control block
        else LP_HIP(hipMemcpy(log, b->d_log, std::min(done, MAX_LOG) * sizeof(lpslam_hip_ba_iter_log), hipMemcpyDeviceToHost));
    }
    if (done_out) *done_out = done;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_optimize(lpslam_hip_ba* b, int32_t robust, int32_t iters, lpslam_hip_ba_iter_log* log, int32_t* done_out)
{
    const int rc = lpslam_hip_ba_optimize_begin(b, robust, iters);
    return rc ? rc : lpslam_hip_ba_optimize_end(b, log, done_out);
}

// optimize() with a HIP event after every launch: where the time of the chain goes, kernel by kernel, measured in place on the
// problem's stream (bench.py's roofline of the dominant kernel).  No graph replay; the events serialise nothing the chain does
// not serialise itself (every launch depends on its predecessor).
int lpslam_hip_ba_optimize_profiled(lpslam_hip_ba* b, int32_t robust, int32_t iters, lpslam_hip_ba_kernel_times* out)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b || !out) { set_error("null argument"); return LPSLAM_HIP_ERR_INVALID; }
    if (iters < 0 || iters > MAX_LOG) { set_error("iterations must be in [0,%d]", MAX_LOG); return LPSLAM_HIP_ERR_INVALID; }
    if (b->pending_iters >= 0) { set_error("optimize_begin pending"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(b->ctx->cfg.device));
    memset(out, 0, sizeof(*out));
    std::vector<std::pair<hipEvent_t, int>> marks;
    BaLaunch L = single_launch(b);
    L.robust = robust; b->robust = robust;
    L.marks = &marks;
    int rc = begin_optimize(b, robust, iters);
    L.mark(-1);                                            // start of the chain
    if (!rc && iters > 0) rc = enqueue_batch(L, iters, true);
    if (!rc) rc = read_ctl(b);
    for (int guard = 0; !rc && iters > 0 && !b->h_ctl.stopped && b->h_ctl.outer_done < iters && guard < 16 * MAX_LOG; ++guard) {
        L.mark(-1);
        rc = enqueue_batch(L, iters - b->h_ctl.outer_done, false);
        if (!rc) rc = read_ctl(b);
    }
    for (size_t i = 1; i < marks.size() && !rc; ++i) {
        const int k = marks[i].second;
        if (k < 0 || k >= LPSLAM_HIP_BA_KERNELS) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, marks[i - 1].first, marks[i].first) == hipSuccess) { out->ms[k] += ms; out->launches[k] += 1; }
    }
    for (auto& m : marks) (void)hipEventDestroy(m.first);
    out->iterations = b->h_ctl.outer_done;
    // launches of the factorisation per mark: the panel-pair chain is several launches behind one mark
    out->launches_per_mark[LPSLAM_HIP_BA_K_CHOL] = b->h_view.band_hbw >= 0 ? 1 : (b->dim_pad / NB + 1) / 2;
    for (int k = 0; k < LPSLAM_HIP_BA_KERNELS; ++k) if (k != LPSLAM_HIP_BA_K_CHOL) out->launches_per_mark[k] = 1;
    out->dim = b->dim;
    return rc;
}

// ---- batched solve: B independent problems, ONE launch chain (blockIdx.y = problem) ------------------------------------------
// What a host that serves several SLAM sessions (or several windows of one map) on one GPU calls: every kernel of the chain is
// launched once for the whole batch with the launch extents of its largest problem, each problem follows its own control block
// (a problem that has finished, or terminated, idles through the remaining launches), and the control blocks and logs of all
// problems come back in one copy.  The single-problem chain is latency bound; a batch fills the chip.
static int batch_sync_streams(lpslam_hip_ba* const* ps, int n, hipStream_t s)
{
    for (int i = 0; i < n; ++i) {
        if (ps[i]->stream == s) continue;
        if (!ps[i]->ev) LP_HIP(hipEventCreateWithFlags(&ps[i]->ev, hipEventDisableTiming));
        LP_HIP(hipEventRecord(ps[i]->ev, ps[i]->stream));
        LP_HIP(hipStreamWaitEvent(s, ps[i]->ev, 0));
    }
    return LPSLAM_HIP_OK;
}
static int batch_check(lpslam_hip_ba* const* ps, int n)
{
    if (!ps || n < 1) { set_error("empty batch"); return LPSLAM_HIP_ERR_INVALID; }
    for (int i = 0; i < n; ++i) {
        if (!ps[i]) { set_error("null problem in batch (entry %d)", i); return LPSLAM_HIP_ERR_INVALID; }
        if (ps[i]->ctx->cfg.device != ps[0]->ctx->cfg.device) { set_error("batch spans devices (entry %d)", i); return LPSLAM_HIP_ERR_INVALID; }
        if (ps[i]->pending_iters >= 0) { set_error("batch entry %d has an optimize_begin pending", i); return LPSLAM_HIP_ERR_INVALID; }
        if (!ps[i]->built) { set_error("batch entry %d has been prepared but not built (lpslam_hip_ba_build_batch)", i); return LPSLAM_HIP_ERR_INVALID; }
        for (int k = 0; k < i; ++k) if (ps[k] == ps[i]) { set_error("problem listed twice in a batch (entries %d, %d)", k, i); return LPSLAM_HIP_ERR_INVALID; }
    }
    return LPSLAM_HIP_OK;
}
// device array of the problems' views (a block of the first problem's context) + its launch extents
struct BatchViews {
    lpslam_hip_ctx* ctx = nullptr; void* blk = nullptr; size_t cap = 0; void* hst = nullptr; size_t hcap = 0;
    BaLaunch L;
    ~BatchViews() { if (blk) lp_pool_free(ctx, blk, cap); if (hst) lp_pin_big_free(ctx, hst, hcap); }
};
static int batch_views(lpslam_hip_ba* const* ps, int n, size_t extra_bytes, BatchViews* bv)
{
    bv->ctx = ps[0]->ctx;
    const size_t view_bytes = (size_t)n * sizeof(BaView), total = ((view_bytes + 255) & ~(size_t)255) + extra_bytes;
    int rc = lp_pool_alloc(bv->ctx, total, &bv->blk, &bv->cap); if (rc) return rc;
    bv->hst = lp_pin_big_alloc(bv->ctx, total, &bv->hcap);
    if (!bv->hst) { set_error("page-locked staging of %zu bytes failed", total); return LPSLAM_HIP_ERR_DEVICE; }
    BaLaunch& L = bv->L;
    L.d_views = (const BaView*)bv->blk; L.s = ps[0]->stream; L.ctx = bv->ctx;
    for (int i = 0; i < n; ++i) { memcpy((uint8_t*)bv->hst + (size_t)i * sizeof(BaView), &ps[i]->h_view, sizeof(BaView)); L.add(ps[i]); }
    LP_HIP(hipMemcpyAsync(bv->blk, bv->hst, view_bytes, hipMemcpyHostToDevice, L.s));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_reset_batch(lpslam_hip_ba* const* ps, int32_t n)
{
    int rc = batch_check(ps, n); if (rc) return rc;
    LP_HIP(hipSetDevice(ps[0]->ctx->cfg.device));
    BatchViews bv;
    if ((rc = batch_sync_streams(ps, n, ps[0]->stream))) return rc;
    if ((rc = batch_views(ps, n, 0, &bv))) return rc;
    long n_max = 1;
    for (int i = 0; i < n; ++i) {
        n_max = std::max<long>(n_max, std::max<long>(std::max<long>(7L * ps[i]->n_poses, 3L * ps[i]->n_points), ps[i]->n_obs));
        ps[i]->h_ctl = BaCtl{}; ps[i]->h_ctl.ni = 2; ps[i]->h_ctl.need_lin = 1; ps[i]->h_ctl.first = 1;
    }
    hipLaunchKernelGGL(k_ba_reset, dim3((unsigned)((n_max + 255) / 256), n), dim3(256), 0, bv.L.s, bv.L.d_views);
    LP_HIP(hipGetLastError());
    LP_HIP(hipStreamSynchronize(bv.L.s));          // the view array is released on return
    for (int i = 0; i < n; ++i) release_stage(ps[i]);
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_optimize_batch(lpslam_hip_ba* const* ps, int32_t n, int32_t robust, int32_t iters, lpslam_hip_ba_iter_log* logs,
                                 int32_t log_stride, int32_t* done)
{
    int rc = batch_check(ps, n); if (rc) return rc;
    if (iters < 0 || iters > MAX_LOG) { set_error("iterations must be in [0,%d]", MAX_LOG); return LPSLAM_HIP_ERR_INVALID; }
    if (logs && log_stride < iters) { set_error("log_stride %d is smaller than the iteration count %d", log_stride, iters); return LPSLAM_HIP_ERR_INVALID; }
    for (int i = 1; i < n; ++i)
        if (ps[i]->points_fixed != ps[0]->points_fixed) { set_error("batch mixes motion-only and full problems (entry %d)", i); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(ps[0]->ctx->cfg.device));
    BatchViews bv;
    const size_t collect_bytes = (size_t)n * COLLECT_STRIDE;
    if ((rc = batch_sync_streams(ps, n, ps[0]->stream))) return rc;
    if ((rc = batch_views(ps, n, collect_bytes, &bv))) return rc;
    BaLaunch& L = bv.L;
    L.robust = robust; L.points_fixed = ps[0]->points_fixed;
    for (int i = 0; i < n; ++i) ps[i]->robust = robust;
    const size_t coll_off = ((size_t)n * sizeof(BaView) + 255) & ~(size_t)255;
    uint8_t* d_coll = (uint8_t*)bv.blk + coll_off;
    uint8_t* h_coll = (uint8_t*)bv.hst + coll_off;
    hipLaunchKernelGGL(k_ba_arm, dim3(1, n), dim3(64), 0, L.s, L.d_views, iters);
    auto collect = [&]() -> int {
        hipLaunchKernelGGL(k_ba_collect, dim3(1, n), dim3(64), 0, L.s, L.d_views, d_coll, iters);
        LP_HIP(hipGetLastError());
        LP_HIP(hipMemcpyAsync(h_coll, d_coll, collect_bytes, hipMemcpyDeviceToHost, L.s));
        LP_HIP(hipStreamSynchronize(L.s));
        for (int i = 0; i < n; ++i) memcpy(&ps[i]->h_ctl, h_coll + (size_t)i * COLLECT_STRIDE, sizeof(BaCtl));
        return LPSLAM_HIP_OK;
    };
    if (iters > 0 && (rc = enqueue_batch(L, iters, true))) return rc;
    if ((rc = collect())) return rc;
    for (int guard = 0; iters > 0 && guard < 16 * MAX_LOG; ++guard) {
        int remaining = 0;
        for (int i = 0; i < n; ++i) if (!ps[i]->h_ctl.stopped) remaining = std::max(remaining, iters - ps[i]->h_ctl.outer_done);
        if (remaining <= 0) break;
        if ((rc = enqueue_batch(L, remaining, false))) return rc;
        if ((rc = collect())) return rc;
    }
    for (int i = 0; i < n; ++i) if ((rc = report_faults(ps[i]))) { for (int k = 0; k < n; ++k) release_stage(ps[k]); return rc; }
    for (int i = 0; i < n; ++i) {
        release_stage(ps[i]);
        const int d = ps[i]->h_ctl.outer_done;
        if (done) done[i] = d;
        if (logs && d) memcpy(logs + (size_t)i * log_stride, h_coll + (size_t)i * COLLECT_STRIDE + sizeof(BaCtl), (size_t)std::min(d, MAX_LOG) * sizeof(lpslam_hip_ba_iter_log));
    }
    return LPSLAM_HIP_OK;
}

// ---- partitioned (multi-GPU) solve: one LM trial in three phases with the caller's all-reduces in between ------------------
//   lpslam_hip_ba_step_begin : (linearise if needed) + partial Schur complement -> reduced buffer [S | rhs | b_p | diag H_pp |
//                              chi2]: SUM all-reduce; scalar buffer entry [4] = max diag H_ll: MAX all-reduce (first trial)
//   lpslam_hip_ba_step_solve : lambda control start, factor, solve, update into the trial state; scalar buffer [1] = trial
//                              chi2 and [2] = landmark scale term: SUM all-reduce ([3], the pose term, is identical on all ranks)
//   lpslam_hip_ba_step_end   : accept / reject; reports the control state
// Every phase ends with a stream synchronise so the caller's collective may touch the buffers right away.
int lpslam_hip_ba_timeouts(lpslam_hip_ba* b, int32_t* band, int32_t* update)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (band) *band = b->faults_band;
    if (update) *update = b->faults_update;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_get_solver(lpslam_hip_ba* b, int32_t* solver, int32_t* block_half_bandwidth)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (solver) *solver = b->h_view.band_hbw >= 0 ? LPSLAM_HIP_BA_SOLVER_BAND : LPSLAM_HIP_BA_SOLVER_DENSE;
    if (block_half_bandwidth) *block_half_bandwidth = b->band_hbw_structure;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_set_solver(lpslam_hip_ba* b, int32_t solver)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (b->pending_iters >= 0) { set_error("set_solver between optimize_begin and optimize_end"); return LPSLAM_HIP_ERR_INVALID; }
    int want = solver == LPSLAM_HIP_BA_SOLVER_DENSE ? -1 : b->band_hbw_structure;
    if (solver == LPSLAM_HIP_BA_SOLVER_BAND && b->band_hbw_structure < 0) { set_error("the window is not block-banded (a landmark spans more than %d free keyframes, or the system exceeds %d unknowns)", BD_MAXHBW + 1, 16 * BC_MAXS); return LPSLAM_HIP_ERR_INVALID; }
    if (want == b->h_view.band_hbw) return LPSLAM_HIP_OK;
    LP_HIP(hipSetDevice(b->ctx->cfg.device));
    LP_HIP(hipStreamSynchronize(b->stream));
    release_stage(b);
    b->h_view.band_hbw = want;
    LP_HIP(hipMemcpy(&b->d_view->band_hbw, &b->h_view.band_hbw, sizeof(int), hipMemcpyHostToDevice));
    // the other solver's leftovers in S (factor entries outside the band / inside it) must not be taken for matrix entries
    const size_t n = (size_t)b->dim_pad;
    LP_HIP(hipMemsetAsync(b->d_red, 0, n * n * sizeof(double), b->stream));
    if (b->dim_pad > b->dim + 1) hipLaunchKernelGGL(k_bs_identity, dim3((b->dim_pad - b->dim - 1 + 255) / 256), dim3(256), 0, b->stream, b->d_red, b->dim, b->dim_pad);
    LP_HIP(hipStreamSynchronize(b->stream));
    return LPSLAM_HIP_OK;
}

// the partitioned (all-reduced) solve works on the dense reduced buffer
static int ensure_dense(lpslam_hip_ba* b) { return b->h_view.band_hbw >= 0 ? lpslam_hip_ba_set_solver(b, LPSLAM_HIP_BA_SOLVER_DENSE) : LPSLAM_HIP_OK; }

int lpslam_hip_ba_step_begin(lpslam_hip_ba* b, int32_t robust, int32_t first)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(b->ctx->cfg.device));
    int rc;
    if ((rc = ensure_dense(b))) return rc;
    if (first) { if ((rc = begin_optimize(b, robust, MAX_LOG))) return rc; }
    b->robust = robust;
    // lambda is needed by the Schur complement but lambda_0 depends on all-reduced diagonals: on the very first trial the
    // caller runs begin twice (first = 1: linearisation only; first = 0 after the reduction of the diagonals)
    if ((rc = enqueue_linearize(single_launch(b), 0))) return rc;
    if (!first) { if ((rc = enqueue_reduce(single_launch(b), 0))) return rc; }
    LP_HIP(hipStreamSynchronize(b->stream));
    release_stage(b);
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_step_lambda0(lpslam_hip_ba* b)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    hipLaunchKernelGGL(k_lm_begin, dim3(1, 1), dim3(64), 0, b->stream, b->d_view);
    LP_HIP(hipGetLastError());
    LP_HIP(hipStreamSynchronize(b->stream));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_step_solve(lpslam_hip_ba* b)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    int rc = enqueue_solve(single_launch(b), 0); if (rc) return rc;
    LP_HIP(hipStreamSynchronize(b->stream));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_step_end(lpslam_hip_ba* b, int32_t* accepted, int32_t* iteration_finished)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    const int before = b->h_ctl.outer_done;
    hipLaunchKernelGGL(k_lm_decide, dim3(1, 1), dim3(64), 0, b->stream, b->d_view);
    LP_HIP(hipGetLastError());
    int rc = read_ctl(b); if (rc) return rc;
    if (accepted) *accepted = b->h_ctl.last_accepted;
    if (iteration_finished) *iteration_finished = (b->h_ctl.outer_done != before || b->h_ctl.stopped) ? 1 : 0;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_status(lpslam_hip_ba* b, int32_t* outer_done, int32_t* stopped, double* lambda, double* chi2)
{
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (outer_done) *outer_done = b->h_ctl.outer_done;
    if (stopped) *stopped = b->h_ctl.stopped;
    if (lambda) *lambda = b->h_ctl.lambda;
    if (chi2) *chi2 = b->h_ctl.current_chi;
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

int lpslam_hip_ba_reset(lpslam_hip_ba* b)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    b->h_ctl = BaCtl{};
    b->h_ctl.ni = 2; b->h_ctl.need_lin = 1; b->h_ctl.first = 1;
    // one launch instead of two device copies and a fill (three runtime operations of ~4 us each on the solve's stream)
    const long n_max = std::max<long>(std::max<long>(7L * b->n_poses, 3L * b->n_points), b->n_obs);
    hipLaunchKernelGGL(k_ba_reset, dim3((unsigned)((n_max + 255) / 256), 1), dim3(256), 0, b->stream, b->d_view);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

// New creation-time values for an existing structure (the observation graph stays): what lets a mapping thread build the
// structure of the next window (lpslam_hip_ba_create is asynchronous) while the previous window is still being solved, and hand
// over the poses / landmarks that solve produced when it is done.
int lpslam_hip_ba_set_state(lpslam_hip_ba* b, const double* poses, const double* points)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (b->pending_iters >= 0) { set_error("lpslam_hip_ba_set_state while a solve is in flight"); return LPSLAM_HIP_ERR_INVALID; }
    if (!poses && !(points && b->n_points)) return lpslam_hip_ba_reset(b);
    if (uint8_t* x = ensure_xfer(b)) {
        if (b->xfer_in_pending) { LP_HIP(hipEventSynchronize(b->xfer_in_read)); b->xfer_in_pending = false; }      // the previous set_state's kernel has read its values
        double* in_poses = (double*)x; double* in_points = in_poses + 7 * (size_t)b->n_poses;
        if (poses) memcpy(in_poses, poses, 7 * (size_t)b->n_poses * sizeof(double));
        if (points && b->n_points) memcpy(in_points, points, 3 * (size_t)b->n_points * sizeof(double));
        b->h_ctl = BaCtl{};
        b->h_ctl.ni = 2; b->h_ctl.need_lin = 1; b->h_ctl.first = 1;
        const long n_max = std::max<long>(std::max<long>(7L * b->n_poses, 3L * b->n_points), b->n_obs);
        hipLaunchKernelGGL(k_ba_reset_from_host, dim3((unsigned)((n_max + 255) / 256), 1), dim3(256), 0, b->stream, b->d_view,
                           poses ? in_poses : nullptr, (points && b->n_points) ? in_points : nullptr);
        LP_HIP(hipGetLastError());
        LP_HIP(hipEventRecord(b->xfer_in_read, b->stream));
        b->xfer_in_pending = true;
        return LPSLAM_HIP_OK;
    }
    if (poses) LP_HIP(hipMemcpyAsync((void*)b->h_view.poses0, poses, 7 * (size_t)b->n_poses * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (points && b->n_points) LP_HIP(hipMemcpyAsync((void*)b->h_view.points0, points, 3 * (size_t)b->n_points * sizeof(double), hipMemcpyHostToDevice, b->stream));
    return lpslam_hip_ba_reset(b);
}

int lpslam_hip_ba_get(lpslam_hip_ba* b, double* poses, double* points)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (uint8_t* x = ensure_xfer(b)) {
        auto even = [](size_t n) { return (n + 1) & ~(size_t)1; };      // 16-byte aligned halves
        double* out_poses = (double*)x + even(7 * (size_t)b->n_poses + 3 * (size_t)std::max(b->n_points, 1));
        double* out_points = out_poses + even(7 * (size_t)b->n_poses);
        const bool want_points = points && b->n_points;
        const long n_max = std::max<long>(poses ? 7L * b->n_poses : 0, want_points ? 3L * b->n_points : 0);
        if (n_max > 0) {
            hipLaunchKernelGGL(k_ba_state_to_host, dim3((unsigned)((n_max + 511) / 512), 1), dim3(256), 0, b->stream, b->d_view, poses ? out_poses : nullptr, want_points ? out_points : nullptr);
            LP_HIP(hipGetLastError());
        }
        LP_HIP(hipStreamSynchronize(b->stream));
        release_stage(b);
        if (poses) memcpy(poses, out_poses, 7 * (size_t)b->n_poses * sizeof(double));
        if (want_points) memcpy(points, out_points, 3 * (size_t)b->n_points * sizeof(double));
        return LPSLAM_HIP_OK;
    }
    const int cur = b->h_ctl.cur;
    if (poses) LP_HIP(hipMemcpyAsync(poses, b->d_poses[cur], 7 * (size_t)b->n_poses * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    if (points && b->n_points) LP_HIP(hipMemcpyAsync(points, b->d_points[cur], 3 * (size_t)b->n_points * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    LP_HIP(hipStreamSynchronize(b->stream));
    release_stage(b);
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_set_state_batch(lpslam_hip_ba* const* ps, int32_t n, const double* const* poses, const double* const* points)
{
    if (n < 0 || (n > 0 && !ps)) { set_error("bad batch"); return LPSLAM_HIP_ERR_INVALID; }
    for (int i = 0; i < n; ++i) {
        const int rc = lpslam_hip_ba_set_state(ps[i], poses ? poses[i] : nullptr, points ? points[i] : nullptr);      // (asynchronous: a copy into the exchange block + one launch)
        if (rc) return rc;
    }
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_get_batch(lpslam_hip_ba* const* ps, int32_t n, double* const* poses, double* const* points)
{
    if (n < 0 || (n > 0 && !ps)) { set_error("bad batch"); return LPSLAM_HIP_ERR_INVALID; }
    auto even = [](size_t m) { return (m + 1) & ~(size_t)1; };
    std::vector<uint8_t> through_block((size_t)n, 0);
    // every problem's state kernel first ...
    for (int i = 0; i < n; ++i) {
        lpslam_hip_ba* b = ps[i];
        if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
        double* po = poses ? poses[i] : nullptr; double* pt = points ? points[i] : nullptr;
        uint8_t* x = ensure_xfer(b);
        if (!x) continue;                               // no exchange block: the single call below
        double* out_poses = (double*)x + even(7 * (size_t)b->n_poses + 3 * (size_t)std::max(b->n_points, 1));
        double* out_points = out_poses + even(7 * (size_t)b->n_poses);
        const bool want_points = pt && b->n_points;
        const long n_max = std::max<long>(po ? 7L * b->n_poses : 0, want_points ? 3L * b->n_points : 0);
        if (n_max > 0) {
            hipLaunchKernelGGL(k_ba_state_to_host, dim3((unsigned)((n_max + 511) / 512), 1), dim3(256), 0, b->stream, b->d_view, po ? out_poses : nullptr, want_points ? out_points : nullptr);
            LP_HIP(hipGetLastError());
        }
        through_block[(size_t)i] = 1;
    }
    // ... then the waits and the copies out
    for (int i = 0; i < n; ++i) {
        lpslam_hip_ba* b = ps[i];
        double* po = poses ? poses[i] : nullptr; double* pt = points ? points[i] : nullptr;
        if (!through_block[(size_t)i]) { const int rc = lpslam_hip_ba_get(b, po, pt); if (rc) return rc; continue; }
        LP_HIP(hipStreamSynchronize(b->stream));
        release_stage(b);
        double* out_poses = (double*)b->xfer + even(7 * (size_t)b->n_poses + 3 * (size_t)std::max(b->n_points, 1));
        double* out_points = out_poses + even(7 * (size_t)b->n_poses);
        if (po) memcpy(po, out_poses, 7 * (size_t)b->n_poses * sizeof(double));
        if (pt && b->n_points) memcpy(pt, out_points, 3 * (size_t)b->n_points * sizeof(double));
    }
    return LPSLAM_HIP_OK;
}

int lpslam_hip_ba_chi2(lpslam_hip_ba* b, double* chi2, uint8_t* depth_positive)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b) { set_error("null problem"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b->n_obs) return LPSLAM_HIP_OK;
    hipLaunchKernelGGL(k_ba_obs_chi2, dim3((b->n_obs + 255) / 256, 1), dim3(256), 0, b->stream, b->d_view, b->d_chi_obs, b->d_depth);
    LP_HIP(hipGetLastError());
    if (chi2) LP_HIP(hipMemcpyAsync(chi2, b->d_chi_obs, (size_t)b->n_obs * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    if (depth_positive) LP_HIP(hipMemcpyAsync(depth_positive, b->d_depth, (size_t)b->n_obs, hipMemcpyDeviceToHost, b->stream));
    LP_HIP(hipStreamSynchronize(b->stream));
    release_stage(b);
    return LPSLAM_HIP_OK;
}

}  // extern "C"

int lp_launch_pose_batch(hipStream_t s, const LpPoseReq* reqs, int n)
{
    for (int i0 = 0; i0 < n; i0 += PO_MAX_BATCH) {
        const int m = std::min(n - i0, (int)PO_MAX_BATCH);
        PoBatch b{};
        int n_max = 0;
        for (int i = 0; i < m; ++i) { b.blk[i] = reqs[i0 + i].blk; b.n[i] = reqs[i0 + i].n; b.seq[i] = reqs[i0 + i].seq; n_max = std::max(n_max, reqs[i0 + i].n); }
        const int threads = n_max <= 64 ? 64 : (n_max <= 128 ? 128 : 256), wmax = threads / 64;
        const size_t lds = (size_t)(PO_NV * 66 * wmax + 32) * sizeof(double) + (size_t)64 * wmax * sizeof(PoObs) + 64;
        {
            static std::atomic<bool> attr_set[64];
            int dev = 0; (void)hipGetDevice(&dev);
            if (dev >= 0 && dev < 64 && !attr_set[dev].load()) { (void)hipFuncSetAttribute((const void*)k_pose_optimize_req, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr_set[dev].store(true); }
        }
        hipLaunchKernelGGL(k_pose_optimize_req, dim3((unsigned)m), dim3((unsigned)threads), lds, s, b);
        LP_HIP(hipGetLastError());
    }
    return LPSLAM_HIP_OK;
}

extern "C" {

int lpslam_hip_pose_optimize(lpslam_hip_ctx* ctx, double* pose7, const double* points, int32_t n_points, const lpslam_hip_ba_obs* obs, int32_t n_obs,
                             const lpslam_hip_ba_camera* cam, uint8_t* outlier, int32_t* n_inliers)
{
    if (!ctx || !pose7 || !cam || n_obs < 0 || n_points < 0 || (n_obs > 0 && (!points || !obs))) { set_error("invalid pose-optimiser arguments"); return LPSLAM_HIP_ERR_INVALID; }
    for (int k = 0; k < n_obs; ++k) if (obs[k].point < 0 || obs[k].point >= n_points) { set_error("observation %d references point %d out of range", k, obs[k].point); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->stream;
    const BaCam c{cam->fx, cam->fy, cam->cx, cam->cy, cam->focal_x_baseline, cam->huber_mono, cam->huber_stereo};
    const size_t no = (size_t)std::max(n_obs, 1), np = (size_t)std::max(n_points, 1);
    // dynamic LDS: transposed reduction buffer + as many observations as fit beside it + one activity byte per observation
    constexpr size_t kPoLdsBudget = 150 * 1024;
    const size_t fixed_lds = PO_NV * PO_TR * sizeof(double) + no + 64;
    const int cache_n = (int)std::min<size_t>((size_t)n_obs, fixed_lds < kPoLdsBudget ? (kPoLdsBudget - fixed_lds) / sizeof(PoObs) : 0);
    const size_t lds = PO_NV * PO_TR * sizeof(double) + (size_t)cache_n * sizeof(PoObs) + no + 64;
    if (lds > kPoLdsBudget + 4096) { set_error("pose optimiser: %d observations exceed the LDS activity array", n_obs); return LPSLAM_HIP_ERR_CAPACITY; }
    {
        static std::atomic<bool> po_attr[64];
        int dev = 0; (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && !po_attr[dev].load()) {
            (void)hipFuncSetAttribute((const void*)k_pose_optimize<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPoLdsBudget + 4096));
            (void)hipFuncSetAttribute((const void*)k_pose_optimize<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPoLdsBudget + 4096));
            po_attr[dev].store(true);
        }
    }
    const bool all_cached = cache_n == n_obs;
    // page-locked block: pose (in and out) | inlier count | done flag | then either the packed observations and the outlier bytes
    // (all_cached: the kernel works on this block directly) or a mirror of the device block below
    const size_t off_pts = 128, off_obs = off_pts + 3 * np * sizeof(double), off_out = off_obs + no * sizeof(lpslam_hip_ba_obs);
    const size_t off_packed = 128, off_flags = off_packed + no * sizeof(PoObs);
    const size_t need = all_cached ? off_flags + no : off_out + no;
    if (ctx->h_match_bytes < need) {
        if (ctx->h_match) { LP_HIP(hipStreamSynchronize(s)); (void)hipHostFree(ctx->h_match); }
        ctx->h_match = nullptr; ctx->h_match_bytes = 0;
        LP_HIP(hipHostMalloc((void**)&ctx->h_match, need * 2, hipHostMallocDefault));
        ctx->h_match_bytes = need * 2;
    }
    uint8_t* hb = ctx->h_match;
    memcpy(hb, pose7, 7 * sizeof(double));
    int32_t inl = 0;
    if (all_cached) {
        PoObs* packed = (PoObs*)(hb + off_packed);
        for (int k = 0; k < n_obs; ++k) {
            const lpslam_hip_ba_obs& o = obs[k];
            const double* p = points + 3 * (size_t)o.point;
            packed[k] = PoObs{o.u, o.v, o.ur, o.inv_sigma2, {p[0], p[1], p[2]}};
        }
        int* flag = (int*)(hb + PO_BLK_FLAG);
        const int seq = lp_next_seq(ctx->po_seq);
        __atomic_store_n(flag, 0, __ATOMIC_RELAXED);          // (the block is shared with the matchers' staging: whatever they left here is not a sequence number)
        static const bool four_waves_env = [] { const char* e = getenv("LPSLAM_HIP_PO_FOUR_WAVES"); return e && atoi(e) != 0; }();      // measurements: the round-4 kernel for every size
        static const int po1_max = [] { const char* e = getenv("LPSLAM_HIP_PO_W1_MAX"); const int v = e ? atoi(e) : 256; return std::min(std::max(v, 0), 256); }();      // measurements: where the one-observation-per-lane kernel hands over
        bool delivered = false;
        if (n_obs <= po1_max && !four_waves_env) {
            // a tracked frame: one observation per lane on 1, 2 or 4 wavefronts (k_pose_optimize_req); the camera travels in the block
            memcpy(hb + PO_BLK_CAM, &c, sizeof(BaCam));
            const LpPoseReq req{hb, n_obs, seq};
            // several sessions tracking at once: the request joins the others' in one launch (share.hip) and comes back delivered
            const int shared = lp_share_pose(ctx, req, flag);
            if (shared < 0) return -shared;
            if (shared == LP_SHARE_DONE) delivered = true;
            else { const int rc = lp_launch_pose_batch(s, &req, 1); if (rc) return rc; }
        } else
        hipLaunchKernelGGL(k_pose_optimize<true>, dim3(1), dim3(PO_T), lds, s, (double*)hb, (const double*)nullptr, (const lpslam_hip_ba_obs*)nullptr, packed, n_obs, c,
                           hb + off_flags, (int*)(hb + 56), cache_n, flag, seq);
        LP_HIP(hipGetLastError());
        // the kernel's last store releases `seq`: poll it (a few hundred microseconds at most), fall back to the stream when it does not come
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = delivered;
        for (int spin = 0; !seen; ++spin) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
            lp_poll_pause(spin);
            if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        if (!seen) {
            LP_HIP(hipStreamSynchronize(s));
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { set_error("pose optimiser: the kernel did not complete"); return LPSLAM_HIP_ERR_DEVICE; }
        }
        memcpy(pose7, hb, 7 * sizeof(double));
        memcpy(&inl, hb + 56, sizeof(int));
        memcpy(&ctx->po_passes, hb + 60, sizeof(int));
        {
            static const bool trace = getenv("LPSLAM_HIP_PO_TRACE") != nullptr;
            if (trace) fprintf(stderr, "pose_optimize: %d observations, %d inliers, %d passes, %.1f us\n", n_obs, inl, ctx->po_passes,
                               1e-3 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
        }
        if (outlier && n_obs) memcpy(outlier, hb + off_flags, (size_t)n_obs);
        if (n_inliers) *n_inliers = inl;
        return LPSLAM_HIP_OK;
    }
    // more observations than the LDS holds: one block of the context's cache, pose | n_inliers | points | observations | outlier
    void* blk = nullptr; size_t cap = 0;
    { const int rc = lp_pool_alloc(ctx, off_out + no, &blk, &cap); if (rc) return rc; }
    auto release = [&]() { lp_pool_free(ctx, blk, cap); };
#define PO_HIP(x) do { if ((x) != hipSuccess) { release(); set_error("HIP call failed: %s", #x); return LPSLAM_HIP_ERR_DEVICE; } } while (0)
    uint8_t* base = (uint8_t*)blk;
    double* d_pose = (double*)base; int* d_n = (int*)(base + 64); double* d_pts = (double*)(base + off_pts);
    lpslam_hip_ba_obs* d_obs = (lpslam_hip_ba_obs*)(base + off_obs); uint8_t* d_out = base + off_out;
    if (n_points) memcpy(hb + off_pts, points, 3 * (size_t)n_points * sizeof(double));
    if (n_obs) memcpy(hb + off_obs, obs, (size_t)n_obs * sizeof(lpslam_hip_ba_obs));
    PO_HIP(hipMemcpyAsync(base, hb, off_out, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pose_optimize<false>, dim3(1), dim3(PO_T), lds, s, d_pose, d_pts, d_obs, (const PoObs*)nullptr, n_obs, c, d_out, d_n, cache_n, (int*)nullptr, 0);
    PO_HIP(hipGetLastError());
    PO_HIP(hipMemcpyAsync(hb, base, 128, hipMemcpyDeviceToHost, s));                    // pose and inlier count
    if (outlier && n_obs) PO_HIP(hipMemcpyAsync(hb + off_out, d_out, (size_t)n_obs, hipMemcpyDeviceToHost, s));
    PO_HIP(hipStreamSynchronize(s));
    memcpy(pose7, hb, 7 * sizeof(double));
    memcpy(&inl, hb + 64, sizeof(int));
    memcpy(&ctx->po_passes, hb + 68, sizeof(int));
    if (outlier && n_obs) memcpy(outlier, hb + off_out, (size_t)n_obs);
#undef PO_HIP
    release();
    if (n_inliers) *n_inliers = inl;
    return LPSLAM_HIP_OK;
}

#ifdef LPSLAM_SCHUR_STAMPS
extern "C" __attribute__((visibility("default"))) int lpslam_hip_debug_schur_stamps(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_schur_stamps), (size_t)n * sizeof(unsigned long long)); }
#endif
#ifdef LPSLAM_UPD_STAMPS
extern "C" __attribute__((visibility("default"))) int lpslam_hip_debug_upd_stamps(double* out32) { return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_upd_stamps), 32 * sizeof(double)); }
#endif
#ifdef LPSLAM_PO_STAMPS
extern "C" __attribute__((visibility("default"))) int lpslam_hip_debug_po_stamps(double* out16) { return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_po_stamps), 16 * sizeof(double)); }
#endif

int lpslam_hip_ba_pose_optimize(lpslam_hip_ba* b, uint8_t* outlier, int32_t* n_inliers)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
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
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
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

// lpslam_hip_ba_local for n prepared (not yet built) windows at once: ONE structure build, the two optimisations as batched launch
// chains (blockIdx.y = window), the per-observation chi2 passes and activity masks of all windows between them with one wait each.
// What the mapping threads of several sessions submit together (share.hip); the arithmetic per window is that of the single call.
int lp_ba_local_batch(lpslam_hip_ba* const* ps, int n, int first_iters, int second_iters, uint8_t* const* outliers, double* const* poses_out, double* const* points_out)
{
    int rc = lpslam_hip_ba_build_batch(ps, n); if (rc) return rc;
    hipStream_t s = ps[0]->stream;
    for (int i = 1; i < n; ++i) if (ps[i]->stream != s) { set_error("a shared batch of windows needs them on one stream"); return LPSLAM_HIP_ERR_INVALID; }
    std::vector<std::vector<double>> chi((size_t)n);
    std::vector<std::vector<uint8_t>> pos((size_t)n), active((size_t)n);
    auto chi2_all = [&]() -> int {
        for (int i = 0; i < n; ++i) {
            lpslam_hip_ba* b = ps[i];
            const size_t no = (size_t)std::max(b->n_obs, 1);
            chi[(size_t)i].resize(no); pos[(size_t)i].resize(no);
            if (!b->n_obs) continue;
            hipLaunchKernelGGL(k_ba_obs_chi2, dim3((b->n_obs + 255) / 256, 1), dim3(256), 0, s, b->d_view, b->d_chi_obs, b->d_depth);
            LP_HIP(hipGetLastError());
            LP_HIP(hipMemcpyAsync(chi[(size_t)i].data(), b->d_chi_obs, (size_t)b->n_obs * sizeof(double), hipMemcpyDeviceToHost, s));
            LP_HIP(hipMemcpyAsync(pos[(size_t)i].data(), b->d_depth, (size_t)b->n_obs, hipMemcpyDeviceToHost, s));
        }
        LP_HIP(hipStreamSynchronize(s));
        return LPSLAM_HIP_OK;
    };
    if ((rc = lpslam_hip_ba_optimize_batch(ps, n, 1, first_iters, nullptr, 0, nullptr))) return rc;
    if ((rc = chi2_all())) return rc;
    for (int i = 0; i < n; ++i) {
        lpslam_hip_ba* b = ps[i];
        active[(size_t)i].assign((size_t)std::max(b->n_obs, 1), 1);
        for (int k = 0; k < b->n_obs; ++k) { const double thr = b->h_ur[(size_t)k] < 0 ? 5.99146 : 7.81473; if (thr < chi[(size_t)i][(size_t)k] || !pos[(size_t)i][(size_t)k]) active[(size_t)i][(size_t)k] = 0; }
        if (!b->n_obs) continue;
        LP_HIP(hipMemcpyAsync(b->d_act_in, active[(size_t)i].data(), (size_t)b->n_obs, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_ba_gather_active, dim3((b->n_obs + 255) / 256), dim3(256), 0, s, b->d_act_in, b->d_o_orig, b->d_o_active, b->n_obs);
        LP_HIP(hipGetLastError());
    }
    if ((rc = lpslam_hip_ba_optimize_batch(ps, n, 0, second_iters, nullptr, 0, nullptr))) return rc;
    if ((rc = chi2_all())) return rc;
    for (int i = 0; i < n; ++i) {
        lpslam_hip_ba* b = ps[i];
        if (outliers && outliers[i]) for (int k = 0; k < b->n_obs; ++k) { const double thr = b->h_ur[(size_t)k] < 0 ? 5.99146 : 7.81473; outliers[i][k] = (!active[(size_t)i][(size_t)k]) || (thr < chi[(size_t)i][(size_t)k]) || !pos[(size_t)i][(size_t)k]; }
    }
    if ((rc = lpslam_hip_ba_get_batch(ps, n, poses_out, points_out))) return rc;
    for (int i = 0; i < n; ++i) { release_stage(ps[i]); ps[i]->quiesced = true; }
    return LPSLAM_HIP_OK;
}

extern "C" {

// A keyframe's local bundle adjustment as ONE call: the window is created from the caller's arrays, solved (lpslam_hip_ba_local:
// first_iters with the robust kernel, outlier classification, second_iters without), its state read back into `poses` / `points` and
// destroyed -- what a mapping thread does per keyframe.  With several sessions submitting windows at the same time (shared launches on)
// the windows are built and solved together by one of the calling threads.
int lpslam_hip_ba_local_window(lpslam_hip_ctx* ctx, double* poses, const uint8_t* fixed, int32_t n_poses, double* points, int32_t n_points,
                               const lpslam_hip_ba_obs* obs, int32_t n_obs, const lpslam_hip_ba_camera* cam, int32_t first_iters, int32_t second_iters, uint8_t* outlier)
{
    lpslam_hip_ba* b = nullptr;
    int rc = lpslam_hip_ba_prepare(ctx, poses, fixed, n_poses, points, n_points, obs, n_obs, cam, &b);
    if (rc) return rc;
    const int shared = (ctx->role_solve && b->stream == ctx->role_solve) ? lp_share_ba_local(ctx, b, first_iters, second_iters, outlier, poses, points) : LP_SHARE_DIRECT;
    if (shared < 0) rc = -shared;
    else if (shared == LP_SHARE_DIRECT) {
        lpslam_hip_ba* one[1] = {b};
        rc = lpslam_hip_ba_build_batch(one, 1);
        if (!rc) rc = lpslam_hip_ba_local(b, first_iters, second_iters, outlier);
        if (!rc) rc = lpslam_hip_ba_get(b, poses, points);
    }
    lpslam_hip_ba_destroy(b);
    return rc;
}

}  // extern "C"


// ---- landmark-partitioned global BA driven from C++: RCCL all-reduces enqueued on the problem's own stream ---------------------
// north star: "host code stays C++ ... RCCL all-reduce over xGMI only for the shared-pose normal equations" (SURVEY.md 8(e)).
// Every rank holds all poses and the observations of its landmarks.  One LM trial on the stream, no host synchronisation in it:
//   linearise -> [first trial of a call: SUM (b_p, diag H_pp, chi2) + MAX (max diag H_ll) -> lambda_0] -> partial Schur complement
//   -> pack the lower triangle of S + rhs + b_p + diag H_pp + chi2 -> ONE sum all-reduce (5.9 MB at 200 keyframes instead of the
//   11.8 MB of the dense square) -> unpack -> factor / solve (redundantly on every rank) -> landmark back substitution, trial chi2 ->
//   SUM (trial chi2, landmark scale term) -> accept / reject on the device, identical on every rank.
// RCCL is bound at run time (dlopen): the library has no link-time dependency on it and single-GPU users never load it.
namespace {

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclFloat64 = 8, kNcclSum = 0, kNcclMax = 2;      // rccl.h: ncclFloat64, ncclSum, ncclMax
nccl_allreduce_fn load_nccl_allreduce(std::string* why)
{
    static std::atomic<nccl_allreduce_fn> cached{nullptr};
    nccl_allreduce_fn f = cached.load();
    if (f) return f;
    // the copy already in the process first (a host that links RCCL, or torch's bundled one), then the system library
    void* h = nullptr;
    const char* last = nullptr;
    const struct { const char* name; int flags; } tries[] = {{"librccl.so.1", RTLD_NOW | RTLD_NOLOAD}, {"librccl.so", RTLD_NOW | RTLD_NOLOAD},
        {"librccl.so.1", RTLD_NOW | RTLD_GLOBAL}, {"librccl.so", RTLD_NOW | RTLD_GLOBAL}, {"/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL}};
    for (const auto& t : tries) {
        (void)dlerror();
        h = dlopen(t.name, t.flags);
        if (h) break;
        if (!(t.flags & RTLD_NOLOAD)) { const char* e = dlerror(); if (e) { if (why) *why = e; last = e; } }      // read once: dlerror() clears itself
    }
    if (!h) { if (why && !last) *why = "not found"; return nullptr; }
    (void)dlerror();
    f = (nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
    if (!f) { const char* e = dlerror(); if (why) *why = e ? e : "ncclAllReduce: symbol not found"; return nullptr; }
    cached.store(f);
    return f;
}
struct NcclUser { nccl_allreduce_fn fn; void* comm; };
int nccl_adapter(void* user, void* buf, size_t count, int32_t op, void* stream)
{
    const NcclUser* u = (const NcclUser*)user;
    return u->fn(buf, buf, count, kNcclFloat64, op == LPSLAM_HIP_REDUCE_MAX ? kNcclMax : kNcclSum, u->comm, (hipStream_t)stream);
}

// lower triangle of the dim x dim reduced system (row r, columns 0..r) <-> packed [r (r + 1) / 2 + c]; the tail of the reduced
// buffer (rhs | b_p | diag H_pp | chi2, 3 n + 8 doubles) rides behind it
__global__ __launch_bounds__(256) void k_ba_pack(const BaView* __restrict__ views, double* packed, int unpack)
{
    BaView v = views[0];                                 // one problem; blockIdx.y is the matrix row here
    const int n = v.dim_pad, dim = v.dim;
    const size_t tri = (size_t)dim * (dim + 1) / 2;
    const int r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (r < dim) {
        if (c <= r) {
            const size_t p = (size_t)r * (r + 1) / 2 + c;
            if (unpack) v.S[(size_t)r * n + c] = packed[p]; else packed[p] = v.S[(size_t)r * n + c];
        }
    } else if (r == dim) {
        for (int i = c; i < 3 * n + 8; i += gridDim.x * 256) { if (unpack) v.rhs[i] = packed[tri + i]; else packed[tri + i] = v.rhs[i]; }
    }
}

}  // namespace

extern "C" int lpslam_hip_ba_optimize_partitioned(lpslam_hip_ba* b, void* nccl_comm, int32_t robust, int32_t iters, lpslam_hip_ba_iter_log* log, int32_t* done_out)
{
    if (!b || !nccl_comm) { set_error("null problem / communicator"); return LPSLAM_HIP_ERR_INVALID; }
    std::string why;
    NcclUser u{load_nccl_allreduce(&why), nccl_comm};
    if (!u.fn) { set_error("RCCL (librccl.so) could not be loaded: %s", why.c_str()); return LPSLAM_HIP_ERR_DEVICE; }
    return lpslam_hip_ba_optimize_partitioned_with(b, nccl_adapter, &u, robust, iters, log, done_out);
}

extern "C" int lpslam_hip_ba_optimize_partitioned_with(lpslam_hip_ba* b, lpslam_hip_allreduce_fn allreduce_cb, void* user, int32_t robust, int32_t iters,
                                                       lpslam_hip_ba_iter_log* log, int32_t* done_out)
{
    if (b && !b->built) { set_error("the problem has been prepared but not built (lpslam_hip_ba_build_batch)"); return LPSLAM_HIP_ERR_INVALID; }
    if (!b || !allreduce_cb) { set_error("null problem / all-reduce callback"); return LPSLAM_HIP_ERR_INVALID; }
    if (iters < 0 || iters > MAX_LOG) { set_error("iterations must be in [0,%d]", MAX_LOG); return LPSLAM_HIP_ERR_INVALID; }
    if (b->pending_iters >= 0) { set_error("optimize_begin pending"); return LPSLAM_HIP_ERR_INVALID; }
    auto allreduce = [&](double* buf, size_t count, int op) -> int { return allreduce_cb(user, buf, count, op, (void*)b->stream); };      // in place, on the problem's stream
    LP_HIP(hipSetDevice(b->ctx->cfg.device));
    { const int rd = ensure_dense(b); if (rd) return rd; }      // every rank all-reduces the dense reduced buffer, whatever the window's shape
    hipStream_t s = b->stream;
    const size_t n = (size_t)b->dim_pad, tri = (size_t)b->dim * (b->dim + 1) / 2, packed_n = tri + 3 * n + 8;
    void* pk = nullptr; size_t pk_cap = 0;
    int rc = lp_pool_alloc(b->ctx, packed_n * sizeof(double), &pk, &pk_cap); if (rc) return rc;
    double* d_packed = (double*)pk;
    auto release = [&]() { lp_pool_free(b->ctx, pk, pk_cap); };
#define PT_NCCL(call) do { const int r_ = (call); if (r_ != 0) { (void)hipStreamSynchronize(s); release(); set_error("RCCL call failed (%d): %s", r_, #call); return LPSLAM_HIP_ERR_DEVICE; } } while (0)
#define PT_TRY(x) do { rc = (x); if (rc) { (void)hipStreamSynchronize(s); release(); return rc; } } while (0)
    BaLaunch L = single_launch(b);
    L.robust = robust; b->robust = robust;
    double* tail = b->d_red + n * n;                         // rhs | b_p | diag H_pp | chi2
    PT_TRY(begin_optimize(b, robust, iters));
    auto enqueue_units = [&](int units, bool first_batch) -> int {
        for (int u = 0; u < units; ++u) {
            int r2;
            if ((r2 = enqueue_linearize(L, 0))) return r2;
            if (first_batch && u == 0) {
                // lambda_0 = 1e-5 max diag H over ALL ranks' landmarks and the summed pose blocks: needed before the first Schur complement
                if (allreduce(tail, 3 * n + 8, LPSLAM_HIP_REDUCE_SUM)) { set_error("all-reduce (diagonals) failed"); return LPSLAM_HIP_ERR_DEVICE; }
                if (allreduce(b->d_scal + 4, 1, LPSLAM_HIP_REDUCE_MAX)) { set_error("all-reduce (max diag) failed"); return LPSLAM_HIP_ERR_DEVICE; }
                hipLaunchKernelGGL(k_lm_begin, dim3(1, 1), dim3(64), 0, s, L.d_views);
            }
            if ((r2 = enqueue_reduce(L, 0))) return r2;
            if (b->dim > 0) {
                hipLaunchKernelGGL(k_ba_pack, dim3((b->dim + 255) / 256, b->dim + 1, 1), dim3(256), 0, s, L.d_views, d_packed, 0);
                if (allreduce(d_packed, packed_n, LPSLAM_HIP_REDUCE_SUM)) { set_error("all-reduce (reduced system) failed"); return LPSLAM_HIP_ERR_DEVICE; }
                hipLaunchKernelGGL(k_ba_pack, dim3((b->dim + 255) / 256, b->dim + 1, 1), dim3(256), 0, s, L.d_views, d_packed, 1);
            } else if (allreduce(tail, 3 * n + 8, LPSLAM_HIP_REDUCE_SUM)) { set_error("all-reduce failed"); return LPSLAM_HIP_ERR_DEVICE; }
            if ((r2 = enqueue_solve(L, 0))) return r2;
            if (allreduce(b->d_scal + 1, 2, LPSLAM_HIP_REDUCE_SUM)) { set_error("all-reduce (trial chi2) failed"); return LPSLAM_HIP_ERR_DEVICE; }
            hipLaunchKernelGGL(k_lm_decide, dim3(1, 1), dim3(64), 0, s, L.d_views);
            LP_HIP(hipGetLastError());
        }
        return LPSLAM_HIP_OK;
    };
    const int want_log = (log && b->pin) ? iters : 0;
    if (iters > 0) PT_TRY(enqueue_units(iters, true));
    PT_TRY(read_ctl(b, want_log));
    // every rank sees the same control block (identical inputs after every all-reduce), so every rank runs the same number of units
    for (int guard = 0; iters > 0 && !b->h_ctl.stopped && b->h_ctl.outer_done < iters && guard < 16 * MAX_LOG; ++guard) {
        PT_TRY(enqueue_units(iters - b->h_ctl.outer_done, false));
        PT_TRY(read_ctl(b, want_log));
    }
#undef PT_NCCL
#undef PT_TRY
    release();
    if ((rc = report_faults(b))) return rc;
    const int done = b->h_ctl.outer_done;
    if (log && done) {
        if (want_log) memcpy(log, b->pin->log, (size_t)std::min(done, MAX_LOG) * sizeof(lpslam_hip_ba_iter_log));
        else LP_HIP(hipMemcpy(log, b->d_log, std::min(done, MAX_LOG) * sizeof(lpslam_hip_ba_iter_log), hipMemcpyDeviceToHost));
    }
    if (done_out) *done_out = done;
    return LPSLAM_HIP_OK;
}

#include "sim3.inl"
