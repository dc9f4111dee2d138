// ba_update.inl -- the rest of an LM trial behind the reduced solve, in ONE launch (included by ba.hip).
//
// [UPSTREAM] g2o BlockSolver::solve (back substitution x_l = H_ll^-1 (b_l - W^T x_p)), OptimizableGraph::push / oplus of every
// vertex, activeRobustChi2 of the trial state and OptimizationAlgorithmLevenberg::solve's accept / reject
// (g2o@691dc51; reached from the reference through the mapping thread openvslam::system starts,
// /root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:238-239).
//
// Rounds 1-4 ran this as two launches: k_ba_backsub (landmarks, trial poses) and k_ba_trial (trial chi2 by keyframe, the lambda control,
// and -- beside them, on speculation -- the complete linearisation of the trial state).  Both walk the same observations and the second
// reads back what the first wrote, with a launch boundary in between.  k_ba_update is one landmark-grouped pass, a workgroup per block of
// <= LAND_B landmarks and (normally) <= 256 observations, one observation per thread, every position-dependent load issued once:
//     0. the trial poses exp(x_p) * pose of every keyframe into LDS (one thread per keyframe; block 0 also stores them)
//     1. r_j = sum W^T x_p over the landmark's observations (ordered sums through LDS), x_l = (H_ll + lambda)^-1 (b_l - r_j): the trial
//        landmark stays in LDS (and is stored)
//     2. the same observations at the trial state: residual -> robust chi2 (the trial's cost), Jacobians -> W and the ordered sums H_ll,
//        b_l of the OTHER linearisation set (a rejected trial leaves the accepted state's set alone)
//     the block that finishes last adds the blocks' chi2 and scale terms in block order and runs g2o's lambda control (lm_decide).
// What is NOT needed for the decision -- H_pp, b_p of the new state, sums per KEYFRAME -- is not computed here: when the trial is
// accepted the deciding block raises the problem's "pose side pending" word and the Schur kernel that opens the next trial computes
// them in its leading workgroups (ba_pose_side_wave: SPLIT wavefronts per keyframe, as the explicit linearisation does), beside the
// landmark groups / pair lists; their consumers -- the band reduction in the launch behind it, the diagonal and rhs blocks of
// k_ba_schur in the same launch -- find them complete (the latter wait on a count; the producers precede them in dispatch order and
// wait for nobody, so the wait cannot deadlock; it is bounded all the same and a time-out is counted, lpslam_hip_ba_timeouts).
// A first version kept those keyframe blocks inside this launch, waiting for the trial landmarks: the write-through stores, the drain
// and the barrier of that hand-over sat on every landmark block's path (2.2 us of 20, in-kernel stamps) and the keyframe blocks ended
// the launch -- 26.7 us against 26.8 for the two launches it replaced.
// Sums are ordered: the same bytes alone or in a batch.  The partitioned (all-reduced) solve keeps the two-launch form.

#ifdef LPSLAM_UPD_STAMPS
// development: wall-clock stamps (100 MHz) of landmark block 0 [0..15], the block that decides [16..23] and keyframe block 0 [24..31]
__device__ double g_upd_stamps[32];
// (the clock as an asm volatile statement with a memory clobber: the builtin floats against the surrounding code)
__device__ __forceinline__ unsigned long long upd_clk() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define UPD_STAMP(cond, k) do { if ((cond) && threadIdx.x == 0) g_upd_stamps[k] = (double)upd_clk(); } while (0)
#else
#define UPD_STAMP(cond, k) do {} while (0)
#endif

// One CSR entry (an observation in landmark-major order) as a landmark block holds it across both passes: every load that depends on
// the entry's position alone is issued once, at the top of the kernel, beside the trial poses' arithmetic.
struct UpdEntry { int k, p, slot, jl; double u, v, ur, w; };
__device__ __forceinline__ UpdEntry upd_entry(const BaView& v, int s, int j0)
{
    const size_t cs = csr_stride(v.n_obs);
    GPTR(const int) c_pose = (GPTR(const int))(v.csr + 4 * cs); GPTR(const int) c_point = c_pose + cs; GPTR(const int) c_slot = (GPTR(const int))(v.csr + 5 * cs);
    UpdEntry e;
    e.k = v.pt_obs[s]; e.p = c_pose[s]; e.slot = c_slot[s]; e.jl = c_point[s] - j0;
    e.u = v.csr[s]; e.v = v.csr[cs + s]; e.ur = v.csr[2 * cs + s]; e.w = v.csr[3 * cs + s];
    return e;
}

__device__ __forceinline__ void upd_land_body(BaView& v, int bid, int robust, int points_fixed, int cur, double lambda)
{
    __shared__ double sh[256 * 9];
    __shared__ double s_pose[UPD_MAXP * 7];
    __shared__ double s_pt[LAND_B * 3], s_q[LAND_B * 3];
    __shared__ int s_start[LAND_B + 1];
    __shared__ double s_red[8];
    const int tid = threadIdx.x, nxt = cur ^ 1;
    // the block's landmarks [j0, j1) and CSR entries [s_lo, s_hi): at most LAND_B landmarks and (unless one landmark alone has more) 256
    // entries, cut on the host at creation (land_start: (first landmark, first entry) per block)
    GPTR(const int2) ls = reinterpret_cast<GPTR(const int2)>(v.land_start);
    const int2 lb0 = ls[bid], lb1 = ls[bid + 1];
    const int j0 = lb0.x, j1 = lb1.x, s_lo = lb0.y, s_hi = lb1.y;
    GPTR(const double) poses_old = sel2(v.poses_buf[0], v.poses_buf[1], cur);
    GPTR(const double) points_old = sel2(v.points_buf[0], v.points_buf[1], cur);
    GPTR(double) poses_out = sel2(v.poses_buf[0], v.poses_buf[1], nxt);
    GPTR(double) points_out = sel2(v.points_buf[0], v.points_buf[1], nxt);
    const SetOff so = set_offsets(v.n_poses, v.n_points, v.n_obs, v.n_free, v.dim_pad);
    GPTR(const double) d_old = sel2(v.set_d[0], v.set_d[1], cur);
    GPTR(double) d_new = sel2(v.set_d[0], v.set_d[1], nxt);
    GPTR(const double) W_old = d_old + so.W; GPTR(const double) Hll_old = d_old; GPTR(const double) bl_old = d_old + so.bl;
    GPTR(double) W_new = d_new + so.W; GPTR(double) Hll_new = d_new; GPTR(double) bl_new = d_new + so.bl;
    const int l = tid >> 3, c = tid & 7;                   // the sums: landmark j0 + l, component c (and 8 with c == 0)
    UPD_STAMP(bid == 0, 0);
    // ---- everything that depends on positions alone, in flight together: this thread's entry of the first chunk, the landmark's old
    //      block and rhs (thread (l, 0)), the segment starts
    const bool have0 = s_lo + tid < s_hi;
    UpdEntry e0 = upd_entry(v, have0 ? s_lo + tid : max(s_hi - 1, 0), j0);
    if (v.n_obs == 0) { e0.k = 0; e0.p = 0; e0.slot = -1; e0.jl = 0; }
    double hraw[6] = {0, 0, 0, 0, 0, 0}, b_old[3] = {0, 0, 0}, x_old[3] = {0, 0, 0};
    const bool solver_thread = c == 0 && j0 + l < j1;
    if (solver_thread) {
        const size_t j = (size_t)(j0 + l);
#pragma unroll
        for (int i = 0; i < 6; ++i) hraw[i] = Hll_old[6 * j + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) { b_old[i] = bl_old[3 * j + i]; x_old[i] = points_old[3 * j + i]; }
    }
    if (tid <= LAND_B) s_start[tid] = v.pt_start[min(j0 + tid, j1)];
    UPD_STAMP(bid == 0, 1);
    // ---- 0. trial poses (one thread per keyframe; the loads above are on their way meanwhile)
    for (int p = tid; p < v.n_poses; p += 256) {
        double pin[7], pout[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) pin[i] = poses_old[7 * p + i];
        const int slot = v.pose_slot[p];
        if (slot >= 0) {
            double d[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = v.xp[6 * slot + i];
            po_oplus(pin, d, pout);
        } else {
#pragma unroll
            for (int i = 0; i < 7; ++i) pout[i] = pin[i];
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) s_pose[7 * p + i] = pout[i];
        if (bid == 0) {
#pragma unroll
            for (int i = 0; i < 7; ++i) poses_out[7 * p + i] = pout[i];
        }
    }
    UPD_STAMP(bid == 0, 2);
    // what depends on the entry: its W row, its keyframe's update, its activity flag
    const bool act0 = have0 && v.o_active[e0.k] != 0;
    // ---- 1. back substitution: r = sum W^T x_p, per landmark in CSR order
    auto back_term = [&](const UpdEntry& e, bool have, double (&rs)[3]) __attribute__((always_inline)) {
        rs[0] = rs[1] = rs[2] = 0;
        if (have && e.slot >= 0) {
            const double2* Wa = reinterpret_cast<const double2*>(W_old + 18 * (size_t)e.k);
            double w[18], x[6];
#pragma unroll
            for (int q = 0; q < 9; ++q) { const double2 a2 = Wa[q]; w[2 * q] = a2.x; w[2 * q + 1] = a2.y; }
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = v.xp[6 * e.slot + i];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) rs[cc] += w[rr * 3 + cc] * x[rr];
        }
    };
    double racc = 0;
    {
        double rs[3];
        back_term(e0, have0, rs);
#pragma unroll
        for (int i = 0; i < 3; ++i) sh[tid * 3 + i] = rs[i];
    }
    UPD_STAMP(bid == 0, 3);
    __syncthreads();                                        // (also: s_start, s_pose)
    UPD_STAMP(bid == 0, 4);
    const int seg_lo = s_start[min(l, j1 - j0)], seg_hi = s_start[min(l + 1, j1 - j0)];
    for (int chunk = s_lo; chunk < s_hi; chunk += 256) {
        if (chunk > s_lo) {                                 // a landmark with more than 256 observations: further chunks, loaded as they come
            const bool have = chunk + tid < s_hi;
            const UpdEntry e = upd_entry(v, have ? chunk + tid : s_hi - 1, j0);
            double rs[3];
            back_term(e, have, rs);
#pragma unroll
            for (int i = 0; i < 3; ++i) sh[tid * 3 + i] = rs[i];
            __syncthreads();
        }
        if (c < 3) {
            const int a0 = max(seg_lo, chunk), a1 = min(seg_hi, chunk + 256);
            for (int t = a0; t < a1; ++t) racc += sh[(t - chunk) * 3 + c];
        }
        __syncthreads();
    }
    if (c < 3) s_q[l * 3 + c] = racc;
    __syncthreads();
    UPD_STAMP(bid == 0, 5);
    double sc = 0;
    if (solver_thread) {
        const size_t j = (size_t)(j0 + l);
        const double q0 = b_old[0] - s_q[l * 3], q1 = b_old[1] - s_q[l * 3 + 1], q2 = b_old[2] - s_q[l * 3 + 2];
        double h[6];
        point_hinv(hraw, lambda, h);
        const double x0 = h[0] * q0 + h[1] * q1 + h[2] * q2;
        const double x1 = h[1] * q0 + h[3] * q1 + h[4] * q2;
        const double x2 = h[2] * q0 + h[4] * q1 + h[5] * q2;
        const double n0 = x_old[0] + x0, n1 = x_old[1] + x1, n2 = x_old[2] + x2;
        s_pt[l * 3] = n0; s_pt[l * 3 + 1] = n1; s_pt[l * 3 + 2] = n2;
        points_out[3 * j] = n0; points_out[3 * j + 1] = n1; points_out[3 * j + 2] = n2;
        sc = x0 * (lambda * x0 + b_old[0]) + x1 * (lambda * x1 + b_old[1]) + x2 * (lambda * x2 + b_old[2]);
    }
    sc = wave_sum(sc);
    if ((tid & 63) == 0) s_red[tid >> 6] = sc;
    UPD_STAMP(bid == 0, 6);
    __syncthreads();                                        // the trial landmarks are in LDS
    UPD_STAMP(bid == 0, 7);
    if (tid == 0) st_sc1(&v.part[bid], (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
    // ---- 2. the trial state: chi2, and its linearisation (observation side) into the other set
    double acc = 0, acc8 = 0;
#ifdef LPSLAM_UPD_REP2
  for (int rep = 0; rep < 2; ++rep) {       // development: the pass a second time (hot instruction cache, same data) -- results are wrong, only the stamps count
    if (rep == 1) { UPD_STAMP(bid == 0, 12); acc = 0; acc8 = 0; }
#endif
    for (int chunk = s_lo; chunk < s_hi; chunk += 256) {
        const bool have = chunk + tid < s_hi;
        UpdEntry e = e0;
        bool act = act0;
        if (chunk > s_lo) { e = upd_entry(v, have ? chunk + tid : s_hi - 1, j0); act = have && v.o_active[e.k] != 0; }
        double hs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double chi = 0;
        if (have) {
            double* Wk = W_new + 18 * (size_t)e.k;
            bool lin = false;
            double R[9], er[3], pc[3], w = 0;
            int D = 2;
            if (act) {
                const double X[3] = {s_pt[3 * e.jl], s_pt[3 * e.jl + 1], s_pt[3 * e.jl + 2]};
                double rho0;
                quat_to_rot(s_pose + 7 * e.p, R);
                D = ba_residual_vals(v.cam, e.u, e.v, e.ur, R, s_pose + 7 * e.p + 4, X, er, pc);
                w = ba_weight_vals(v.cam, e.w, D, er, robust, &rho0);
                chi = rho0;
                lin = !points_fixed;
            }
            if (!lin) {
#pragma unroll
                for (int i = 0; i < 18; ++i) Wk[i] = 0.0;
            } else {
                UPD_STAMP(bid == 0, 14);
                double A[3][3], B[3][6];
                ba_jacobians(v.cam, R, pc, D, A, B);
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
                    for (int r = 0; r < 3; ++r) s3 += A[r][a] * (-w * er[r]);
                    hs[6 + a] = s3;
                }
                UPD_STAMP(bid == 0, 15);
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        double s2 = 0;
                        if (e.slot >= 0) {
#pragma unroll
                            for (int r = 0; r < 3; ++r) s2 += B[r][a] * w * A[r][cc];
                        }
                        Wk[a * 3 + cc] = s2;
                    }
            }
        }
        UPD_STAMP(bid == 0, 8);
#pragma unroll
        for (int i = 0; i < 9; ++i) sh[tid * 9 + i] = hs[i];
        // the chunk's chi2: butterfly per wavefront, the wavefronts in order, the chunks in order
        const double cw = wave_sum(chi);
        if ((tid & 63) == 0) s_red[4 + (tid >> 6)] = cw;
        __syncthreads();
        const int a0 = max(seg_lo, chunk), a1 = min(seg_hi, chunk + 256);
        for (int t = a0; t < a1; ++t) {
            acc += sh[(t - chunk) * 9 + c];
            if (c == 0) acc8 += sh[(t - chunk) * 9 + 8];
        }
        if (tid == 0) s_red[0] = (chunk == s_lo ? 0.0 : s_red[0]) + ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
        __syncthreads();
    }
#ifdef LPSLAM_UPD_REP2
    if (rep == 1) UPD_STAMP(bid == 0, 13);
  }
#endif
    {
        const int j = j0 + l;
        if (j < j1) {
            if (c < 6) Hll_new[6 * (size_t)j + c] = acc; else bl_new[3 * (size_t)j + (c - 6)] = acc;
            if (c == 0) bl_new[3 * (size_t)j + 2] = acc8;
        }
    }
    if (tid == 0) st_sc1(&v.part[v.land_blocks + bid], s_lo < s_hi ? s_red[0] : 0.0);
    UPD_STAMP(bid == 0, 9);
}

// H_pp, b_p of state `cur` for keyframe p, slice sp (one wavefront) into set `cur`'s partials: the pose side of a linearisation, as
// pose_part_body computes it, for the Schur kernels' leading workgroups.  publish: the 27 sums are stored write-through and the
// wavefront counts itself into sync word [4] (k_ba_schur's diagonal and rhs blocks, in the same launch, wait for that count).
// sum over the 64 lanes of a wavefront, in every lane: DPP inside the 16-lane rows (xor 1, xor 2, half mirror, mirror), then the four
// rows through scalar registers -- 12 DPP moves, 8 lane reads and 7 additions where the butterfly of wave_sum takes 12 trips through
// the LDS crossbar; a fixed tree
template <int CTRL>
__device__ __forceinline__ double upd_dpp(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double upd_lane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double upd_wave_sum(double v)
{
    v += upd_dpp<0xB1>(v); v += upd_dpp<0x4E>(v); v += upd_dpp<0x141>(v); v += upd_dpp<0x140>(v);
    return (upd_lane(v, 0) + upd_lane(v, 16)) + (upd_lane(v, 32) + upd_lane(v, 48));
}

constexpr int UPD_PB_MAX = 2;                   // observations per lane whose loads are in flight together (a keyframe with more than 64 * SPLIT * UPD_PB = 1024 takes further rounds)
template <int UPD_PB>
__device__ __forceinline__ void ba_pose_side_wave(BaView& v, int p, int sp, int robust, int cur, bool publish)
{
    static_assert(UPD_PB >= 1 && UPD_PB <= UPD_PB_MAX, "observations in flight per lane");
    const int lane = threadIdx.x & 63;
    if (p >= v.n_poses) return;
    const int slot = v.pose_slot[p];
    if (slot < 0) return;                                   // a fixed keyframe has no block of the system
    GPTR(const double) poses = sel2(v.poses_buf[0], v.poses_buf[1], cur);
    GPTR(const double) points = sel2(v.points_buf[0], v.points_buf[1], cur);
    GPTR(double) partial = sel2(v.set_z[0], v.set_z[1], cur);
    const int s_end = v.ps_start[p + 1];
    double pose[7], R[9];
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = poses[7 * p + i];
    double h[21], b[6];
#pragma unroll
    for (int i = 0; i < 21; ++i) h[i] = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = 0;
    bool first = true;
    for (int s0 = v.ps_start[p] + sp * 64 + lane; s0 < s_end || first; s0 += 64 * SPLIT * UPD_PB) {
        // observations are stored keyframe by keyframe: index = position.  UPD_PB of them per lane: flags, landmark indices and constants
        // first, then the landmarks, then the arithmetic
        bool on[UPD_PB]; int jj[UPD_PB]; double ou[UPD_PB], ov[UPD_PB], our[UPD_PB], ow[UPD_PB], X[UPD_PB][3];
#pragma unroll
        for (int u = 0; u < UPD_PB; ++u) {
            const int s = s0 + u * 64 * SPLIT;
            const int k = max(min(s, s_end - 1), 0);
            on[u] = s < s_end && v.o_active[k] != 0;
            jj[u] = v.o_point[k]; ou[u] = v.o_u[k]; ov[u] = v.o_v[k]; our[u] = v.o_ur[k]; ow[u] = v.o_w[k];
        }
#pragma unroll
        for (int u = 0; u < UPD_PB; ++u)
#pragma unroll
            for (int i = 0; i < 3; ++i) X[u][i] = points[3 * (size_t)jj[u] + i];
        if (first) { quat_to_rot(pose, R); first = false; }
#pragma unroll
        for (int u = 0; u < UPD_PB; ++u) {
            if (!on[u]) continue;
            double e[3], pc[3], rho0;
            const int D = ba_residual_vals(v.cam, ou[u], ov[u], our[u], R, pose + 4, X[u], e, pc);
            const double w = ba_weight_vals(v.cam, ow[u], D, e, robust, &rho0);
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
    }
    double* out = partial + ((size_t)slot * SPLIT + sp) * PV;
#pragma unroll
    for (int i = 0; i < 21; ++i) h[i] = upd_wave_sum(h[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = upd_wave_sum(b[i]);
    if (lane == 0) {
        if (publish) {
#pragma unroll
            for (int i = 0; i < 21; ++i) st_sc1(out + i, h[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) st_sc1(out + 21 + i, b[i]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(ba_sync_words(v) + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
            for (int i = 0; i < 21; ++i) out[i] = h[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) out[21 + i] = b[i];
        }
    }
}
// consumer side (k_ba_schur's diagonal / rhs blocks, one wavefront each): every producing wavefront of this launch has counted itself
__device__ __forceinline__ void ba_pose_side_wait(const BaView& v)
{
    const int need = v.n_free * SPLIT;
    if ((threadIdx.x & 63) == 0) {
        int spins = 0;
        while (__hip_atomic_load(ba_sync_words(v) + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
        if (spins >= (1 << 22)) __hip_atomic_fetch_add(ba_sync_words(v) + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void k_ba_update(const BaView* __restrict__ views, int robust, int points_fixed)
{
    BA_VIEW(v);
    BA_VIEW_HEAD("s"(v.land_blocks), "s"(v.ctl));
    const int land_blocks = v.land_blocks;
    UPD_STAMP(blockIdx.x == 0, 10);
    if ((int)blockIdx.x >= land_blocks || !upd_takes(v.n_points, v.n_free, v.n_poses)) return;
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle()) return;
    const int cur = fl.cur;                                // (nothing flips it before the last block's decision, which no other block reads behind)
    upd_land_body(v, blockIdx.x, robust, points_fixed, cur, fl.lambda);
    UPD_STAMP(blockIdx.x == 0, 11);
    if (!ba_last_block_sc1(v.ctl, land_blocks)) return;
    UPD_STAMP(true, 16);
    // ---- the landmark block that finishes last: totals in block order, then g2o's accept / reject
    // (every load of a wavefront is in flight before the first is used: written as a loop over 64-strided elements the three trips to
    // memory of 157 blocks came one after the other, 1.4 us of a 15.5 us kernel; the order of the additions is the loop's)
    __shared__ double s_tot[3];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    auto strided_sum = [&](const double* src, int n) -> double {
        double a = 0;
        for (int i0 = 0; i0 < n; i0 += 256) {
            double t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + lane + 64 * u; t[u] = i < n ? ld_sc1(&src[i]) : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i0 + lane + 64 * u < n) a += t[u];
        }
        return wave_sum(a);
    };
    BaCtl c0;                                              // thread 0: the control block and the factorisation's verdict travel beside the partials
    double fail0 = 0;
    if (tid == 0) { c0 = *v.ctl; fail0 = v.scal[5]; }
    if (wave == 0) {
        const double a = strided_sum(v.part + land_blocks, land_blocks);
        if (lane == 0) s_tot[0] = a;
    } else if (wave == 1) {
        const double a = strided_sum(v.part, land_blocks);
        if (lane == 0) s_tot[1] = a;
    } else if (wave == 2) {
        double a = 0;
        const double lambda = fl.lambda;
        for (int p0 = 0; p0 < v.n_poses; p0 += 64) {
            const int p = p0 + lane;
            const int slot = p < v.n_poses ? v.pose_slot[p] : -1;
            double x[6], b[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) { x[q] = slot >= 0 ? v.xp[6 * slot + q] : 0.0; b[q] = slot >= 0 ? v.bp[6 * slot + q] : 0.0; }
            if (slot < 0) continue;
#pragma unroll
            for (int q = 0; q < 6; ++q) a += x[q] * (lambda * x[q] + b[q]);
        }
        a = wave_sum(a);
        if (lane == 0) s_tot[2] = a;
    }
    __syncthreads();
    UPD_STAMP(true, 17);
    if (tid == 0) {
        v.scal[1] = s_tot[0]; v.scal[2] = s_tot[1]; v.scal[3] = s_tot[2];
        const int accepted = lm_decide(v, s_tot[0], fail0, s_tot[1], s_tot[2], true, &c0);
        // the pose side of the new state's linearisation is the next Schur launch's to compute -- when there is a new state
        int* sw = ba_sync_words(v);
        sw[3] = accepted; sw[4] = 0; sw[2] = 0;      // ([2]: groups of the next Schur launch that have stored their share, ba_band.inl)
    }
    UPD_STAMP(true, 18);
}
