// ba_update.inl -- the rest of an LM trial behind the reduced solve, in ONE launch (included by ba.hip).
//
// [UPSTREAM] g2o BlockSolver::solve (back substitution x_l = H_ll^-1 (b_l - W^T x_p)), OptimizableGraph::push / oplus of every
// vertex, activeRobustChi2 of the trial state and OptimizationAlgorithmLevenberg::solve's accept / reject
// (g2o@691dc51; reached from the reference through the mapping thread openvslam::system starts,
// /root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:238-239).
//
// Rounds 1-4 ran this as two launches: k_ba_backsub (landmarks, trial poses) and k_ba_trial (trial chi2 by keyframe, the lambda control,
// and -- beside them, on speculation -- the complete linearisation of the trial state).  Both walk the same observations, the second
// reads back what the first wrote, and between them sit a launch boundary and two ramps: 8.5 + 12.2 us + the gap of a 95 us iteration.
// k_ba_update is one landmark-grouped pass:
//
//   landmark workgroups (blocks [0, land_blocks): LAND_B landmarks each, their observations in CSR order)
//     0. the trial poses exp(x_p) * pose of every keyframe into LDS (one thread per keyframe; block 0 also stores them)
//     1. r_j = sum W^T x_p over the landmark's observations (ordered sums through LDS), x_l = (H_ll + lambda)^-1 (b_l - r_j): the trial
//        landmark stays in LDS and is stored write-through; the block then counts itself into the ready word (ba_sync_words)
//     2. the same observations once more at the trial state: residual -> robust chi2 (the trial's cost), Jacobians -> W and the ordered
//        sums H_ll, b_l of the OTHER linearisation set (a rejected trial leaves the accepted state's set alone)
//     the block that finishes last adds the blocks' chi2 and scale terms in block order and runs g2o's lambda control (lm_decide)
//   keyframe workgroups (blocks [land_blocks, land_blocks + pose_blocks): SPLIT wavefronts per keyframe)
//     wait until every landmark block has published its trial landmarks (the ready word (ba_sync_words) == land_blocks), then H_pp, b_p of the trial
//     state per (keyframe, slice) into the other set's partials -- read by the NEXT launch only (the Schur kernels add the slices up),
//     so they are off the path to the decision.
//
// The wait cannot deadlock: a keyframe block waits for landmark blocks of its own problem only, which precede it in dispatch order
// and wait for nobody (induction over the dispatch index, whatever else fills the compute units).  It is bounded all the same; a
// time-out is counted in the problem's fault words (lpslam_hip_ba_timeouts) and reported by optimize_end.
// the ready word (ba_sync_words) and ctl->cur_launch are set by the Schur kernel that opens the trial (k_ba_schur / k_schur_group).
// Sums are ordered: the same bytes alone or in a batch.  The partitioned (all-reduced) solve keeps the two-launch form.


__device__ __forceinline__ void upd_land_body(BaView& v, int bid, int robust, int points_fixed, int cur, double lambda)
{
    __shared__ double sh[256 * 9];
    __shared__ double s_pose[UPD_MAXP * 7];
    __shared__ double s_pt[LAND_B * 3], s_q[LAND_B * 3];
    __shared__ int s_start[LAND_B + 1];
    __shared__ double s_red[8];
    const int tid = threadIdx.x, nxt = cur ^ 1;
    const int j0 = bid * LAND_B, j1 = min(j0 + LAND_B, v.n_points);
    GPTR(const double) poses_old = sel2(v.poses_buf[0], v.poses_buf[1], cur);
    GPTR(const double) points_old = sel2(v.points_buf[0], v.points_buf[1], cur);
    GPTR(double) poses_out = sel2(v.poses_buf[0], v.poses_buf[1], nxt);
    GPTR(double) points_out = sel2(v.points_buf[0], v.points_buf[1], nxt);
    const SetOff so = set_offsets(v.n_poses, v.n_points, v.n_obs, v.n_free, v.dim_pad);
    GPTR(const double) d_old = sel2(v.set_d[0], v.set_d[1], cur);
    GPTR(double) d_new = sel2(v.set_d[0], v.set_d[1], nxt);
    GPTR(const double) W_old = d_old + so.W; GPTR(const double) Hll_old = d_old; GPTR(const double) bl_old = d_old + so.bl;
    GPTR(double) W_new = d_new + so.W; GPTR(double) Hll_new = d_new; GPTR(double) bl_new = d_new + so.bl;
    // ---- 0. trial poses
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
            for (int i = 0; i < 7; ++i) st_sc1(poses_out + 7 * p + i, pout[i]);
        }
    }
    if (tid <= LAND_B) s_start[tid] = v.pt_start[min(j0 + tid, j1)];
    __syncthreads();
    const int s_lo = s_start[0], s_hi = s_start[j1 - j0];
    const size_t cs = csr_stride(v.n_obs);
    GPTR(const double) c_u = v.csr; GPTR(const double) c_v = v.csr + cs; GPTR(const double) c_ur = v.csr + 2 * cs; GPTR(const double) c_w = v.csr + 3 * cs;
    GPTR(const int) c_pose = (GPTR(const int))(v.csr + 4 * cs); GPTR(const int) c_point = c_pose + cs;
    const int l = tid >> 3, c = tid & 7;                   // the sums: landmark j0 + l, component c (and 8 with c == 0)
    const int seg_lo = s_start[min(l, j1 - j0)], seg_hi = s_start[min(l + 1, j1 - j0)];
    // ---- 1. back substitution: r = sum W^T x_p, per landmark in CSR order
    double racc = 0;
    for (int chunk = s_lo; chunk < s_hi; chunk += 256) {
        const int s = chunk + tid;
        double rs[3] = {0, 0, 0};
        if (s < s_hi) {
            const int k = v.pt_obs[s];
            const int slot = v.pose_slot[c_pose[s]];
            if (slot >= 0) {
                const double2* Wa = reinterpret_cast<const double2*>(W_old + 18 * (size_t)k);
                double w[18], x[6];
#pragma unroll
                for (int q = 0; q < 9; ++q) { const double2 a2 = Wa[q]; w[2 * q] = a2.x; w[2 * q + 1] = a2.y; }
#pragma unroll
                for (int i = 0; i < 6; ++i) x[i] = v.xp[6 * slot + i];
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) rs[cc] += w[rr * 3 + cc] * x[rr];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) sh[tid * 3 + i] = rs[i];
        __syncthreads();
        if (c < 3) {
            const int a0 = max(seg_lo, chunk), a1 = min(seg_hi, chunk + 256);
            for (int t = a0; t < a1; ++t) racc += sh[(t - chunk) * 3 + c];
        }
        __syncthreads();
    }
    if (c < 3) s_q[l * 3 + c] = racc;
    __syncthreads();
    double sc = 0;
    {
        const int j = j0 + l;
        if (c == 0 && j < j1) {
            const double b0 = bl_old[3 * (size_t)j], b1 = bl_old[3 * (size_t)j + 1], b2 = bl_old[3 * (size_t)j + 2];
            const double q0 = b0 - s_q[l * 3], q1 = b1 - s_q[l * 3 + 1], q2 = b2 - s_q[l * 3 + 2];
            double hraw[6], h[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) hraw[i] = Hll_old[6 * (size_t)j + i];
            point_hinv(hraw, lambda, h);
            const double x0 = h[0] * q0 + h[1] * q1 + h[2] * q2;
            const double x1 = h[1] * q0 + h[3] * q1 + h[4] * q2;
            const double x2 = h[2] * q0 + h[4] * q1 + h[5] * q2;
            const double n0 = points_old[3 * (size_t)j] + x0, n1 = points_old[3 * (size_t)j + 1] + x1, n2 = points_old[3 * (size_t)j + 2] + x2;
            s_pt[l * 3] = n0; s_pt[l * 3 + 1] = n1; s_pt[l * 3 + 2] = n2;
            st_sc1(points_out + 3 * (size_t)j, n0); st_sc1(points_out + 3 * (size_t)j + 1, n1); st_sc1(points_out + 3 * (size_t)j + 2, n2);
            sc = x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
        }
    }
    sc = wave_sum(sc);
    if ((tid & 63) == 0) s_red[tid >> 6] = sc;
    // publish: the trial landmarks (and, block 0, the trial poses) have left this compute unit before the count goes up
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(ba_sync_words(v) + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st_sc1(&v.part[bid], (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
    }
    // ---- 2. the trial state: chi2, and its linearisation (observation side) into the other set
    double acc = 0, acc8 = 0, chi = 0;
    for (int chunk = s_lo; chunk < s_hi; chunk += 256) {
        const int s = chunk + tid;
        double hs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (s < s_hi) {
            const int k = v.pt_obs[s];
            const int p = c_pose[s];
            const int slot = v.pose_slot[p];
            double* Wk = W_new + 18 * (size_t)k;
            bool lin = false;
            double R[9], e[3], pc[3], w = 0;
            int D = 2;
            if (v.o_active[k]) {
                const int jl = c_point[s] - j0;
                const double X[3] = {s_pt[3 * jl], s_pt[3 * jl + 1], s_pt[3 * jl + 2]};
                double rho0;
                quat_to_rot(s_pose + 7 * p, R);
                D = ba_residual_vals(v.cam, c_u[s], c_v[s], c_ur[s], R, s_pose + 7 * p + 4, X, e, pc);
                w = ba_weight_vals(v.cam, c_w[s], D, e, robust, &rho0);
                chi = rho0;
                lin = !points_fixed;
            }
            if (!lin) {
#pragma unroll
                for (int i = 0; i < 18; ++i) Wk[i] = 0.0;
            } else {
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
        // the chunk's chi2, in thread order per wavefront and wavefront order per chunk
        const double cw = wave_sum(chi);
        chi = 0;
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
    {
        const int j = j0 + l;
        if (j < j1) {
            if (c < 6) Hll_new[6 * (size_t)j + c] = acc; else bl_new[3 * (size_t)j + (c - 6)] = acc;
            if (c == 0) bl_new[3 * (size_t)j + 2] = acc8;
        }
    }
    if (tid == 0) st_sc1(&v.part[v.land_blocks + bid], s_lo < s_hi ? s_red[0] : 0.0);
}

// H_pp, b_p of the trial state per (keyframe, slice) into the other set's partials, once every trial landmark is published
__device__ __forceinline__ void upd_pose_body(BaView& v, int bid, int robust, int cur)
{
    const int nxt = cur ^ 1;
    const int lane = threadIdx.x & 63;
    const int wv = bid * 4 + (threadIdx.x >> 6);
    const int p = wv / SPLIT, sp = wv - p * SPLIT;
    if (p >= v.n_poses) return;
    const int slot = v.pose_slot[p];
    if (slot < 0) return;                                   // a fixed keyframe has no block of the system (its pose was copied by landmark block 0)
    GPTR(const double) poses_new = sel2(v.poses_buf[0], v.poses_buf[1], nxt);
    GPTR(const double) points_new = sel2(v.points_buf[0], v.points_buf[1], nxt);
    GPTR(double) partial_new = sel2(v.set_z[0], v.set_z[1], nxt);
    // ---- wait for the landmark blocks (one lane polls; bounded)
    {
        const int need = v.land_blocks;
        int ok = 1;
        if (lane == 0) {
            int spins = 0;
            while (__hip_atomic_load(ba_sync_words(v) + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(4);
            if (spins >= (1 << 22)) { ok = 0; __hip_atomic_fetch_add(ba_sync_words(v) + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        if (!__builtin_amdgcn_readfirstlane(ok)) return;
    }
    double pose[7], R[9];
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = ld_sc1(poses_new + 7 * p + i);
    quat_to_rot(pose, R);
    double h[21], b[6];
#pragma unroll
    for (int i = 0; i < 21; ++i) h[i] = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = 0;
    {
        for (int s = v.ps_start[p] + sp * 64 + lane; s < v.ps_start[p + 1]; s += 64 * SPLIT) {
            const int k = s;                        // observations are stored keyframe by keyframe
            if (!v.o_active[k]) continue;
            const int j = v.o_point[k];
            const double X[3] = {ld_sc1(points_new + 3 * (size_t)j), ld_sc1(points_new + 3 * (size_t)j + 1), ld_sc1(points_new + 3 * (size_t)j + 2)};
            double e[3], pc[3], rho0;
            const int D = ba_residual_vals(v.cam, v.o_u[k], v.o_v[k], v.o_ur[k], R, pose + 4, X, e, pc);
            const double w = ba_weight_vals(v.cam, v.o_w[k], D, e, robust, &rho0);
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
    double* out = partial_new + ((size_t)slot * SPLIT + sp) * PV;
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

__global__ __launch_bounds__(256) void k_ba_update(const BaView* __restrict__ views, int robust, int points_fixed)
{
    BA_VIEW(v);
    BA_VIEW_HEAD("s"(v.land_blocks), "s"(v.pose_blocks), "s"(v.ctl));
    const int land_blocks = v.land_blocks;
    if ((int)blockIdx.x >= land_blocks + v.pose_blocks || !upd_takes(v.n_points, v.n_free, v.n_poses)) return;
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle()) return;
    const int cur = fl.cur_launch;                         // stays put while the decision below flips `cur` (set by the Schur kernel of this trial)
    if ((int)blockIdx.x >= land_blocks) { upd_pose_body(v, (int)blockIdx.x - land_blocks, robust, cur); return; }
    upd_land_body(v, blockIdx.x, robust, points_fixed, cur, fl.lambda);
    if (!ba_last_block_sc1(v.ctl, land_blocks)) return;
    // ---- the landmark block that finishes last: totals in block order, then g2o's accept / reject
    __shared__ double s_tot[3];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wave == 0) {
        double a = 0;
        for (int i = lane; i < land_blocks; i += 64) a += ld_sc1(&v.part[land_blocks + i]);
        a = wave_sum(a);
        if (lane == 0) s_tot[0] = a;
    } else if (wave == 1) {
        double a = 0;
        for (int i = lane; i < land_blocks; i += 64) a += ld_sc1(&v.part[i]);
        a = wave_sum(a);
        if (lane == 0) s_tot[1] = a;
    } else if (wave == 2) {
        double a = 0;
        const double lambda = fl.lambda;
        for (int p = lane; p < v.n_poses; p += 64) {
            const int slot = v.pose_slot[p];
            if (slot < 0) continue;
            for (int q = 0; q < 6; ++q) { const double x = v.xp[6 * slot + q]; a += x * (lambda * x + v.bp[6 * slot + q]); }
        }
        a = wave_sum(a);
        if (lane == 0) s_tot[2] = a;
    }
    __syncthreads();
    if (tid == 0) {
        const double fail = v.scal[5];
        v.scal[1] = s_tot[0]; v.scal[2] = s_tot[1]; v.scal[3] = s_tot[2];
        lm_decide(v, s_tot[0], fail, s_tot[1], s_tot[2], true);
    }
}
