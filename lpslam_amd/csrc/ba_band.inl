// ba_band.inl -- block-banded keyframe windows (included by ba.hip): the landmark-group Schur complement on the FP64 matrix cores
// and the reduced system factored and solved as a BAND by one workgroup.
//
// [UPSTREAM] g2o BlockSolver_6_3::buildSystem + LinearSolverCSparse::solve (the reference builds g2o with
// solver_csparse, /root/reference/conan-packages/g2o-conan/conanfile.py:117-124): the local bundle adjuster's reduced system is
// solved SPARSE, because a landmark is seen by a run of neighbouring keyframes and the Schur complement therefore only couples
// keyframes a few places apart.  When the window handed to lpslam_hip_ba_create has that shape (every landmark's free keyframes
// lie within BD_MAXHBW slots of each other) the problem takes this path; any other window keeps the pair lists and the dense
// panel chain (k_ba_schur / k_chol_pair).  Per LM trial:
//
//   k_schur_group        one workgroup per GROUP of <= 64 landmarks whose observations fall into a window of <= 10 neighbouring
//                        keyframes.  With H_ll + lambda I = L L^T per landmark, Z = W L^-T (one 6 x 3 block per observation) is
//                        staged into LDS as a dense 64 x 3G matrix (rows = keyframe slot x 6, columns = landmark x 3; W is read
//                        ONCE) and the group's share of S,  Z Z^T = sum_j W_j (H_ll,j + lambda I)^-1 W_j^T,  is ten 16 x 16 tiles
//                        of v_mfma_f64_16x16x4 over K = 3G -- the "true dense contraction" of the normal equations -- plus four
//                        tiles for Z (L^-1 b_l), the group's share of the right-hand side.  Sums run over k ascending: fixed order.
//   k_schur_band_reduce  one wavefront per 6 x 6 block of the band: the groups that cover it are a contiguous range (groups are
//                        sorted by their first keyframe); H_pp - sum (+ lambda) in group order -> lower band of S, rhs row.
//   k_chol_band          ONE workgroup per problem: right-looking band Cholesky in 16-column strips.  Wavefront 0 factors
//                        [D; 64 riding rows] with the DPP strips of the chain (strip_factor), wavefront 1 the same D with
//                        [rhs row; I] riding -- forward substitution and L_ss^-T for free, beside it -- the 64 x 64 trailing window
//                        is updated on the matrix cores out of a ring buffer in LDS (the four tiles the next strip needs first,
//                        the other six beside the next strip's factorisation), rows enter the window from L2 one strip ahead.
//                        Backward substitution strip by strip with the stored L_ss^-T.  O(n hb^2) instead of O(n^3 / 3) and
//                        no L^-T rows: 0.7 MFLOP for a 294 x 294 window with 8-keyframe tracks instead of 8.5.
// All sums have a fixed order: results are reproducible, and identical for a problem solved alone or inside a batch.

constexpr int BD_ROWS = 64;                 // rows of a group's window: <= 10 keyframes (60 rows) in four 16-row tiles
constexpr int BD_MAXKF = 10;                // keyframes a group's window may span
constexpr int BD_MAXHBW = 9;                // block half-bandwidth taken: 6 * 9 + 5 = 59 <= 64 riding rows of a strip
constexpr int BD_GMAX = 64;                 // landmarks per group at most (K = 192)
constexpr int BD_PART = BD_ROWS * BD_ROWS + BD_ROWS;      // doubles of a group's partial: window block (lower tiles) + rhs share
constexpr int BD_REC = 8;                   // ints per group record: e0, e1 (entry range), f0 (first free slot), cnt, rows, -, -, -

__host__ __device__ inline int bd_stride(int cnt) { const int k4 = (3 * cnt + 3) & ~3; return ((k4 + 31) & ~31) + 4; }    // % 32 == 4: operand reads 2-way at most
__host__ __device__ inline size_t bd_lds_bytes(int gmax) { return ((size_t)BD_ROWS * bd_stride(gmax) + 3 * (size_t)gmax + 8) * sizeof(double); }

// ---- creation: the entry table, in group order.  Thread per ordered landmark q: its observations (CSR order = keyframe order)
//      become entries (storage slot, window row or -1 for a fixed keyframe, column | flags, landmark).
//      flags: bit 16 = first entry of its landmark (writes the landmark's rhs vector), bit 17 = duplicate (same landmark seen
//      twice by one keyframe: summed by the thread of the run's first entry).
__global__ __launch_bounds__(256) void k_bd_entries(const int* __restrict__ order, const int* __restrict__ qinfo, const int* __restrict__ bstart, int n_ord,
                                                    const int* __restrict__ pt_start, const int* __restrict__ pt_obs, const int* __restrict__ o_pose,
                                                    const int* __restrict__ pose_slot, int4* __restrict__ entries)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_ord) return;
    const int j = order[q], info = qinfo[q], f0 = info >> 8, li = info & 255;
    const int o0 = pt_start[j], o1 = pt_start[j + 1];
    int4* out = entries + bstart[q];
    int prev_pose = -1;
    for (int o = o0; o < o1; ++o) {
        const int s = pt_obs[o], p = o_pose[s], slot = pose_slot[p];
        int flags = 0;
        if (o == o0) flags |= 1 << 16;
        if (p == prev_pose) flags |= 1 << 17;
        prev_pose = p;
        out[o - o0] = make_int4(s, slot < 0 ? -1 : 6 * (slot - f0), 3 * li | flags, j);
    }
}

// L^-1 of H_ll + lambda I = L L^T (lower 3 x 3; zero rows from a non-positive pivot on: the landmark then contributes nothing)
__device__ __forceinline__ void bd_linv(const double* hl, double lambda, double* li)
{
    const double a = hl[0] + lambda, b = hl[1], c = hl[2], d = hl[3] + lambda, e = hl[4], f = hl[5] + lambda;
    double i00 = 0, i10 = 0, i11 = 0, i20 = 0, i21 = 0, i22 = 0;
    if (a > 0) {
        i00 = fast_rsqrt(a);
        const double l10 = b * i00, l20 = c * i00;
        const double t1 = d - l10 * l10;
        if (t1 > 0) {
            i11 = fast_rsqrt(t1);
            const double l21 = (e - l20 * l10) * i11;
            i10 = -l10 * i00 * i11;
            const double t2 = f - l20 * l20 - l21 * l21;
            if (t2 > 0) {
                i22 = fast_rsqrt(t2);
                i21 = -l21 * i11 * i22;
                i20 = -(l20 * i00 + l21 * i10) * i22;
            }
        }
    }
    li[0] = i00; li[1] = i10; li[2] = i11; li[3] = i20; li[4] = i21; li[5] = i22;
}

__global__ __launch_bounds__(256) void k_schur_group(const BaView* __restrict__ views)
{
    BA_VIEW_XCD(v, bx);
    BA_VIEW_HEAD("s"(v.band_hbw), "s"(v.band_groups), "s"(v.ctl), "s"(v.band_tab));
    if (v.band_hbw < 0 || bx >= v.band_groups) return;
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle()) return;
    const double lambda = fl.lambda;
    ba_lin_set(v, fl.cur);
    extern __shared__ __attribute__((aligned(16))) double bd_lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    GPTR(const int) rec = v.band_tab + (size_t)BD_REC * bx;
    const int e0 = rec[0], e1 = rec[1], cnt = rec[3], rows = rec[4];
    const int K4 = (3 * cnt + 3) & ~3, stride = bd_stride(cnt);
    double* const Z = bd_lds;                        // [64][stride]: row = 6 (slot - f0) + r, column = 3 landmark + c
    double* const U = Z + BD_ROWS * stride;          // [K4]: L^-1 b_l of the group's landmarks
    const int row_tiles = (rows + 15) >> 4;
    {
        f64x2* z2 = reinterpret_cast<f64x2*>(Z);
        const int n2 = (16 * row_tiles * stride) >> 1;
        const f64x2 zero = {0.0, 0.0};
        for (int i = tid; i < n2; i += 256) z2[i] = zero;
        if (tid < K4) U[tid] = 0.0;
    }
    __syncthreads();
    GPTR(const int4) ent = reinterpret_cast<GPTR(const int4)>(v.band_ent);
    // two entries per thread and round in flight (entry -> W row / landmark block are dependent round trips)
    for (int eb = e0 + tid; eb < e1; eb += 512) {
        int4 en[2];
        double w[2][18], hl[2][6], bl[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) en[u] = ent[min(eb + 256 * u, e1 - 1)];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const double2* Wa = reinterpret_cast<const double2*>(v.W + 18 * (size_t)en[u].x);
#pragma unroll
            for (int q = 0; q < 9; ++q) { const double2 a2 = Wa[q]; w[u][2 * q] = a2.x; w[u][2 * q + 1] = a2.y; }
#pragma unroll
            for (int q = 0; q < 6; ++q) hl[u][q] = v.Hll[6 * (size_t)en[u].w + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) bl[u][q] = v.bl[3 * (size_t)en[u].w + q];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (eb + 256 * u >= e1) continue;
            const int col = en[u].z & 0xFFFF, flags = en[u].z >> 16;
            double li[6];
            bd_linv(hl[u], lambda, li);
            if (flags & 1) {                              // first entry of the landmark: u = L^-1 b_l
                U[col] = li[0] * bl[u][0];
                U[col + 1] = li[1] * bl[u][0] + li[2] * bl[u][1];
                U[col + 2] = li[3] * bl[u][0] + li[4] * bl[u][1] + li[5] * bl[u][2];
            }
            if (en[u].y < 0 || (flags & 2)) continue;     // fixed keyframe / summed by the first entry of its run
            for (int e2 = eb + 256 * u + 1; e2 < e1; ++e2) {      // duplicates (rare): the same landmark seen again by this keyframe
                const int4 d = ent[e2];
                if (!((d.z >> 16) & 2)) break;
                const double* Wd = v.W + 18 * (size_t)d.x;
#pragma unroll
                for (int q = 0; q < 18; ++q) w[u][q] += Wd[q];
            }
            double* zr = Z + en[u].y * stride + col;
#pragma unroll
            for (int r = 0; r < 6; ++r) {                 // Z = W L^-T: column c = sum_{k <= c} W[:, k] Linv[c][k]
                const double w0 = w[u][3 * r], w1 = w[u][3 * r + 1], w2 = w[u][3 * r + 2];
                zr[r * stride] = w0 * li[0];
                zr[r * stride + 1] = w0 * li[1] + w1 * li[2];
                zr[r * stride + 2] = w0 * li[3] + w1 * li[4] + w2 * li[5];
            }
        }
    }
    __syncthreads();
    // ---- Z Z^T (lower tiles) and Z u on the matrix cores; jobs round robin over the four wavefronts
    const int lr = lane & 15, lk = lane >> 4;
    GPTR(double) P = v.band_part + (size_t)bx * BD_PART;
    const int n_sym = row_tiles * (row_tiles + 1) / 2;
    for (int job = wave; job < n_sym + row_tiles; job += 4) {
        int tr, tc;
        const bool is_rhs = job >= n_sym;
        if (is_rhs) { tr = job - n_sym; tc = 0; }
        else { tr = 0; int t = job; while (t > tr) { t -= tr + 1; ++tr; } tc = t; }
        const double* za = Z + (16 * tr + lr) * stride + lk;
        const double* zb = Z + (16 * tc + lr) * stride + lk;
        f64x4 acc = {0, 0, 0, 0};
        if (!is_rhs) {
            for (int k = 0; k < K4; k += 16) {           // four steps' operands fetched together
                double av[4], bv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int kk = min(k + 4 * i, K4 - 4); av[i] = za[kk]; bv[i] = zb[kk]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) if (k + 4 * i < K4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) P[(16 * tr + lk + 4 * q) * BD_ROWS + 16 * tc + lr] = acc[q];
        } else {
            for (int k = 0; k < K4; k += 4) {
                const double bv = lr == 0 ? U[k + lk] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(za[k], bv, acc, 0, 0, 0);
            }
            if (lr == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) P[BD_ROWS * BD_ROWS + 16 * tr + lk + 4 * q] = acc[q];
            }
        }
    }
}

// ---- the groups' shares summed into the band of S (one wavefront per 6 x 6 block, groups in order) + the rhs row ----------------
__global__ __launch_bounds__(64) void k_schur_band_reduce(const BaView* __restrict__ views, int fused)
{
    BA_VIEW_XCD(v, bx);
    BA_VIEW_HEAD("s"(v.band_hbw), "s"(v.n_free), "s"(v.ctl), "s"(v.band_tab), "s"(v.band_groups_cap));
    if (v.band_hbw < 0) return;
    const int bw = v.band_hbw + 1, nf = v.n_free;
    const int n_blk = nf * bw;
    if (bx >= n_blk + nf) return;
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle()) return;
    const double lambda = fl.lambda;
    ba_lin_set(v, fl.cur);
    const int lane = threadIdx.x;
    const int n = v.dim_pad;
    GPTR(const int) recs = v.band_tab;
    GPTR(const int) glo = v.band_tab + (size_t)BD_REC * v.band_groups_cap;
    GPTR(const int) ghi = glo + nf;
    const bool is_rhs = bx >= n_blk;
    const int i = is_rhs ? bx - n_blk : bx / bw;
    const int k = is_rhs ? i : i - (bx - i * bw);
    if (k < 0) return;
    const int g0 = glo[i], g1 = ghi[k];               // groups that may cover rows of keyframe i and columns of keyframe k (inclusive)
    const int r = lane / 6, c = lane - 6 * r;
    const bool act = is_rhs ? lane < 6 : lane < 36;
    double sum = 0;
    for (int gb = g0; gb <= g1; gb += 4) {
        double val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = min(gb + u, g1);
            const int f0 = recs[BD_REC * g + 2], rows = recs[BD_REC * g + 4];
            const bool cover = gb + u <= g1 && 6 * (i - f0) + 6 <= rows && k >= f0 && act;
            const size_t off = is_rhs ? (size_t)(BD_ROWS * BD_ROWS + 6 * (i - f0) + lane) : (size_t)((6 * (i - f0) + r) * BD_ROWS + 6 * (k - f0) + c);
            val[u] = cover ? v.band_part[(size_t)g * BD_PART + off] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) sum += val[u];
    }
    if (is_rhs) {
        if (lane < 6) {
            const int qd = 6 * lane - lane * (lane - 1) / 2;
            double bsum = 0, dsum = 0;
            for (int sp = 0; sp < SPLIT; ++sp) { const double* pr = v.partial + ((size_t)i * SPLIT + sp) * PV; bsum += pr[21 + lane]; dsum += pr[qd]; }
            const double val = bsum - sum;
            v.rhs[6 * i + lane] = val;
            if (fused) v.S[(size_t)v.dim * n + 6 * i + lane] = val;
            v.bp[6 * i + lane] = bsum; v.hppdiag[6 * i + lane] = dsum;
        }
        if (!fused && i == 0 && lane == 62) *v.chi_cur = *v.chi_loc;
        if (fused && i == 0 && lane == 63) { v.S[(size_t)v.dim * n + v.dim] = 1e200; v.scal[5] = 0.0; }
        return;
    }
    if (lane >= 36) return;
    if (i == k) {
        const int ra = min(r, c), rc = max(r, c);
        const int q = 6 * ra - ra * (ra - 1) / 2 + (rc - ra);
        double hpp = 0;
        for (int sp = 0; sp < SPLIT; ++sp) hpp += v.partial[((size_t)i * SPLIT + sp) * PV + q];
        double val = hpp - sum;
        if (fused && r == c) val += lambda;
        v.S[(size_t)(6 * i + r) * n + 6 * i + c] = val;
    } else {
        v.S[(size_t)(6 * i + r) * n + 6 * k + c] = -sum;
    }
}

// ---- band Cholesky + solve in one workgroup ------------------------------------------------------------------------------------
constexpr int BC_RING = 96;                 // rows / columns of the window ring: the 80 live ones + the 16 entering
constexpr int BC_RS = BC_RING + 1;          // row stride (odd: the riding rows of a strip read conflict-free)
constexpr int BC_MAXS = 19;                 // strips: dim <= 304
constexpr int BC_LDS_DOUBLES = BC_RING * BC_RS + 2 * 64 * 17 + BC_MAXS * 16 * 17 + 3 * 320 + 32;
constexpr int BC_LDS_BYTES = BC_LDS_DOUBLES * 8;
__host__ __device__ inline bool bc_fits(int dim) { return dim > 0 && dim <= 16 * BC_MAXS; }

__device__ __forceinline__ int bc_ring(int r) { return r % BC_RING; }

__global__ __launch_bounds__(256) void k_chol_band(const BaView* __restrict__ views)
{
    const BaView& vw = views[blockIdx.y];
    const int dim = vw.dim, n = vw.dim_pad, hbw = vw.band_hbw;
    GPTR(double) S = vw.S; GPTR(double) xp = vw.xp; GPTR(double) scal = vw.scal; GPTR(BaCtl) ctl = vw.ctl;
    asm volatile("" :: "s"(dim), "s"(n), "s"(hbw), "s"(S), "s"(xp), "s"(scal), "s"(ctl));
    if (hbw < 0 || dim == 0) return;
    if (ba_flags(ctl).idle()) return;
    extern __shared__ __attribute__((aligned(16))) double bc_lds[];
    double* const Win = bc_lds;                          // ring window, element (r, c) at [r % 96][c % 96]
    double* const LxS0 = Win + BC_RING * BC_RS;          // riding rows of the strip just factored [64][17], two buffers: wavefronts 2 / 3
    double* const Tinv = LxS0 + 2 * 64 * 17;             // still read strip s - 1's while wavefront 0 writes strip s's; L_ss^-T of every strip [s][16][17]
    double* const rhsv = Tinv + BC_MAXS * 16 * 17;       // the rhs row as it is updated / y
    double* const xv = rhsv + 320;                       // solution (zero beyond dim)
    double* const vtmp = xv + 320;                       // [16] + Ly [16]
    double* const Ly = vtmp + 16;
    __shared__ int s_fail;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int ns = (dim + 15) >> 4;
    auto load_s = [&](int r, int c) -> double {          // S(r, c), r >= c; identity beyond dim
        return (r < dim) ? S[(size_t)r * n + c] : (r == c ? 1.0 : 0.0);
    };
    // rows [r0, r1) enter the ring: their columns max(0, r - 79) .. r
    auto load_rows = [&](int r0, int r1, int t0, int nt) {
        const int cnt = (r1 - r0) * 80;
        for (int i = t0; i < cnt; i += nt) {
            const int r = r0 + i / 80, c = r - 79 + (i % 80);
            if (c >= 0) Win[bc_ring(r) * BC_RS + bc_ring(c)] = load_s(r, c);
        }
    };
    if (tid == 0) s_fail = 0;
    for (int i = tid; i < 320; i += 256) { rhsv[i] = i < dim ? S[(size_t)dim * n + i] : 0.0; xv[i] = 0.0; }
    load_rows(0, 80, tid, 256);
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        const int c0 = 16 * s;
        double* const LxS = LxS0 + (s & 1) * 64 * 17;
        const double* const LxP = LxS0 + ((s & 1) ^ 1) * 64 * 17;       // the previous strip's
        if (wave < 2) {
            // wavefront 0: [D; rows c0+16 .. c0+79]; wavefront 1: [D; rhs row; I]
            double d[16], x[16];
            const int dr = bc_ring(c0 + lr) * BC_RS;
#pragma unroll
            for (int c = 0; c < 16; ++c) d[c] = Win[dr + bc_ring(c0 + c)];
            if (wave == 0) {
                const int xr = bc_ring(c0 + 16 + lane) * BC_RS;
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = Win[xr + bc_ring(c0 + c)];
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = lane == 0 ? rhsv[c0 + c] : (lane == c + 1 ? 1.0 : 0.0);
            }
            const bool fail = strip_factor(d, x);
            if (wave == 0) {
                if (fail && lane == 0) s_fail = 1;
#pragma unroll
                for (int c = 0; c < 16; ++c) LxS[lane * 17 + c] = x[c];
                // L to memory (lower band of S, in place) for the backward substitution
                if (lane < 16 && c0 + lane < dim) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) if (c <= lane) S[(size_t)(c0 + lane) * n + c0 + c] = d[c];
                }
                if (c0 + 16 + lane < dim) {
                    f64x2* dst = reinterpret_cast<f64x2*>(S + (size_t)(c0 + 16 + lane) * n + c0);
#pragma unroll
                    for (int c = 0; c < 16; c += 2) { const f64x2 t = {x[c], x[c + 1]}; dst[c >> 1] = t; }
                }
            } else {
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) { rhsv[c0 + c] = x[c]; Ly[c] = x[c]; }
                } else if (lane <= 16) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) Tinv[(s * 16 + lane - 1) * 17 + c] = x[c];
                }
            }
        } else {
            // wavefronts 2, 3: the six tiles of the previous strip's trailing update the next strip does not need first, then the
            // 16 rows that enter the window for the strip after this one
            if (s > 0) {
                const int w0 = c0;                       // previous strip's window starts at its c0 + 16 = this c0
                for (int job = wave - 2; job < 6; job += 2) {
                    const int tr = job < 3 ? job + 1 : (job < 5 ? job - 1 : 3), tc = job < 3 ? 1 : (job < 5 ? 2 : 3);
                    f64x4 acc = {0, 0, 0, 0};
#pragma unroll
                    for (int k4 = 0; k4 < 16; k4 += 4)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(LxP[(16 * tr + lr) * 17 + k4 + lk], LxP[(16 * tc + lr) * 17 + k4 + lk], acc, 0, 0, 0);
                    const int cc = bc_ring(w0 + 16 * tc + lr);
#pragma unroll
                    for (int q = 0; q < 4; ++q) Win[bc_ring(w0 + 16 * tr + lk + 4 * q) * BC_RS + cc] -= acc[q];
                }
            }
            load_rows(c0 + 80, c0 + 96, tid - 128, 128);
        }
        __syncthreads();
        // ---- the four tiles of column 0 of this strip's window (what the next strip loads), one per wavefront; wavefront 1 also
        //      takes the rhs row along
        {
            const int w0 = c0 + 16, tr = wave;
            f64x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int k4 = 0; k4 < 16; k4 += 4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(LxS[(16 * tr + lr) * 17 + k4 + lk], LxS[lr * 17 + k4 + lk], acc, 0, 0, 0);
            const int cc = bc_ring(w0 + lr);
#pragma unroll
            for (int q = 0; q < 4; ++q) Win[bc_ring(w0 + 16 * tr + lk + 4 * q) * BC_RS + cc] -= acc[q];
            if (wave == 1) {
                double a = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) a += Ly[k] * LxS[lane * 17 + k];
                rhsv[w0 + lane] -= a;
            }
        }
        __syncthreads();
    }
    if (tid == 0 && s_fail) scal[5] = 1.0;
    // ---- backward substitution: x_s = L_ss^-T (y_s - L_below,s^T x_below), strips in reverse.  L comes back from memory (this
    //      workgroup's own stores: visible after the barrier), one strip ahead.
    {
        double lv[4];
        auto fetch = [&](int s2) {
            const int c0 = 16 * s2, r = c0 + 16 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) lv[q] = (s2 >= 0 && r < dim) ? __hip_atomic_load(&S[(size_t)r * n + c0 + 4 * wave + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // wavefront 0's stores of L have left for the L2 (the loads above bypass the L1)
        __syncthreads();
        fetch(ns - 1);
        for (int s = ns - 1; s >= 0; --s) {
            const int c0 = 16 * s;
            double p[4];
            const double xr = xv[c0 + 16 + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q) p[q] = lv[q] * xr;
            fetch(s - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) p[q] = wave_sum(p[q]);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) vtmp[4 * wave + q] = rhsv[c0 + 4 * wave + q] - p[q];
            }
            __syncthreads();
            if (tid < 16) {
                double a = 0;
#pragma unroll
                for (int c = 0; c < 16; ++c) a += Tinv[(s * 16 + tid) * 17 + c] * vtmp[c];      // row tid of L_ss^-T (zero left of the diagonal)
                xv[c0 + tid] = a;
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < dim; i += 256) xp[i] = xv[i];
}

// dynamic LDS beyond 64 KB: the attribute belongs to the (function, device) pair
inline void bd_set_attributes()
{
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || attr_set[dev].load()) return;
    (void)hipFuncSetAttribute((const void*)k_schur_group, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bd_lds_bytes(BD_GMAX));
    (void)hipFuncSetAttribute((const void*)k_chol_band, hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS_BYTES);
    attr_set[dev].store(true);
}
