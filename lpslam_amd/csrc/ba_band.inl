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
constexpr int BD_THREADS = 512;             // k_schur_group: eight wavefronts
constexpr int BD_REC = 8;                   // ints per group record: e0, e1 (entry range), f0 (first free slot), cnt, rows, -, -, -

__host__ __device__ inline int bd_stride(int cnt) { const int k4 = (3 * cnt + 3) & ~3; return ((k4 + 31) & ~31) + 4; }    // % 32 == 4: operand reads 2-way at most
__host__ __device__ inline size_t bd_lds_bytes(int gmax) { return ((size_t)BD_ROWS * bd_stride(gmax) + 3 * (size_t)gmax + 8) * sizeof(double); }

// ---- creation: the entry table, in group order: k_bs_band_entries (ba_build.inl).

// sum over the 16 lanes of a DPP row, in every lane (xor 1, xor 2, half mirror, mirror: a fixed tree)
template <int CTRL>
__device__ __forceinline__ double bc_dpp_mov(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bc_row16_sum(double v)
{
    v += bc_dpp_mov<0xB1>(v); v += bc_dpp_mov<0x4E>(v); v += bc_dpp_mov<0x141>(v); v += bc_dpp_mov<0x140>(v);
    return v;
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

// ---- the groups' shares summed into the band of S (256 threads per 6 x 6 block) + the rhs row.  The groups that can cover a block
//      are a contiguous range (groups are sorted by their first keyframe).  Thread t takes element t % 36 of the block and every
//      seventh group of the range (rhs blocks: element t % 6, every 42nd group), its loads independent of each other; the seven
//      (42) partial sums of an element are added in lane order: a fixed summation tree whatever the placement.
//      Runs as a launch of its own (k_schur_band_reduce), or -- `same_launch`, LPSLAM_HIP_BA_REDUCE_IN_SCHUR=1, measured and slower, see
//      enqueue_reduce -- as the trailing workgroups of the k_schur_group launch (two blocks per workgroup; they wait until every group
//      and every pose-side wavefront has counted itself in, and read what those wrote L2-bypassing).  `bx` >= the number of blocks:
//      an idle half.
__device__ __forceinline__ void bd_reduce_block(BaView& v, int bx, int tid, double* part, int fused, double lambda, bool same_launch, bool pose_pending)
{
    const int bw = v.band_hbw + 1, nf = v.n_free;
    const int n_blk = nf * bw;
    const int n = v.dim_pad;
    GPTR(const int) recs = v.band_tab;
    GPTR(const int) glo = v.band_tab + (size_t)BD_REC * v.band_groups_cap;
    GPTR(const int) ghi = glo + nf;
    const bool is_rhs = bx >= n_blk;
    const int i = is_rhs ? bx - n_blk : bx / bw;
    const int k = is_rhs ? i : i - (bx - i * bw);
    const bool live = bx < n_blk + nf && k >= 0;
    const int g0 = live ? glo[i] : 0, g1 = live ? ghi[k] : -1;      // groups that may cover rows of keyframe i and columns of keyframe k (inclusive)
    const int ne = is_rhs ? 6 : 36, ngl = 252 / ne;
    const int e = tid % ne, gl = tid / ne;
    const int r = e / 6, c = e - 6 * r;
    // what does not depend on the groups' results: the covering groups' records (first keyframe, rows), four groups per round
    if (same_launch) {
        // every group of this problem has stored its share, every pose-side wavefront its sums: one lane polls (bounded), the workgroup follows
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(ba_sync_words(v) + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v.band_groups && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
            while (pose_pending && __hip_atomic_load(ba_sync_words(v) + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nf * SPLIT && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
            if (spins >= (1 << 22)) __hip_atomic_fetch_add(ba_sync_words(v), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_barrier" ::: "memory");             // (both halves of the workgroup: uniform)
    }
    double sum = 0;
    if (live && tid < 252) {
        for (int gb = g0 + gl; gb <= g1; gb += 4 * ngl) {
            double val[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = min(gb + u * ngl, g1);
                const int f0 = recs[BD_REC * g + 2], rows = recs[BD_REC * g + 4];
                const bool cover = gb + u * ngl <= g1 && 6 * (i - f0) + 6 <= rows && k >= f0;
                const size_t off = is_rhs ? (size_t)(BD_ROWS * BD_ROWS + 6 * (i - f0) + e) : (size_t)((6 * (i - f0) + r) * BD_ROWS + 6 * (k - f0) + c);
                GPTR(const double) src = v.band_part + (size_t)g * BD_PART + off;
                val[u] = cover ? (same_launch ? ld_sc1(src) : *src) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sum += val[u];
        }
    }
    part[tid] = sum;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!live || tid >= ne) return;
    sum = 0;
    for (int l = 0; l < ngl; ++l) sum += part[l * ne + e];
    const bool sc = same_launch && pose_pending;            // the pose-side partials were written by workgroups of this launch
    if (is_rhs) {
        const int qd = 6 * e - e * (e - 1) / 2;
        double bsum = 0, dsum = 0;
        for (int sp = 0; sp < SPLIT; ++sp) { const double* pr = v.partial + ((size_t)i * SPLIT + sp) * PV; bsum += sc ? ld_sc1(pr + 21 + e) : pr[21 + e]; dsum += sc ? ld_sc1(pr + qd) : pr[qd]; }
        const double val = bsum - sum;
        v.rhs[6 * i + e] = val;
        if (fused) v.S[(size_t)v.dim * n + 6 * i + e] = val;
        v.bp[6 * i + e] = bsum; v.hppdiag[6 * i + e] = dsum;
        if (!fused && i == 0 && e == 0) *v.chi_cur = *v.chi_loc;
        if (fused && i == 0 && e == 1) { v.S[(size_t)v.dim * n + v.dim] = 1e200; v.scal[5] = 0.0; }
        return;
    }
    if (i == k) {
        const int ra = min(r, c), rc = max(r, c);
        const int q = 6 * ra - ra * (ra - 1) / 2 + (rc - ra);
        double hpp = 0;
        for (int sp = 0; sp < SPLIT; ++sp) { const double* pr = v.partial + ((size_t)i * SPLIT + sp) * PV + q; hpp += sc ? ld_sc1(pr) : *pr; }
        double val = hpp - sum;
        if (fused && r == c) val += lambda;
        v.S[(size_t)(6 * i + r) * n + 6 * i + c] = val;
    } else {
        v.S[(size_t)(6 * i + r) * n + 6 * k + c] = -sum;
    }
}
__global__ __launch_bounds__(256) void k_schur_band_reduce(const BaView* __restrict__ views, int fused)
{
    BA_VIEW_XCD(v, bx);
    BA_VIEW_HEAD("s"(v.band_hbw), "s"(v.n_free), "s"(v.ctl), "s"(v.band_tab), "s"(v.band_groups_cap));
    if (v.band_hbw < 0) return;
    if (bx >= v.n_free * (v.band_hbw + 2)) return;
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle()) return;
    ba_lin_set(v, fl.cur);
    __shared__ double part[256];
    bd_reduce_block(v, bx, threadIdx.x, part, fused, fl.lambda, false, false);
}

// The launch's workgroups, in dispatch order: [0, n_poses) the pose side of an accepted state's linearisation (ba_update.inl; idle
// otherwise), [n_poses, + band_groups) the landmark groups, then -- reduce_here -- ceil(n_free (hbw + 2) / 2) reduction workgroups, two
// band blocks each, which wait for the first two kinds.  Waiting workgroups follow the ones they wait for in dispatch order, and those
// wait for nobody: the launch drains whatever else occupies the chip.
template <int WAVES_PER_SIMD>         // 1: no register limit (a single window); 4: 128 registers, two workgroups per compute unit (a batch)
__global__ __launch_bounds__(BD_THREADS, WAVES_PER_SIMD) void k_schur_group(const BaView* __restrict__ views, int fused, int robust, int reduce_here)
{
    static_assert(BD_THREADS == 64 * SPLIT, "a leading workgroup is one keyframe: its SPLIT slices are the workgroup's wavefronts");
    BA_VIEW_XCD(v, bx0);
    BA_VIEW_HEAD("s"(v.band_hbw), "s"(v.band_groups), "s"(v.ctl), "s"(v.band_tab), "s"(v.n_poses), "s"(v.n_free));
    const int lead = v.n_poses;
    const int n_red = reduce_here ? (v.n_free * (v.band_hbw + 2) + 1) / 2 : 0;
    if (v.band_hbw < 0 || bx0 >= lead + v.band_groups + n_red) return;
    const BaFlags fl = ba_flags(v.ctl);
    if (fl.idle()) return;
    const bool pose_pending = fused && ba_sync_words(v)[3] != 0;
    if (bx0 < lead) { if (pose_pending) ba_pose_side_wave<(WAVES_PER_SIMD > 1 ? 1 : 2)>(v, bx0, (int)(threadIdx.x >> 6), robust, fl.cur, reduce_here != 0); return; }
    const int bx = bx0 - lead;
    const double lambda = fl.lambda;
    ba_lin_set(v, fl.cur);
    extern __shared__ __attribute__((aligned(16))) double bd_lds[];
    if (bx >= v.band_groups) {
        bd_reduce_block(v, 2 * (bx - v.band_groups) + (int)(threadIdx.x >> 8), threadIdx.x & 255, bd_lds + 256 * (threadIdx.x >> 8), fused, lambda, true, pose_pending);
        return;
    }
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#ifdef LPSLAM_BC_STAMPS
#define BD_STAMP(k) do { if (bx == 0 && tid == 0) v.S[(size_t)(v.dim + 2) * v.dim_pad + 8 * 22 + (k)] = (double)wall_clock64(); } while (0)
#else
#define BD_STAMP(k) do {} while (0)
#endif
    BD_STAMP(0);
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
        for (int i = tid; i < n2; i += BD_THREADS) z2[i] = zero;
        if (tid < K4) U[tid] = 0.0;
    }
    __syncthreads();
    BD_STAMP(1);
    GPTR(const int4) ent = reinterpret_cast<GPTR(const int4)>(v.band_ent);
    // one entry per thread and round (512 threads: a group of 32 landmarks has ~290): entry -> W row / landmark block are dependent round
    // trips; a second entry in flight per thread cost 72 registers and the second workgroup of a compute unit
    for (int eb = e0 + tid; eb < e1; eb += BD_THREADS) {
        const int4 en = ent[eb];
        double w[18], hl[6], bl[3];
        {
            const double2* Wa = reinterpret_cast<const double2*>(v.W + 18 * (size_t)en.x);
#pragma unroll
            for (int q = 0; q < 9; ++q) { const double2 a2 = Wa[q]; w[2 * q] = a2.x; w[2 * q + 1] = a2.y; }
#pragma unroll
            for (int q = 0; q < 6; ++q) hl[q] = v.Hll[6 * (size_t)en.w + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) bl[q] = v.bl[3 * (size_t)en.w + q];
        }
        BD_STAMP(2);
        const int col = en.z & 0xFFFF, flags = en.z >> 16;
        double li[6];
        bd_linv(hl, lambda, li);
        if (flags & 1) {                              // first entry of the landmark: u = L^-1 b_l
            U[col] = li[0] * bl[0];
            U[col + 1] = li[1] * bl[0] + li[2] * bl[1];
            U[col + 2] = li[3] * bl[0] + li[4] * bl[1] + li[5] * bl[2];
        }
        if (en.y < 0 || (flags & 2)) continue;        // fixed keyframe / summed by the first entry of its run
        for (int e2 = eb + 1; (flags & 4) && e2 < e1; ++e2) {      // duplicates (rare): the same landmark seen again by this keyframe
            const int4 d = ent[e2];
            if (!((d.z >> 16) & 2)) break;
            const double* Wd = v.W + 18 * (size_t)d.x;
#pragma unroll
            for (int q = 0; q < 18; ++q) w[q] += Wd[q];
        }
        double* zr = Z + en.y * stride + col;
#pragma unroll
        for (int r = 0; r < 6; ++r) {                 // Z = W L^-T: column c = sum_{k <= c} W[:, k] Linv[c][k]
            const double w0 = w[3 * r], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
            zr[r * stride] = w0 * li[0];
            zr[r * stride + 1] = w0 * li[1] + w1 * li[2];
            zr[r * stride + 2] = w0 * li[3] + w1 * li[4] + w2 * li[5];
        }
    }
    __syncthreads();
    BD_STAMP(3);
    GPTR(double) P = v.band_part + (size_t)bx * BD_PART;
    // ---- the group's share of the rhs, Z u: eight lanes per row, each every eighth column, summed over the eight in a fixed tree
    {
        const int row = tid >> 3, part = tid & 7;
        double a = 0;
        if (row < 16 * row_tiles) {
            const double* zr = Z + row * stride;
            for (int kk = part; kk < K4; kk += 8) a += zr[kk] * U[kk];
        }
        a += bc_dpp_mov<0xB1>(a); a += bc_dpp_mov<0x4E>(a); a += bc_dpp_mov<0x141>(a);
        if (part == 0 && row < 16 * row_tiles) { if (reduce_here) st_sc1(P + BD_ROWS * BD_ROWS + row, a); else P[BD_ROWS * BD_ROWS + row] = a; }
    }
    // ---- Z Z^T (lower tiles) on the matrix cores: tile t -> wavefront t % 8; the operands of the next four steps are fetched while
    //      the matrix cores work on the current four
    const int lr = lane & 15, lk = lane >> 4;
    const int n_sym = row_tiles * (row_tiles + 1) / 2;
    for (int job = wave; job < n_sym; job += BD_THREADS / 64) {
        int tr = 0, tc;
        { int t = job; while (t > tr) { t -= tr + 1; ++tr; } tc = t; }
        const double* za = Z + (16 * tr + lr) * stride + lk;
        const double* zb = Z + (16 * tc + lr) * stride + lk;
        f64x4 acc = {0, 0, 0, 0};
        double av[4], bv[4], an[4], bn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int kk = min(4 * i, K4 - 4); av[i] = za[kk]; bv[i] = zb[kk]; }
        for (int k = 0; k < K4; k += 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int kk = min(k + 16 + 4 * i, K4 - 4); an[i] = za[kk]; bn[i] = zb[kk]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) if (k + 4 * i < K4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = an[i]; bv[i] = bn[i]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { GPTR(double) dst = P + (16 * tr + lk + 4 * q) * BD_ROWS + 16 * tc + lr; if (reduce_here) st_sc1(dst, acc[q]); else *dst = acc[q]; }
    }
    BD_STAMP(4);
    if (reduce_here) {
        // the share has left this compute unit (write-through, drained by every wavefront) before the group counts itself in
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ba_sync_words(v) + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- band Cholesky + solve in one workgroup ------------------------------------------------------------------------------------
constexpr int BC_RING = 96;                 // rows / columns of the window ring: the 80 live ones + the 16 entering
constexpr int BC_RS = BC_RING + 2;          // row stride: rows 16-byte aligned (the strip's rows are read 16 bytes at a time)
constexpr int BC_MAXS = 19;                 // strips: dim <= 304
constexpr int BC_TWIST_MIN = 12;            // strips from which on the factorisation runs from both ends (k_chol_band)
constexpr int BC_THREADS = 512;             // 8 wavefronts: 0 / 1 factor, 2 / 3 bring rows in, 4 .. 7 form the backward operands; all eight update tiles
constexpr int BC_LDS_DOUBLES = BC_RING * BC_RS + 2 * 64 * 17 + BC_MAXS * 16 * 17 + 2 * 384;
constexpr int BC_LDS_BYTES = BC_LDS_DOUBLES * 8;
__host__ __device__ inline bool bc_fits(int dim) { return dim > 0 && dim <= 16 * BC_MAXS; }

// ring index of base + k for base, k < 96 (base = first live column % 96, kept per strip: no division in the inner loops)
__device__ __forceinline__ int bc_ring(int base, int k) { const int v = base + k; return v >= BC_RING ? v - BC_RING : v; }

// What the backward substitution needs of strip s2, formed off the critical path (by `nw` wavefronts, this one is number w of them):
//   M = L_ss^-T [L_below,s; y_s]^T (16 x 64, matrix cores; the rhs row is row 63 of the strip's riding rows) -> memory
//   [s2][column % 16][row][column / 16] (the order the backward pass reads it in: lane = 4 row + column / 16), stored write-through.  Then x_s = M (-x_below; 1): one 16 x 64 matrix-vector product per strip on the chain
//   instead of a product with L and one with L_ss^-T.
__device__ __forceinline__ void bc_back_operands(int s2, int w, int nw, int lane, const double* Lx, const double* Tinv, GPTR(double) Mg)
{
    const int lr = lane & 15, lk = lane >> 4;
    const double* T = Tinv + s2 * 16 * 17;
    for (int j = w; j < 4; j += nw) {
        f64x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int k4 = 0; k4 < 16; k4 += 4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T[lr * 17 + k4 + lk], Lx[(16 * j + lr) * 17 + k4 + lk], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) st_sc1(Mg + (size_t)s2 * 1024 + lr * 64 + 4 * (lk + 4 * q) + j, acc[q]);     // element (row, col = 16 j + lr) at [lr][row][j]: the backward pass reads [jj][lane]
    }
}

__global__ __launch_bounds__(BC_THREADS) void k_chol_band(const BaView* __restrict__ views)
{
    const BaView& vw = views[blockIdx.y];
    const int dim = vw.dim, n = vw.dim_pad, hbw = vw.band_hbw;
    GPTR(double) S = vw.S; GPTR(double) xp = vw.xp; GPTR(double) scal = vw.scal; GPTR(BaCtl) ctl = vw.ctl;
    GPTR(double) Dg = vw.band_part + (size_t)vw.band_groups_cap * BD_PART;      // exchange block of the two chains (one more partial slot)
    GPTR(int) flag = vw.blk_ticket;                      // the pair lists' tickets are idle on the band path: [0] = "the helper's block is there"
    asm volatile("" :: "s"(dim), "s"(n), "s"(hbw), "s"(S), "s"(xp), "s"(scal), "s"(ctl), "s"(vw.Minv), "s"(Dg), "s"(flag));
    if (hbw < 0 || dim == 0) return;
    // TWISTED factorisation (systems of >= BC_TWIST_MIN strips): workgroup 0 of the problem -- dispatched before workgroup 1 -- eliminates
    // the LAST sB strips of the system, bottom-up, which is the same top-down algorithm on the flipped matrix F(r, c) = S(n-1-r, n-1-c);
    // workgroup 1 eliminates from the top.  A band of half-width <= 64 decouples the two ends; where they meet the helper's Schur
    // complement update on the 64 x 64 block in front of its end (and on the rhs there) is handed over through memory, the top-down
    // chain adds it to its window and factors the rest.  19 strips become a chain of 12.  The helper waits for nobody; the main chain
    // waits for a workgroup that was dispatched before it: no deadlock whatever the placement.
    const int team = blockIdx.x;
    const int ns_all = (dim + 15) >> 4;
    const int sB = ns_all >= BC_TWIST_MIN ? (ns_all - 5) / 2 : 0;
    if (team == 0 && sB == 0) return;
    if (ba_flags(ctl).idle()) return;
    const bool flip = team == 0;
    const int dimA = dim - 16 * sB;                      // unknowns the top-down chain factors (what lies beyond is the helper's: identity to it)
    const int nloc = flip ? dim : dimA;
    const int sM = sB ? (dimA - 64) / 16 : -1;           // the strip in front of which the top-down chain takes the helper's update in
    extern __shared__ __attribute__((aligned(16))) double bc_lds[];
    double* const Win = bc_lds;                          // ring window, element (r, c) at [r % 96][c % 96]
    double* const LxS0 = Win + BC_RING * BC_RS;          // riding rows of the strip just factored [64][17], two buffers: wavefronts 2 / 3
    double* const Tinv = LxS0 + 2 * 64 * 17;             // still read strip s - 1's while wavefront 0 writes strip s's; L_ss^-T of every strip [s][16][17]
    double* const rhsv = Tinv + BC_MAXS * 16 * 17;       // the rhs row as it is updated / y
    double* const xv = rhsv + 384;                       // solution (zero beyond the system; 384 long: the backward pass reads 64 entries behind the last strip)
    __shared__ int s_fail;
    // the barrier of the strip loop: LDS traffic drained, nothing else -- __syncthreads() also waits for every global load and store
    // in flight (the rows on their way in, the write-through stores of M, the prefetched M blocks of the backward pass)
#define BC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int ns = flip ? sB : (dimA + 15) >> 4;         // strips this workgroup factors
    // S(r, c), r >= c, of the band; identity beyond dim.  All loads of a batch are issued before the first store to LDS: written as
    // load -> store per element every element is a round trip of its own (ten per strip: 7 us of a 7 us strip, measured).
    auto load_s = [&](int r, int c) -> double {
        if (!(r < nloc && c >= 0)) return r == c ? 1.0 : 0.0;
        return flip ? S[(size_t)(dim - 1 - c) * n + (dim - 1 - r)] : S[(size_t)r * n + c];       // flipped lower (r, c) = upper (n-1-r, n-1-c) = lower (n-1-c, n-1-r)
    };
    auto load_rhs = [&](int i) -> double { return i < nloc ? S[(size_t)dim * n + (flip ? dim - 1 - i : i)] : 0.0; };
    GPTR(double) Mg = vw.Minv + (flip ? (size_t)BC_MAXS * 1024 : 0);      // M_s of every strip (the dense path's L^-T buffer); the helper's behind the main chain's
#ifdef LPSLAM_BC_STAMPS
    if (tid == 0) S[(size_t)(dim + 2) * n + 8 * (team == 1 ? 20 : 21) + 4] = (double)wall_clock64();
#endif
    if (tid == 0) s_fail = 0;
    for (int i = tid; i < 384; i += BC_THREADS) { rhsv[i] = load_rhs(i); xv[i] = 0.0; }
    // rows 0 .. 79, columns 0 .. r (lower part; the upper part of the first window is never read): 80 columns per row, 13 per thread
    {
        double val[13];
#pragma unroll
        // (neighbouring threads read neighbouring addresses: along a row of S, which for the flipped matrix is along a COLUMN of F)
        for (int q = 0; q < 13; ++q) { const int i = tid + BC_THREADS * q, a = i / 80, b2 = i - 80 * a, r = flip ? b2 : a, c = flip ? a : b2; val[q] = (i < 6400 && c <= r) ? load_s(r, c) : 0.0; }
#pragma unroll
        for (int q = 0; q < 13; ++q) { const int i = tid + BC_THREADS * q, a = i / 80, b2 = i - 80 * a, r = flip ? b2 : a, c = flip ? a : b2; if (i < 6400) Win[r * BC_RS + c] = val[q]; }
    }
    __syncthreads();
    int base = 0;                                        // c0 % 96
#ifdef LPSLAM_BC_STAMPS
    // development: wall-clock stamps (100 MHz) of wavefront 0 per strip into the padding rows of S (read back by tools/dev_band_stamps.py)
    GPTR(double) stamp = S + (size_t)(dim + 2) * n;
#define BC_STAMP(slot) do { if (tid == 0 && team == 1) stamp[8 * s + (slot)] = (double)wall_clock64(); } while (0)
#define BC_STAMP_W(w, slot) do { if (tid == 64 * (w) && team == 1) stamp[8 * s + (slot)] = (double)wall_clock64(); } while (0)
#define BC_STAMPX(idx) do { if (tid == 0) stamp[8 * (team == 1 ? 20 : 21) + (idx)] = (double)wall_clock64(); } while (0)
#else
#define BC_STAMPX(idx) do {} while (0)
#define BC_STAMP(slot) do {} while (0)
#define BC_STAMP_W(w, slot) do {} while (0)
#endif
    // tile (tr, tc) of the window that starts at ring offset `woff` / column `wcol`: Win -= Lx[16 tr ..] Lx[16 tc ..]^T.  Row 63 of Lx is
    // the rhs row: its products go to rhsv, and as a COLUMN it is nobody's (the true row c0 + 79 is zero there)
    auto tile_update = [&](const double* Lx, int tr, int tc, int woff, int wcol) __attribute__((always_inline)) {
        // the tile's old values travel with the operands (one LDS round trip) and seed the accumulators: acc = Win - Lx Lx^T
        const int cc = bc_ring(base, woff + 16 * tc + lr);
        const bool col_ok = !(tc == 3 && lr == 15);
        const int col = wcol + 16 * tc + lr;
        double* dst[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool rhs_row = tr == 3 && q == 3 && lk == 3;
            dst[q] = rhs_row ? rhsv + min(col, 383) : Win + bc_ring(base, woff + 16 * tr + lk + 4 * q) * BC_RS + cc;
        }
        double av[4], bv[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) { av[k4] = Lx[(16 * tr + lr) * 17 + 4 * k4 + lk]; bv[k4] = Lx[(16 * tc + lr) * 17 + 4 * k4 + lk]; }
        f64x4 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = *dst[q];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[k4], bv[k4], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool rhs_row = tr == 3 && q == 3 && lk == 3;
            if (col_ok && (!rhs_row || col < 384)) *dst[q] = acc[q];
        }
    };
    BC_STAMPX(0);
    for (int s = 0; s < ns; ++s) {
        const int c0 = 16 * s;
        if (s == sM && !flip) {
            // The helper's window over the 64 x 64 block [dimA - 64, dimA) and its rhs there: our window takes (helper's value - untouched
            // entry) on top of its own.  The untouched entries are fetched before the wait, the helper's after it (one lane polls).
            double sv[8], dv[8];                                 // eight elements per thread, every load of a batch in flight together
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = tid + BC_THREADS * q, i = idx >> 6, j = idx & 63;      // helper's window coordinates, lower part i >= j
                sv[q] = j <= i ? S[(size_t)(dimA - 1 - j) * n + (dimA - 1 - i)] : 0.0;   // our coordinates: (pj, pi), pi <= pj
            }
            const double sr = tid < 64 ? S[(size_t)dim * n + (dimA - 1 - tid)] : 0.0;
            // (bounded: should the helper never report -- it always does -- the factorisation is flagged as failed instead of the grid hanging)
            if (tid == 0) { int spins = 0; while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2); if (spins >= (1 << 22)) { s_fail = 1; scal[5] = 1.0; __hip_atomic_fetch_add(ba_sync_words(vw), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int idx = tid + BC_THREADS * q; dv[q] = (idx & 63) <= (idx >> 6) ? ld_sc1(Dg + idx) : 0.0; }
            const double dr = tid < 64 ? ld_sc1(Dg + 64 * 64 + tid) : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = tid + BC_THREADS * q, i = idx >> 6, j = idx & 63;
                if (j > i) continue;
                const int pi = dimA - 1 - i, pj = dimA - 1 - j;
                Win[bc_ring(base, pj - c0) * BC_RS + bc_ring(base, pi - c0)] += dv[q] - sv[q];
            }
            if (tid < 64) rhsv[dimA - 1 - tid] += dr - sr;
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch
        }
        BC_STAMP(0);
        double* const LxS = LxS0 + (s & 1) * 64 * 17;
        const double* const LxP = LxS0 + ((s & 1) ^ 1) * 64 * 17;       // the previous strip's
        if (wave < 2) {
            // wavefront 0: [D; rows c0+16 .. c0+78; the rhs row in lane 63 (a band of half-width <= 59 never reaches row c0+79)];
            // wavefront 1: [D; I] -> L_ss^-T, needed only by the backward operands
            double d[16], x[16];
            const f64x2* dp = reinterpret_cast<const f64x2*>(Win + bc_ring(base, lr) * BC_RS + base);
#pragma unroll
            for (int c = 0; c < 8; ++c) { const f64x2 t2 = dp[c]; d[2 * c] = t2.x; d[2 * c + 1] = t2.y; }
            if (wave == 0) {
                const f64x2* xp2 = reinterpret_cast<const f64x2*>(lane < 63 ? Win + bc_ring(base, 16 + lane) * BC_RS + base : rhsv + c0);
#pragma unroll
                for (int c = 0; c < 8; ++c) { const f64x2 t2 = xp2[c]; x[2 * c] = t2.x; x[2 * c + 1] = t2.y; }
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = lane == c ? 1.0 : 0.0;
            }
            BC_STAMP(1);
            const bool fail = strip_factor(d, x);
            BC_STAMP(2);
            if (wave == 0) {
                if (fail && lane == 0) s_fail = 1;
#pragma unroll
                for (int c = 0; c < 16; ++c) LxS[lane * 17 + c] = x[c];       // row 63: y_s, which rides through the tile updates into rhsv
            } else if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) Tinv[(s * 16 + lane) * 17 + c] = x[c];
            }
        } else if (wave < 4) {
            // wavefronts 2, 3: the 16 rows that enter the window for the strip after this one (rows c0+80 .. c0+95, columns r-79 .. r:
            // 1280 values, ten per thread, in flight while ...) the last two tiles of the previous strip's trailing update are applied
            const int t = tid - 128, rr = flip ? (t & 15) : (t >> 3), seg = flip ? (t >> 4) : (t & 7);      // flipped: the row index runs along S's rows
            const int r = c0 + 80 + rr;
            double val[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) val[q] = load_s(r, r - 79 + 10 * seg + q);
            if (s > 0) tile_update(LxP, 3, wave, 0, c0);      // tiles (3, 2) and (3, 3) of the previous strip's window, which starts at this strip's c0
            BC_STAMP_W(2, 6);
            const int rrow = bc_ring(base, 80 + rr) * BC_RS;
#pragma unroll
            for (int q = 0; q < 10; ++q) Win[rrow + bc_ring(base, 1 + rr + 10 * seg + q)] = val[q];
            BC_STAMP_W(2, 7);
        } else if (s > 0) {
            // wavefronts 4 .. 7: what the backward substitution needs of the previous strip
            bc_back_operands(s - 1, wave - 4, 4, lane, LxP, Tinv, Mg);
        }
        BC_STAMP(3);
        BC_BARRIER();
        BC_STAMP(4);
        // ---- this strip's trailing update: the four tiles of column 0 of its window (what the next strip loads) on wavefronts 0 .. 3 --
        //      wavefront 1 also takes the rhs row along --, tiles (1,1) (2,1) (3,1) (2,2) on wavefronts 4 .. 7; (3,2) and (3,3) follow
        //      beside the next strip's factorisation
        tile_update(LxS, wave < 4 ? wave : (wave < 7 ? wave - 3 : 2), wave < 4 ? 0 : (wave < 7 ? 1 : 2), 16, c0 + 16);
        BC_STAMP(5);
        BC_BARRIER();
        base = bc_ring(base, 16);
    }
    BC_STAMPX(1);
    if (tid == 0 && s_fail) scal[5] = 1.0;
    if (flip) {
        // ---- the helper: the last strip's remaining two tiles, then its window over the 64 rows / columns in front of its end goes to the
        //      main chain (which subtracts the untouched entries itself); the last strip's backward operands follow the hand-over
        const double* LxL = LxS0 + ((ns - 1) & 1) * 64 * 17;
        if (wave == 2 || wave == 3) tile_update(LxL, 3, wave, 0, 16 * ns);
        BC_BARRIER();
        const int u0 = 16 * ns;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = tid + BC_THREADS * q, i = idx >> 6, j = idx & 63;
            if (j <= i) st_sc1(Dg + idx, Win[bc_ring(base, i) * BC_RS + bc_ring(base, j)]);
        }
        if (tid < 64) st_sc1(Dg + 64 * 64 + tid, rhsv[u0 + tid]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wavefront's write-through stores (M blocks so far, the block above) have left
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave >= 4) bc_back_operands(ns - 1, wave - 4, 4, lane, LxL, Tinv, Mg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // "all M blocks are there": awaited in front of the backward pass over the helper's strips
        BC_STAMPX(5);
        return;
    }
    // ---- backward substitution: x_s = M_s (-x_below; 1), strips in reverse.  The strips go to the wavefronts in RUNS of three (the last
    //      three to wavefront 0, ...): inside a run x_s passes through LDS within one wavefront (no barrier), a run hands over to the next
    //      with one LDS-only barrier.  Every wavefront holds the M blocks of its run in registers; they are requested before the last
    //      strip's own block is even formed (only wavefront 0 needs that one: a second request after the stores have drained), so no
    //      memory round trip sits on the chain.  Lane = (row i of the strip, quarter q of its 64 columns): sixteen products in four
    //      independent sums, the quarters added by a quad butterfly -- a fixed tree.  Then the helper's strips the same way, in its
    //      order (flipped coordinates u = n - 1 - p), seeded with the 64 unknowns in front of them.
    {
        const int bi = lane >> 2, bq = lane & 3;
        GPTR(double) MgB = vw.Minv + (size_t)BC_MAXS * 1024;
        double* const xB = rhsv;                             // the helper's unknowns, flipped order (rhsv is free after the forward pass)
        if (sB) {
            if (tid == 0) { int spins = 0; while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2); if (spins >= (1 << 22)) { scal[5] = 1.0; __hip_atomic_fetch_add(ba_sync_words(vw), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // wavefronts 0 .. 4 hold the main chain's runs (<= 13 strips when the helper exists, <= 11 otherwise), 5 .. 7 the helper's (<= 7)
        const bool hold_b = wave >= 5;
        double mv[3][16];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int s2 = hold_b ? sB - 1 - 3 * (wave - 5) - k : ns - 1 - 3 * wave - k;
            const bool there = s2 >= 0 && (hold_b || s2 < ns - 1);
            GPTR(double) src = (hold_b ? MgB : Mg) + (size_t)max(s2, 0) * 1024 + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j) mv[k][j] = there ? ld_sc1(src + j * 64) : 0.0;                   // 64 lanes, 512 contiguous bytes per load
        }
        // the last strip's own block goes through LDS (the window is free now), in the order the backward pass reads it: no trip to memory
        // and back in front of the chain
        if (wave < 4) {
            const double* T = Tinv + (ns - 1) * 16 * 17;
            const double* Lx = LxS0 + ((ns - 1) & 1) * 64 * 17;
            f64x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int k4 = 0; k4 < 16; k4 += 4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T[lr * 17 + k4 + lk], Lx[(16 * wave + lr) * 17 + k4 + lk], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) Win[lr * 64 + 4 * (lk + 4 * q) + wave] = acc[q];      // element (row, col = 16 wave + lr) at [lr][row][wave]
        }
        BC_BARRIER();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) mv[0][j] = Win[j * 64 + lane];
        }
        BC_STAMPX(2);
        // one strip: x[c0 .. c0+15] from the 64 entries behind it (vector v, zero beyond the system: it is 384 long)
        auto strip_x = [&](double* v, const double (&m)[16], int c0) __attribute__((always_inline)) {
            const f64x2* xp2 = reinterpret_cast<const f64x2*>(v + c0 + 16 + 16 * bq);
            double p4[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f64x2 x2 = xp2[j];
                const double x1 = (bq == 3 && j == 7) ? -1.0 : x2.y;          // column 63 of M is L_ss^-T y_s (the rhs row rode along as row 63)
                p4[(2 * j) & 3] = fma(m[2 * j], x2.x, p4[(2 * j) & 3]);
                p4[(2 * j + 1) & 3] = fma(m[2 * j + 1], x1, p4[(2 * j + 1) & 3]);
            }
            double p = (p4[0] + p4[1]) + (p4[2] + p4[3]);
            p += bc_dpp_mov<0xB1>(p); p += bc_dpp_mov<0x4E>(p);
            if (bq == 0) v[c0 + bi] = -p;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the write before this wavefront's reads for its next strip
        };
        for (int w = 0; w < 5; ++w) {
            if (ns - 1 - 3 * w < 0) break;
            if (wave == w) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { const int s = ns - 1 - 3 * w - k; if (s >= 0) strip_x(xv, mv[k], 16 * s); }
            }
            BC_BARRIER();
        }
        if (sB) {
            for (int u = tid; u < 384; u += BC_THREADS) xB[u] = (u >= 16 * sB && u < dim) ? xv[dim - 1 - u] : 0.0;
            BC_BARRIER();
            for (int w = 0; w < 3; ++w) {
                if (sB - 1 - 3 * w < 0) break;
                if (wave == 5 + w) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const int sw = sB - 1 - 3 * w - k; if (sw >= 0) strip_x(xB, mv[k], 16 * sw); }
                }
                BC_BARRIER();
            }
            for (int u = tid; u < 16 * sB && u < dim; u += BC_THREADS) xp[dim - 1 - u] = xB[u];
        }
    }
    BC_STAMPX(3);
    for (int i = tid; i < dimA; i += BC_THREADS) xp[i] = xv[i];
}

// dynamic LDS beyond 64 KB: the attribute belongs to the (function, device) pair.  A refusal is reported HERE, with its reason -- left
// unchecked it would surface as a failed launch of the band kernels, far from its cause (lpslam_hip_ba_create calls this for every
// window that takes the band path and fails with the message).
inline hipError_t bd_set_attributes()
{
    static std::atomic<int> attr_state[64];             // 0: not tried, 1: set, 2: refused
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const int st = attr_state[dev].load();
    if (st == 1) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)k_schur_group<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bd_lds_bytes(BD_GMAX));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_schur_group<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bd_lds_bytes(BD_GMAX));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_chol_band, hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS_BYTES);
    attr_state[dev].store(e == hipSuccess ? 1 : 2);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("band path: %zu / %d bytes of dynamic LDS for k_schur_group / k_chol_band were refused (%s)", bd_lds_bytes(BD_GMAX), BC_LDS_BYTES, hipGetErrorString(e)); }
    return e;
}
