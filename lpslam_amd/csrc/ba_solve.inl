// ba_solve.inl -- the reduced system of a keyframe window factored and solved by ONE workgroup (included by ba.hip).
//
// [UPSTREAM] g2o LinearSolver::solve on the Schur complement.  The panel-pair chain (k_chol_pair) spends
// its time on kernel boundaries and on panels that every workgroup factors again; for the systems of a local window
// (dim + 1 <= 304: 50 free keyframes) the whole lower triangle fits on ONE compute unit -- 153 16x16 tiles in the accumulator
// registers of 7 wavefronts (22 tiles = 176 registers each; 8 wavefronts per workgroup leave every one 256 registers), the two
// panels in flight in LDS (144 KB) -- so the factorisation needs no
// inter-workgroup hand-over at all, a batch of windows is one workgroup per window, and the rest of the chip stays free for
// the front end.  Right-looking, 32-column panels (two 16-row "tile columns"), rows handled as 16-row strips:
//   per panel q:  1. the wavefronts that own tiles of panel q's two tile columns apply the previous panel to them (f64 matrix
//                    cores, operands from LDS) and put them into the other LDS buffer;
//                 2. wavefront 7 factors the 32x32 diagonal block in registers ([D; I] -> L, W = L^-T, as k_chol_pair's panel
//                    factorisation) WHILE the other 7 apply the previous panel to the rest of their tiles (lookahead);
//                 3. the strips below the diagonal block become L = A W (matrix cores) in place, and go to memory.
// The right-hand side rides along as row `dim` (forward substitution for free); the backward substitution walks the panels
// in reverse with W_q and the rows of L (read back from the L2, prefetched one panel ahead).
// All sums have a fixed order: results are reproducible and identical for a problem solved alone or inside a batch.

constexpr int CW_THREADS = 512;
constexpr int CW_UW = 7;                        // wavefronts that own tiles; wavefront 7 factors the diagonal blocks
constexpr int CW_MAXT = 19;                     // 16-row tile rows held: dim + 1 <= 304
constexpr int CW_SLOTS = 22;                    // tiles per owning wavefront: ceil(153 / 7)
constexpr int CW_STRIP = 16 * 32;               // doubles of one strip (16 rows x 32 columns of a panel)
constexpr int CW_BUF0 = CW_MAXT * CW_STRIP;     // even panels (panel 0 holds all 19 strips)
constexpr int CW_BUF1 = (CW_MAXT - 2) * CW_STRIP;
constexpr int CW_W = 32 * 32;
constexpr int CW_VEC = 320;
constexpr int CW_LDS_DOUBLES = CW_BUF0 + CW_BUF1 + CW_W + 2 * CW_VEC;      // strips, W / factor scratch, t and x vectors
constexpr int CW_LDS_BYTES = CW_LDS_DOUBLES * 8 + 1024;                     // + tile table and flags

__host__ __device__ inline bool cw_fits(int dim) { return dim > 0 && dim + 1 <= 16 * CW_MAXT; }

// element (r, c) of a strip / of W: rows of 32 doubles, columns XOR-swizzled by the row so that the matrix-core operand reads
// (16 rows x 2 neighbouring columns per half wavefront, ds_read_b64) hit 32 different 8-byte banks
__device__ __forceinline__ int cw_swz(int r, int c) { return r * 32 + (c ^ ((2 * r) & 31)); }

// acc -= A_strip (16 x 32) * B_strip^T (32 x 16): eight v_mfma_f64_16x16x4; the operands of four steps are fetched together
// (sixteen registers: the 22 accumulator tiles leave no room for all eight)
__device__ __forceinline__ f64x4 cw_tile_update(const double* sa, const double* sb, int lr, int lk, f64x4 acc)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        double av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { av[i] = sa[cw_swz(lr, 16 * h + 4 * i + lk)]; bv[i] = sb[cw_swz(lr, 16 * h + 4 * i + lk)]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[i], bv[i], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// one item of the triangular solve below a diagonal block: half h of strip `st` (raw A, 16 x 32) times W (upper triangular) ->
// 16 x 16 tile of L into strip `dst` and to memory.  Half 0 (columns 0-15) needs the first 16 terms only.
template <int H>
__device__ __forceinline__ void cw_trsm_item(const double* st, double* dst, const double* Wb, GPTR(double) Srow, int n, int lr, int lk)
{
    f64x4 o = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < (H ? 2 : 1); ++g) {              // four steps' operands at a time (register budget, as cw_tile_update)
        double av[4], wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { av[i] = st[cw_swz(lr, 16 * g + 4 * i + lk)]; wv[i] = Wb[cw_swz(16 * g + 4 * i + lk, 16 * H + lr)]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], wv[i], o, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int r = lk + 4 * q4;
        dst[cw_swz(r, 16 * H + lr)] = o[q4];
        Srow[(size_t)r * n + 16 * H + lr] = o[q4];
    }
}
// the items of panel q: strips below the block (ns of them) x two halves; the 32-term halves come first so that every
// wavefront gets a mix.  X: raw strips (tile row ti at (ti - 2q)), Y: L strips (tile row ti at (ti - 2q - 2)).
__device__ __forceinline__ void cw_trsm(int q, int T, int wave, const double* X, double* Y, const double* Wb, GPTR(double) S, int n, int lr, int lk)
{
    const int ns = T - 2 * q - 2;
    for (int id = wave; id < 2 * ns; id += 8) {
        const int h = id < ns ? 1 : 0, i = id < ns ? id : id - ns;
        const double* st = X + (i + 2) * CW_STRIP;
        double* dst = Y + i * CW_STRIP;
        GPTR(double) Srow = S + (size_t)((2 * q + 2 + i) * 16) * n + 32 * q;
        if (h) cw_trsm_item<1>(st, dst, Wb, Srow, n, lr, lk);
        else cw_trsm_item<0>(st, dst, Wb, Srow, n, lr, lk);
    }
}

__global__ __launch_bounds__(CW_THREADS) void k_chol_wg(const BaView* __restrict__ views)
{
    BA_VIEW(v);
    if (!cw_fits(v.dim) || v.band_hbw >= 0) return;      // larger systems go through the panel-pair chain, banded ones through k_chol_band
    if (ba_idle(v.ctl)) return;
    extern __shared__ __attribute__((aligned(16))) double cw_lds[];
    double* const X = cw_lds;                            // raw strips of the panel being factored (19)
    double* const Y = cw_lds + CW_BUF0;                  // L strips of the last finished panel (17)
    double* const Wb = Y + CW_BUF1;
    double* const tvec = Wb + CW_W;
    double* const xvec = tvec + CW_VEC;
    unsigned char* const tab = reinterpret_cast<unsigned char*>(xvec + CW_VEC);     // tile row / tile column of (wave, slot); then flags
    int* const s_fail = reinterpret_cast<int*>(tab + 2 * CW_UW * CW_SLOTS + 4);
    int* const s_diag = s_fail + 1;                      // diagonal-block tiles written so far (all panels)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int n = v.dim_pad, dim = v.dim;
    const int T = (dim + 1 + 15) >> 4;                   // tile rows that exist (the rest of the padded matrix is identity)
    const int P = (T + 1) >> 1;                          // panels
    GPTR(double) S = v.S;
    GPTR(double) Wg = v.Minv;                            // W_q blocks, [q][32][32]

    // ---- tile table: tiles (ti >= tk >= 2) in column-major order go round robin to the 7 owning wavefronts
    if (tid < CW_UW * CW_SLOTS) { tab[2 * tid] = 255; tab[2 * tid + 1] = 255; }
    if (tid == 0) { *s_fail = 0; *s_diag = 0; }
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int tk = 2; tk < T; ++tk)
            for (int ti = tk; ti < T; ++ti, ++t) {
                const int w = t % CW_UW, s = t / CW_UW;
                tab[2 * (w * CW_SLOTS + s)] = (unsigned char)ti; tab[2 * (w * CW_SLOTS + s) + 1] = (unsigned char)tk;
            }
    }

    if (wave == CW_UW) {
        // ======================= wavefront 7: the diagonal blocks =======================
        int want = 0;
        for (int q = 0; q < P; ++q) {
            const bool second = 2 * q + 1 < T;           // the block's second tile row exists
            // The block by DPP strips (chol_panel_core, the panel chain's arithmetic: 2 x 2.8 k cycles + the block product, against
            // 2 x 10 k for the readlane form this kernel used): lane l carries row l & 15 of the replicated diagonal block of each
            // strip, lanes 0-15 ride rows 16..31 of the block along, lanes 16-47 the identity rows 0..31 (they become L^-T).
            PanelRegs p;
            const int r = lane & 15, ir = lane - 16;     // ir: identity row of lanes 16-47
            if (q == 0) {
                // the first block straight from memory, while the others bring panel 0 into LDS
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    p.dA[c] = c <= r ? S[(size_t)r * n + c] : 0.0;
                    double xv = 0.0, yv = 0.0;
                    if (lane < 16) { xv = second ? S[(size_t)(16 + lane) * n + c] : 0.0; yv = second ? (c <= lane ? S[(size_t)(16 + lane) * n + 16 + c] : 0.0) : (c == lane ? 1.0 : 0.0); }
                    else if (lane < 48) { xv = c == ir ? 1.0 : 0.0; yv = 16 + c == ir ? 1.0 : 0.0; }
                    p.x[c] = xv; p.y[c] = yv;
                }
            } else {
                // wait for the block's tiles (a counter in LDS: no workgroup barrier, the other wavefronts go on with the lookahead)
                want += second ? 3 : 1;
                while (__hip_atomic_load(s_diag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                // every lane reads (valid LDS addresses), then selects: block rows / identity rows, no branch per element
                const double* row1 = X + CW_STRIP;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const double d0 = X[cw_swz(r, c)], x1 = row1[cw_swz(r, c)], y1 = row1[cw_swz(r, 16 + c)];
                    p.dA[c] = c <= r ? d0 : 0.0;
                    double xv = 0.0, yv = 0.0;
                    if (lane < 16) { xv = second ? x1 : 0.0; yv = second ? (c <= lane ? y1 : 0.0) : (c == lane ? 1.0 : 0.0); }
                    else if (lane < 48) { xv = c == ir ? 1.0 : 0.0; yv = 16 + c == ir ? 1.0 : 0.0; }
                    p.x[c] = xv; p.y[c] = yv;
                }
            }
            const bool fail = chol_panel_core<48>(p, lane, Wb);          // scratch: Wb + the two vectors behind it (1664 >= 2 x 48 x 17 doubles)
            if (__ballot(fail) != 0 && lane == 0) *s_fail = 1;
            if (lane >= 16 && lane < 48) {               // W_q = L^-T: row ir = [x | y] from the diagonal on
#pragma unroll
                for (int c = 0; c < 16; ++c) { Wb[cw_swz(ir, c)] = c >= ir ? p.x[c] : 0.0; Wb[cw_swz(ir, 16 + c)] = 16 + c >= ir ? p.y[c] : 0.0; }
            }
            __syncthreads();                             // (2) W_q is in LDS, panel q's raw strips are in X, nobody reads Y any more
            cw_trsm(q, T, wave, X, Y, Wb, S, n, lr, lk);
            // the block itself to memory, off the critical path: L (the row of the right-hand side may live here) and W_q
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) if (c <= lane) S[(size_t)(32 * q + lane) * n + 32 * q + c] = p.dA[c];
                if (second) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        S[(size_t)(32 * q + 16 + lane) * n + 32 * q + c] = p.x[c];
                        if (c <= lane) S[(size_t)(32 * q + 16 + lane) * n + 32 * q + 16 + c] = p.dB[c];
                    }
                }
            } else if (lane < 48) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    Wg[(size_t)q * CW_W + ir * 32 + c] = c >= ir ? p.x[c] : 0.0;
                    Wg[(size_t)q * CW_W + ir * 32 + 16 + c] = 16 + c >= ir ? p.y[c] : 0.0;
                }
            }
            __syncthreads();                             // (3) the L strips of panel q are in Y
        }
    } else {
        // ======================= wavefronts 0-6: tiles =======================
        // panel 0 from memory into X (all strips)
        for (int e = tid; e < T * CW_STRIP; e += CW_UW * 64) {
            const int ti = e >> 9, r = (e >> 5) & 15, c = e & 31;
            X[ti * CW_STRIP + cw_swz(r, c)] = S[(size_t)(ti * 16 + r) * n + c];
        }
        int s_tt[CW_SLOTS];                              // tile row | tile column << 8 per slot (scalar registers)
#define s_ti_(s) (s_tt[s] & 255)
#define s_tk_(s) (s_tt[s] >> 8)
        f64x4 acc[CW_SLOTS];
#pragma unroll
        for (int s = 0; s < CW_SLOTS; ++s) {
            s_tt[s] = __builtin_amdgcn_readfirstlane((int)tab[2 * (wave * CW_SLOTS + s)] | ((int)tab[2 * (wave * CW_SLOTS + s) + 1] << 8));
            acc[s] = f64x4{0, 0, 0, 0};
            if (s_tk_(s) != 255) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) acc[s][q4] = S[(size_t)(s_ti_(s) * 16 + lk + 4 * q4) * n + s_tk_(s) * 16 + lr];
            }
        }
        for (int q = 0; q < P; ++q) {
            const int c0 = 2 * q, c1 = 2 * q + 1;
            if (q > 0) {
                // 1. the tiles of panel q's tile columns: last update (panel q - 1: L strips in Y, tile row ti at ti - 2q), then into X;
                //    the diagonal block's tiles come first in every wavefront's slot order and are announced to wavefront 7
#pragma unroll
                for (int s = 0; s < CW_SLOTS; ++s) {
                    if (s_tk_(s) == c0 || s_tk_(s) == c1) {
                        acc[s] = cw_tile_update(Y + (s_ti_(s) - 2 * q) * CW_STRIP, Y + (s_tk_(s) - 2 * q) * CW_STRIP, lr, lk, acc[s]);
                        double* dst = X + (s_ti_(s) - 2 * q) * CW_STRIP;
                        const int cb = (s_tk_(s) - c0) * 16;
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) dst[cw_swz(lk + 4 * q4, cb + lr)] = acc[s][q4];
                        if (s_ti_(s) <= c1) {
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) __hip_atomic_fetch_add(s_diag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
                // 2. lookahead: the rest of the trailing matrix sees panel q - 1 while wavefront 7 factors the diagonal block
#pragma unroll
                for (int s = 0; s < CW_SLOTS; ++s) {
                    if (s_tk_(s) > c1 && s_tk_(s) != 255)
                        acc[s] = cw_tile_update(Y + (s_ti_(s) - 2 * q) * CW_STRIP, Y + (s_tk_(s) - 2 * q) * CW_STRIP, lr, lk, acc[s]);
                }
            }
            __syncthreads();                             // (2)
            // 3. strips below the diagonal block: L = A W_q on the matrix cores, X -> Y (all eight wavefronts share the items)
            cw_trsm(q, T, wave, X, Y, Wb, S, n, lr, lk);
            __syncthreads();                             // (3)
        }
#undef s_ti_
#undef s_tk_
    }

    // ======================= backward substitution: x = L^-T y, y = row `dim` of L =======================
    __syncthreads();                                     // every strip / block of L and every W_q is in memory (workgroup scope: one CU)
    if (tid < CW_VEC) tvec[tid] = tid < dim ? S[(size_t)dim * n + tid] : 0.0;
    // prefetch for the last panel: W_q (two entries per thread) and the rows of L left of the block (thread = column)
    int q = P - 1;
    double wq0 = Wg[(size_t)q * CW_W + tid], wq1 = Wg[(size_t)q * CW_W + 512 + tid];      // W_q[r][c], r = tid / 32 (+ 16), c = tid % 32
    double lrow[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { const int r = 32 * q + i; lrow[i] = (tid < 32 * q && r < dim) ? S[(size_t)r * n + tid] : 0.0; }
    __syncthreads();
    for (; q >= 0; --q) {
        const int r0 = 32 * q, nr = min(32, dim - r0);
        // x_q = W_q z_q over the block's real rows / columns (the rows of the right-hand side and of the padding take no part)
        {
            const int r = tid >> 5, c = tid & 31;
            const double z = c < nr ? tvec[r0 + c] : 0.0;
            double p0 = (r < nr && c < nr && c >= r) ? wq0 * z : 0.0;
            double p1 = (r + 16 < nr && c < nr && c >= r + 16) ? wq1 * z : 0.0;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { p0 += __shfl_xor(p0, o, 32); p1 += __shfl_xor(p1, o, 32); }
            if (c == 0) {
                if (r < nr) { xvec[r] = p0; v.xp[r0 + r] = p0; }
                if (r + 16 < nr) { xvec[r + 16] = p1; v.xp[r0 + r + 16] = p1; }
            }
        }
        if (q > 0) { wq0 = Wg[(size_t)(q - 1) * CW_W + tid]; wq1 = Wg[(size_t)(q - 1) * CW_W + 512 + tid]; }
        __syncthreads();
        // earlier columns lose the block's contribution: t[c] -= sum_r L[r0 + r][c] x[r]   (rows beyond dim were loaded as zeros)
        if (tid < r0) {
            double sum = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) sum += lrow[i] * (i < nr ? xvec[i] : 0.0);
            tvec[tid] -= sum;
        }
        if (q > 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) lrow[i] = (tid < 32 * (q - 1)) ? S[(size_t)(32 * (q - 1) + i) * n + tid] : 0.0;
        }
        __syncthreads();
    }
    if (tid == 0 && *s_fail) v.scal[5] = 1.0;
}
