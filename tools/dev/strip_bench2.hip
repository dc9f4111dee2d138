#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
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
// One pivot of a strip.  What every later pivot waits for is the chain  pivot -> 1 / sqrt -> scaled column -> next pivot, so it is
// kept as short as the arithmetic allows (measured: 190 cycles per pivot with two Goldschmidt steps and the next pivot read back
// from the updated column; a dependent FP64 operation costs 8-9 cycles here, not its 4 issue cycles):
//   * 1 / sqrt(d) = y0 (1 + e / 2 + 3 e^2 / 8), e = 1 - d y0^2, y0 = v_rsq_f64 (2^-24, measured): one cubic step, four dependent
//     levels, 1.4e-16 relative error over 4M samples (two Goldschmidt steps: seven levels, 2.1e-16);
//   * the positivity test runs beside v_rsq_f64 and costs one select after it (a failed pivot continues on 1.0: harmless finite
//     numbers, the factorisation is flagged and its result discarded);
//   * the next pivot is a(jj+1, jj+1) - l(jj+1)^2 with both operands broadcast BEFORE this pivot's root is known, so it follows
//     the root by two operations instead of waiting for the column update and a DPP read-back (same fma as the update: same bits).
__device__ __forceinline__ double pivot_rsqrt_cubic(double d, bool ok)
{
    double y0 = __builtin_amdgcn_rsq(d);
    y0 = ok ? y0 : 1.0;
    const double dg = ok ? d : 1.0;
    const double t = dg * y0;
    const double e = fma(-t, y0, 1.0);
    const double pp = fma(0.375, e, 0.5), ye = y0 * e;
    return fma(ye, pp, y0);
}
template <int JJ>
__device__ __forceinline__ void strip_step(double (&d)[16], double (&x)[16], double piv, bool& fail)
{
    double p = 0.0, q = 0.0;
    if constexpr (JJ < 15) { p = dpp_bcast<JJ + 1>(d[JJ]); q = dpp_bcast<JJ + 1>(d[JJ + 1]); }
    const bool ok = piv > 0.0;
    fail |= !ok;
    const double rs = pivot_rsqrt_cubic(piv, ok);
    const double l = d[JJ] * rs, lx = x[JJ] * rs;
    d[JJ] = l; x[JJ] = lx;
    if constexpr (JJ < 15) {
        const double lp = p * rs;
        const double next = fma(-lp, lp, q);
        dpp_rank1<JJ + 1>(d, l, l);
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

// ---- variant: one asm statement per instruction, so that the scheduler may interleave the pivot chain with the updates
template <int C>
__device__ __forceinline__ void fm1(double& a, double src, double mul)
{
    asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(src), "v"(mul), "n"(C));
}
__device__ __forceinline__ void dpp_settle(double& v) { asm("s_nop 1" : "+v"(v)); }      // two wait states between a VALU write and a DPP read of it
template <int L>
__device__ __forceinline__ double bc1(double v)
{
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
    return r;
}
template <int JJ, int C>
__device__ __forceinline__ void upd_cols(double (&d)[16], double (&x)[16], double l, double lx)
{
    if constexpr (C < 16) {
        fm1<C>(d[C], l, l);
        fm1<C>(x[C], l, lx);
        upd_cols<JJ, C + 1>(d, x, l, lx);
    }
}
template <int JJ>
__device__ __forceinline__ void strip_step2(double (&d)[16], double (&x)[16], double piv, bool& fail)
{
    double p = 0.0, q = 0.0;
    if constexpr (JJ < 15) { double a = d[JJ], b = d[JJ + 1]; dpp_settle(a); dpp_settle(b); p = bc1<JJ + 1>(a); q = bc1<JJ + 1>(b); d[JJ] = a; d[JJ + 1] = b; }
    const bool ok = piv > 0.0;
    fail |= !ok;
    const double rs = pivot_rsqrt_cubic(piv, ok);
    double l = d[JJ] * rs; const double lx = x[JJ] * rs;
    d[JJ] = l; x[JJ] = lx;
    if constexpr (JJ < 15) {
        const double lp = p * rs;
        const double next = fma(-lp, lp, q);
        dpp_settle(l);
        upd_cols<JJ, JJ + 1>(d, x, l, lx);
        strip_step2<JJ + 1>(d, x, next, fail);
    }
}
__device__ __forceinline__ bool strip_factor2(double (&d)[16], double (&x)[16]) { bool fail = false; double a = d[0]; dpp_settle(a); strip_step2<0>(d, x, bc1<0>(a), fail); return fail; }

// ---- variant 2: software-pipelined by hand -- the root of the NEXT pivot is started before this step's updates and its dependent
// operations are placed between groups of them (waves issue in order: a dependent chain only overlaps what stands between its links)
template <int C0, int C1>
__device__ __forceinline__ void upd_range(double (&d)[16], double (&x)[16], double l, double lx)
{
    if constexpr (C0 < C1) {
        fm1<C0>(d[C0], l, l);
        fm1<C0>(x[C0], l, lx);
        upd_range<C0 + 1, C1>(d, x, l, lx);
    }
}
constexpr int imin(int a, int b) { return a < b ? a : b; }
// on entry: rs = 1 / sqrt(pivot JJ), p = d[JJ] of lane JJ + 1, q = d[JJ + 1] of lane JJ + 1 (values before this step's updates)
template <int JJ>
__device__ __forceinline__ void strip_step3(double (&d)[16], double (&x)[16], double rs, double p, double q, bool& fail)
{
    double l = d[JJ] * rs;
    const double lx = x[JJ] * rs;
    d[JJ] = l; x[JJ] = lx;
    if constexpr (JJ < 15) {
        const double lp = p * rs;
        const double next = fma(-lp, lp, q);
        dpp_settle(l);
        constexpr int c1 = imin(JJ + 3, 16), rest = 16 - c1, g = (rest + 3) / 4;
        constexpr int c2 = imin(c1 + g, 16), c3 = imin(c2 + g, 16), c4 = imin(c3 + g, 16);
#define SB() __builtin_amdgcn_sched_barrier(0)
        SB();
        upd_range<JJ + 1, c1>(d, x, l, lx);                 // the two columns the next step's p / q come from
        SB();
        const bool ok = next > 0.0;
        fail |= !ok;
        double y0 = __builtin_amdgcn_rsq(next);
        SB();
        upd_range<c1, c2>(d, x, l, lx);
        SB();
        double pn = 0.0, qn = 0.0;
        if constexpr (JJ < 14) { double a = d[JJ + 1], b = d[JJ + 2]; dpp_settle(a); dpp_settle(b); pn = bc1<JJ + 2>(a); qn = bc1<JJ + 2>(b); d[JJ + 1] = a; d[JJ + 2] = b; }
        y0 = ok ? y0 : 1.0;
        const double dg = ok ? next : 1.0;
        const double t = dg * y0;
        SB();
        upd_range<c2, c3>(d, x, l, lx);
        SB();
        const double e = fma(-t, y0, 1.0);
        SB();
        upd_range<c3, c4>(d, x, l, lx);
        SB();
        const double pp = fma(0.375, e, 0.5), ye = y0 * e;
        SB();
        upd_range<c4, 16>(d, x, l, lx);
        SB();
        const double rsn = fma(ye, pp, y0);
        strip_step3<JJ + 1>(d, x, rsn, pn, qn, fail);
    }
}
__device__ __forceinline__ bool strip_factor3(double (&d)[16], double (&x)[16])
{
    bool fail = false;
    double a = d[0], b = d[1]; dpp_settle(a); dpp_settle(b);
    const double piv = bc1<0>(a), p = bc1<1>(a), q = bc1<1>(b);
    d[0] = a; d[1] = b;
    const bool ok = piv > 0.0; fail |= !ok;
    strip_step3<0>(d, x, pivot_rsqrt_cubic(piv, ok), p, q, fail);
    return fail;
}

// ---- variant 3: fewest instructions per pivot -- pivot read back from the updated column (one broadcast), guard before the root
// (compare + one 64-bit select), cubic root step; block asm for the updates
template <int JJ>
__device__ __forceinline__ void strip_step4(double (&d)[16], double (&x)[16], double piv, bool& fail)
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
        strip_step4<JJ + 1>(d, x, next, fail);
    }
}
__device__ __forceinline__ bool strip_factor4(double (&d)[16], double (&x)[16]) { bool fail = false; strip_step4<0>(d, x, dpp_bcast<0>(d[0]), fail); return fail; }
template <int V>
__global__ __launch_bounds__(64) void k_bench(const double* A, double* out, int reps, long long* cyc)
{
    const int lane = threadIdx.x, r = lane & 15;
    double d0[16], x0[16], d[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { d0[c] = A[r * 16 + c]; x0[c] = A[(16 + lane) * 16 + c]; }
    double carry = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { d[c] = d0[c] + carry; x[c] = x0[c]; }
        if (V == 0) strip_factor(d, x); else if (V == 1) strip_factor2(d, x); else if (V == 2) strip_factor3(d, x); else strip_factor4(d, x);
        carry = d[15] * 1e-300;
    }
    const long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int c = 0; c < 16; ++c) { if (lane < 16) out[lane * 16 + c] = d[c]; out[(16 + lane) * 16 + c] = x[c]; }
    if (lane == 0) cyc[0] = t1 - t0;
}
int main()
{
    const int n = 16, rows = 80;
    std::vector<double> A(rows * n), L(rows * n), G(rows * n);
    srand(1);
    for (auto& g : G) g = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < rows; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += G[i * n + k] * G[j * n + k]; A[i * n + j] = s + (i == j ? 4.0 : 0.0); }
    L = A;
    for (int j = 0; j < n; ++j) {
        double dd = L[j * n + j];
        for (int k = 0; k < j; ++k) dd -= L[j * n + k] * L[j * n + k];
        dd = sqrt(dd);
        for (int i = j; i < rows; ++i) { double s = A[i * n + j]; for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k]; L[i * n + j] = i == j ? dd : s / dd; }
    }
    double *dA, *dO; long long* dC;
    hipMalloc(&dA, 8 * rows * n); hipMalloc(&dO, 8 * rows * n); hipMalloc(&dC, 8);
    hipMemcpy(dA, A.data(), 8 * rows * n, hipMemcpyHostToDevice);
    const int reps = 2000;
    for (int v = 0; v < 4; ++v) {
        for (int pass = 0; pass < 2; ++pass) {
            if (v == 0) hipLaunchKernelGGL(k_bench<0>, 1, 64, 0, 0, dA, dO, reps, dC); else if (v == 1) hipLaunchKernelGGL(k_bench<1>, 1, 64, 0, 0, dA, dO, reps, dC); else if (v == 2) hipLaunchKernelGGL(k_bench<2>, 1, 64, 0, 0, dA, dO, reps, dC); else hipLaunchKernelGGL(k_bench<3>, 1, 64, 0, 0, dA, dO, reps, dC);
            hipDeviceSynchronize();
        }
        std::vector<double> O(rows * n); long long cyc = 0;
        hipMemcpy(O.data(), dO, 8 * rows * n, hipMemcpyDeviceToHost); hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost);
        double err = 0;
        for (int i = 0; i < rows; ++i) for (int j = 0; j < n; ++j) if (i >= j) err = fmax(err, fabs(O[i * n + j] - L[i * n + j]));
        printf("%s: %.0f cycles per strip (%.1f per pivot), max err %.3e\n", v == 0 ? "block asm (committed)      " : v == 1 ? "one asm per instruction    " : v == 2 ? "software pipelined by hand " : "read-back, guard first     ", (double)cyc / reps, (double)cyc / reps / 16, err);
    }
    return 0;
}
