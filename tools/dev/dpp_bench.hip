// micro-benchmark: 16-column LDL^T strip, D replicated in every 16-lane row, own rows ride along; multipliers by DPP row_newbcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define DPPF(c) "v_fmac_f64_dpp %" #c ", %16, %17 row_newbcast:" #c " row_mask:0xf bank_mask:0xf\n\t"
#define F15 DPPF(15)
#define F14 DPPF(14) F15
#define F13 DPPF(13) F14
#define F12 DPPF(12) F13
#define F11 DPPF(11) F12
#define F10 DPPF(10) F11
#define F9 DPPF(9) F10
#define F8 DPPF(8) F9
#define F7 DPPF(7) F8
#define F6 DPPF(6) F7
#define F5 DPPF(5) F6
#define F4 DPPF(4) F5
#define F3 DPPF(3) F4
#define F2 DPPF(2) F3
#define F1 DPPF(1) F2
#define ACC16(a) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
// a[c] += bcast(src, lane c of the row) * mul for c = FROM .. 15
template <int FROM>
__device__ __forceinline__ void dpp_rank1(double (&a)[16], double src, double mul)
{
#define CASE(k, S) if constexpr (FROM == k) asm("s_nop 1\n\t" S : ACC16(a) : "v"(src), "v"(mul));
    CASE(1, F1) CASE(2, F2) CASE(3, F3) CASE(4, F4) CASE(5, F5) CASE(6, F6) CASE(7, F7) CASE(8, F8)
    CASE(9, F9) CASE(10, F10) CASE(11, F11) CASE(12, F12) CASE(13, F13) CASE(14, F14) CASE(15, F15)
#undef CASE
}
template <int L>
__device__ __forceinline__ double dpp_bcast(double v)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
    return r;
}
__device__ __forceinline__ double rcp_nr(double d)
{
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double r1 = fma(r0, e, r0);
    const double ee = e * e;
    return fma(r1, ee, r1);
}
template <int JJ, int NS>
__device__ __forceinline__ void ldl_step(double (&d)[16], double (&x)[NS][16], double& piv, double (&dinv)[16])
{
    const double rinv = rcp_nr(piv);
    dinv[JJ] = rinv;
    if constexpr (JJ < 15) {
        const double nt = -d[JJ] * rinv;
        dpp_rank1<JJ + 1>(d, d[JJ], nt);
        piv = dpp_bcast<JJ + 1>(d[JJ + 1]);
#pragma unroll
        for (int s = 0; s < NS; ++s) { const double ntx = -x[s][JJ] * rinv; dpp_rank1<JJ + 1>(x[s], d[JJ], ntx); }
        ldl_step<(JJ < 15 ? JJ + 1 : 15), NS>(d, x, piv, dinv);
    }
}
template <int NS>
__global__ __launch_bounds__(64) void k_bench(const double* A, double* out, double* out_dinv, int reps, long long* cyc)
{
    const int lane = threadIdx.x, r = lane & 15;
    double d0[16], x0[NS][16], d[16], x[NS][16], dinv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        d0[c] = A[r * 16 + c];
#pragma unroll
        for (int s = 0; s < NS; ++s) x0[s][c] = A[(16 + s * 64 + lane) * 16 + c];
    }
    double carry = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            d[c] = d0[c] + carry;
#pragma unroll
            for (int s = 0; s < NS; ++s) x[s][c] = x0[s][c];
        }
        double piv = dpp_bcast<0>(d[0]);
        ldl_step<0, NS>(d, x, piv, dinv);
        carry = d[15] * 1e-300;
    }
    const long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (lane < 16) out[lane * 16 + c] = d[c];
#pragma unroll
        for (int s = 0; s < NS; ++s) out[(16 + s * 64 + lane) * 16 + c] = x[s][c];
        if (lane == 0) out_dinv[c] = dinv[c];
    }
    if (lane == 0) cyc[0] = t1 - t0;
}

int main()
{
    const int n = 16, rows = 16 + 3 * 64;
    std::vector<double> A(rows * n), U(rows * n), dv(n);
    srand(1);
    std::vector<double> G(rows * n);
    for (auto& g : G) g = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += G[i * n + k] * G[j * n + k];
            A[i * n + j] = s + (i == j ? 4.0 : 0.0);
        }
    // host reference: unscaled columns u_ij = a_ij - sum_k u_ik u_jk / d_k, d_j = u_jj
    U = A;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < rows; ++i) {
            if (i < j) continue;
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= U[i * n + k] * U[j * n + k] / U[k * n + k];
            U[i * n + j] = s;
        }
    double *dA, *dO, *dD; long long* dC;
    CK(hipMalloc(&dA, sizeof(double) * rows * n)); CK(hipMalloc(&dO, sizeof(double) * rows * n)); CK(hipMalloc(&dD, 128)); CK(hipMalloc(&dC, 8));
    CK(hipMemcpy(dA, A.data(), sizeof(double) * rows * n, hipMemcpyHostToDevice));
    const int reps = 2000;
    for (int ns = 1; ns <= 3; ++ns) {
        CK(hipMemset(dO, 0, sizeof(double) * rows * n));
        for (int pass = 0; pass < 2; ++pass) {
            if (ns == 1) hipLaunchKernelGGL(k_bench<1>, 1, 64, 0, 0, dA, dO, dD, reps, dC);
            if (ns == 2) hipLaunchKernelGGL(k_bench<2>, 1, 64, 0, 0, dA, dO, dD, reps, dC);
            if (ns == 3) hipLaunchKernelGGL(k_bench<3>, 1, 64, 0, 0, dA, dO, dD, reps, dC);
            CK(hipDeviceSynchronize());
        }
        std::vector<double> O(rows * n); long long cyc = 0; double di[16];
        CK(hipMemcpy(O.data(), dO, sizeof(double) * rows * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(di, dD, 128, hipMemcpyDeviceToHost));
        double err = 0, errd = 0;
        for (int i = 0; i < 16 + ns * 64; ++i) for (int j = 0; j < n; ++j) if (i >= j) err = fmax(err, fabs(O[i * n + j] - U[i * n + j]));
        for (int j = 0; j < n; ++j) errd = fmax(errd, fabs(di[j] * U[j * n + j] - 1.0));
        printf("D + %d own sets (%d rows): %.0f cycles per 16-column strip (%.1f per pivot), max err %.3e, dinv err %.3e\n", ns, 16 + ns * 64, (double)cyc / reps, (double)cyc / reps / 16, err, errd);
    }
    return 0;
}
