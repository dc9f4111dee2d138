// micro-benchmark: latency of a 16-column register panel factorisation (one wavefront, lane = row), several formulations
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pivot_rsqrt64(double d)
{
    double rs = __builtin_amdgcn_rsq(d);
    rs = rs * fma(-0.5 * d * rs, rs, 1.5);
    return rs * fma(-0.5 * d * rs, rs, 1.5);
}
__device__ __forceinline__ double rcp_nr(double d)
{
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double r1 = fma(r0, e, r0);
    const double ee = e * e;
    return fma(r1, ee, r1);
}
// variant 0: as in ba.hip today
__device__ __forceinline__ void fac_v0(double (&a)[16], int lane)
{
    double dcur = readlane_f64(a[0], 0);
    double rs = pivot_rsqrt64(dcur);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const double lcol = a[jj] * rs;
        a[jj] = lane == jj ? dcur * rs : lcol;
        if (jj + 1 < 16) {
            a[jj + 1] = fma(-lcol, readlane_f64(lcol, jj + 1), a[jj + 1]);
            double dn = readlane_f64(a[jj + 1], jj + 1);
            const double rn = pivot_rsqrt64(dn);
#pragma unroll
            for (int c = jj + 2; c < 16; ++c) a[c] = fma(-lcol, readlane_f64(lcol, c), a[c]);
            dcur = dn; rs = rn;
        }
    }
}
// variant 1: LDL^T with the pivot recurrence on a uniform path (reciprocal + 1 fma per pivot), square roots once at the end
__device__ __forceinline__ void fac_v1(double (&a)[16], int lane)
{
    double d = readlane_f64(a[0], 0);
    double rinv = rcp_nr(d);
    double dvec = 1.0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const double u = a[jj];
        dvec = lane == jj ? d : dvec;
        const double t = u * rinv;
        if (jj + 1 < 16) {
            const double p = readlane_f64(u, jj + 1), q = readlane_f64(a[jj + 1], jj + 1);
            const double dn = fma(-(p * p), rinv, q);
            const double rn = rcp_nr(dn);
#pragma unroll
            for (int c = jj + 1; c < 16; ++c) a[c] = fma(-t, readlane_f64(u, c), a[c]);
            d = dn; rinv = rn;
        }
    }
    const double rsv = pivot_rsqrt64(dvec);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) a[jj] *= readlane_f64(rsv, jj);
}
// variant 2: as 1, multipliers through LDS broadcast reads instead of v_readlane
__device__ __forceinline__ void fac_v2(double (&a)[16], int lane, double* scr)
{
    double d = readlane_f64(a[0], 0);
    double rinv = rcp_nr(d);
    double dvec = 1.0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const double u = a[jj];
        dvec = lane == jj ? d : dvec;
        const double t = u * rinv;
        if (jj + 1 < 16) {
            const double p = readlane_f64(u, jj + 1), q = readlane_f64(a[jj + 1], jj + 1);
            const double dn = fma(-(p * p), rinv, q);
            const double rn = rcp_nr(dn);
            double* s = scr + (jj & 1) * 16;
            if (lane < 16) s[lane] = u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = jj + 1; c < 16; ++c) a[c] = fma(-t, s[c], a[c]);
            d = dn; rinv = rn;
        }
    }
    const double rsv = pivot_rsqrt64(dvec);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) a[jj] *= readlane_f64(rsv, jj);
}

template <int V>
__global__ __launch_bounds__(64) void k_bench(const double* A, double* out, int reps, long long* cyc)
{
    __shared__ double scr[64];
    const int lane = threadIdx.x;
    double a0[16], a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a0[c] = A[lane * 16 + c];
    double carry = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = a0[c] + carry;
        if (V == 0) fac_v0(a, lane);
        else if (V == 1) fac_v1(a, lane);
        else fac_v2(a, lane, scr);
        carry = readlane_f64(a[15], 15) * 1e-300;            // serialise the repetitions
    }
    const long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int c = 0; c < 16; ++c) out[lane * 16 + c] = a[c];
    if (lane == 0) cyc[0] = t1 - t0;
}

int main()
{
    const int n = 16, rows = 64;
    std::vector<double> A(rows * n), L(rows * n);
    // SPD 16x16 on top, 48 more rows below (the rows that ride along)
    srand(1);
    std::vector<double> G(rows * n);
    for (auto& g : G) g = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += G[i * n + k] * G[j * n + k];
            A[i * n + j] = s + (i == j ? 4.0 : 0.0);
        }
    // host reference: L = A L11^-T
    L = A;
    for (int j = 0; j < n; ++j) {
        double d = L[j * n + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        d = sqrt(d);
        for (int i = 0; i < rows; ++i) {
            if (i < j) continue;
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = i == j ? d : s / d;
        }
    }
    double *dA, *dO; long long* dC;
    CK(hipMalloc(&dA, sizeof(double) * rows * n)); CK(hipMalloc(&dO, sizeof(double) * rows * n)); CK(hipMalloc(&dC, 8));
    CK(hipMemcpy(dA, A.data(), sizeof(double) * rows * n, hipMemcpyHostToDevice));
    const int reps = 2000;
    for (int v = 0; v < 3; ++v) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(k_bench<0>, 1, 64, 0, 0, dA, dO, reps, dC);
            if (v == 1) hipLaunchKernelGGL(k_bench<1>, 1, 64, 0, 0, dA, dO, reps, dC);
            if (v == 2) hipLaunchKernelGGL(k_bench<2>, 1, 64, 0, 0, dA, dO, reps, dC);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double> O(rows * n); long long cyc = 0;
        CK(hipMemcpy(O.data(), dO, sizeof(double) * rows * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost));
        double err = 0;
        for (int i = 0; i < rows; ++i) for (int j = 0; j < n; ++j) if (i >= j) err = fmax(err, fabs(O[i * n + j] - L[i * n + j]));
        printf("variant %d: %.3f us per 16-column panel (%.1f ns per pivot), counter %.0f per panel, max err %.3e\n", v, ms * 1e3 / reps, ms * 1e6 / reps / 16, (double)cyc / reps, err);
    }
    return 0;
}
