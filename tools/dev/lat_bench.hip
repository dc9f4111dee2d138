// latency / issue micro-measurements on one wavefront: f64 fma chain, independent fma, readlane, mfma f64 chain and independent, rcp, rsq
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define N 4096
template <int V>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, double x0)
{
    double x = x0 + threadIdx.x * 1e-9, y = x0 * 0.5, z = x0 * 0.25, w = x0 * 0.125;
    f64x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0}, acc4 = {0, 0, 0, 0};
    int li = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N / 4; ++i) {
        if (V == 0) { x = fma(x, 1.0000001, 1e-9); x = fma(x, 1.0000001, 1e-9); x = fma(x, 1.0000001, 1e-9); x = fma(x, 1.0000001, 1e-9); }   // dependent
        if (V == 1) { x = fma(x, 1.0000001, 1e-9); y = fma(y, 1.0000001, 1e-9); z = fma(z, 1.0000001, 1e-9); w = fma(w, 1.0000001, 1e-9); }   // independent
        if (V == 2) { li = __builtin_amdgcn_readlane(li, 5) + threadIdx.x; li = __builtin_amdgcn_readlane(li, 6) + threadIdx.x; li = __builtin_amdgcn_readlane(li, 7) + threadIdx.x; li = __builtin_amdgcn_readlane(li, 8) + threadIdx.x; }  // readlane + add dependent
        if (V == 3) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); }
        if (V == 4) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc3, 0, 0, 0); acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc4, 0, 0, 0); }
        if (V == 5) { x = __builtin_amdgcn_rcp(x) + 1.0; x = __builtin_amdgcn_rcp(x) + 1.0; x = __builtin_amdgcn_rcp(x) + 1.0; x = __builtin_amdgcn_rcp(x) + 1.0; }   // rcp + add dependent
        if (V == 6) { x = __builtin_amdgcn_rsq(x) + 1.0; x = __builtin_amdgcn_rsq(x) + 1.0; x = __builtin_amdgcn_rsq(x) + 1.0; x = __builtin_amdgcn_rsq(x) + 1.0; }
        if (V == 7) {   // readlane pairs feeding independent fmas (the bulk pattern)
            const int lo = __double2loint(x), hi = __double2hiint(x);
            y = fma(-x, __hiloint2double(__builtin_amdgcn_readlane(hi, 1), __builtin_amdgcn_readlane(lo, 1)), y);
            z = fma(-x, __hiloint2double(__builtin_amdgcn_readlane(hi, 2), __builtin_amdgcn_readlane(lo, 2)), z);
            w = fma(-x, __hiloint2double(__builtin_amdgcn_readlane(hi, 3), __builtin_amdgcn_readlane(lo, 3)), w);
            acc[0] = fma(-x, __hiloint2double(__builtin_amdgcn_readlane(hi, 4), __builtin_amdgcn_readlane(lo, 4)), acc[0]);
        }
        if (V == 8) { float a = (float)x; a = fmaf(a, 1.0000001f, 1e-9f); a = fmaf(a, 1.0000001f, 1e-9f); a = fmaf(a, 1.0000001f, 1e-9f); a = fmaf(a, 1.0000001f, 1e-9f); x = a; }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + y + z + w + acc[0] + acc[1] + acc2[0] + acc3[0] + acc4[0] + li;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double* o; long long* c; hipMalloc(&o, 64 * 8); hipMalloc(&c, 8);
    const char* names[] = {"f64 fma dependent", "f64 fma independent", "readlane_b32 + add dependent", "mfma_f64_16x16x4 dependent", "mfma_f64_16x16x4 independent", "rcp_f64 + add dependent", "rsq_f64 + add dependent", "2 readlane + fma (bulk pattern)", "cvt+4 f32 fma dep+cvt"};
    for (int v = 0; v < 9; ++v) {
        for (int p = 0; p < 2; ++p) {
            switch (v) {
            case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, o, c, 1.0); break; case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, o, c, 1.0); break;
            case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, o, c, 1.0); break; case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, o, c, 1.0); break;
            case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, o, c, 1.0); break; case 5: hipLaunchKernelGGL(k<5>, 1, 64, 0, 0, o, c, 1.0); break;
            case 6: hipLaunchKernelGGL(k<6>, 1, 64, 0, 0, o, c, 1.0); break; case 7: hipLaunchKernelGGL(k<7>, 1, 64, 0, 0, o, c, 1.0); break;
            case 8: hipLaunchKernelGGL(k<8>, 1, 64, 0, 0, o, c, 1.0); break;
            }
            hipDeviceSynchronize();
        }
        long long cyc = 0; hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
        printf("%-36s %.2f cycles per op (loop of 4 per iteration)\n", names[v], (double)cyc / N);
    }
    return 0;
}
