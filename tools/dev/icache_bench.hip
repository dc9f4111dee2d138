// micro-benchmark: what does a COLD instruction cache cost a latency-bound kernel, and is it cold on every launch?
// A kernel of straight-line FP64 code (template: N_FMA dependent-free fma blocks, ~8 bytes each) runs its body TWICE per launch and
// stamps the wall clock (100 MHz) around each run: run 1 fetches its instructions, run 2 finds them in the instruction cache.
// The same kernel is launched several times back to back (a small other kernel in between or not): if run 1 of launch 2, 3, ... is as
// slow as run 1 of launch 1 the instruction cache does not survive the launch boundary.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// the clock as an asm volatile statement: the compiler keeps its order against the other asm volatile statements (the builtin floats)
__device__ __forceinline__ unsigned long long clk()
{
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define wall_clock64 clk
template <int N>
__device__ __forceinline__ void body(double (&a)[8], double x)
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i & 7]) : "v"(x));      // eight independent chains, N instructions of 8 bytes at N addresses
}
template <int N>
__global__ __launch_bounds__(64) void k_code(double* out, unsigned long long* stamps, double x)
{
    double a[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const unsigned long long t0 = wall_clock64();
    body<N>(a, x);
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    const unsigned long long t1 = wall_clock64();
    // the same code once more through a backward branch: same addresses, now cached
    double b[8] = {a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]};
    unsigned long long t2 = t1, t3 = t1;
    for (int rep = 0; rep < 2; ++rep) {
        t2 = wall_clock64();
#pragma nounroll
        for (int it = 0; it < 1; ++it) {
            body<N>(b, x);
            asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
        }
        t3 = wall_clock64();
        if (threadIdx.x == 0) { stamps[4 * blockIdx.x + 2 + rep] = t3 - t2; }
    }
    if (threadIdx.x == 0) { stamps[4 * blockIdx.x] = t1 - t0; }
    out[blockIdx.x * 64 + threadIdx.x] = b[0] + b[1] + b[2] + b[3] + b[4] + b[5] + b[6] + b[7];
}
__global__ void k_other(double* p) { p[blockIdx.x * 64 + threadIdx.x] += 1.0; }

template <int N>
int run(hipStream_t s, double* d, unsigned long long* ds, int blocks, bool other_between, bool graph)
{
    std::vector<unsigned long long> h((size_t)4 * blocks);
    printf("%5d instructions (~%d KB), %3d workgroups, %s%s:", N, N * 8 / 1024 + 1, blocks, graph ? "graph" : "direct", other_between ? ", another kernel between" : "");
    hipGraphExec_t ge = nullptr;
    if (graph) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_code<N>, dim3(blocks), dim3(64), 0, s, d, ds, 1.0000001);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    for (int launch = 0; launch < 4; ++launch) {
        if (other_between) hipLaunchKernelGGL(k_other, dim3(256), dim3(64), 0, s, d + 65536);
        if (graph) CK(hipGraphLaunch(ge, s)); else hipLaunchKernelGGL(k_code<N>, dim3(blocks), dim3(64), 0, s, d, ds, 1.0000001);
        CK(hipMemcpyAsync(h.data(), ds, h.size() * 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double cold = 0, hot = 0;
        for (int b = 0; b < blocks; ++b) { cold += (double)h[4 * b]; hot += (double)h[4 * b + 3]; }
        printf("  [launch %d: first run %.2f us, repeated %.2f us]", launch, 0.01 * cold / blocks, 0.01 * hot / blocks);
    }
    printf("\n");
    return 0;
}
int main()
{
    double* d; CK(hipMalloc(&d, 1 << 22)); CK(hipMemset(d, 0, 1 << 22));
    unsigned long long* ds; CK(hipMalloc(&ds, 1 << 16));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int blocks : {1, 256}) {
        for (int g = 0; g < 2; ++g) {
            if (run<500>(s, d, ds, blocks, false, g)) return 1;
            if (run<2000>(s, d, ds, blocks, false, g)) return 1;
            if (run<4000>(s, d, ds, blocks, false, g)) return 1;
        }
        if (run<4000>(s, d, ds, blocks, true, false)) return 1;
    }
    return 0;
}
