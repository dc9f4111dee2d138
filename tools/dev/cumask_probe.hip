// CU-mask probe (MI355X): (1) which CUs a stream created with hipExtStreamCreateWithCUMask(all ones except the low 8 r bits) uses
// (expected: bit b -> XCC b % 8, so r CUs of every XCC stay free); (2) how long a small, LDS-heavy kernel on a high-priority
// stream takes while a chip-filling kernel of short workgroups runs on (a) an ordinary stream, (b) the masked stream.
// build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
// Result (round 3): the mask works as the KFD maps it (r = 4: 28 CUs of every XCC; r = 8: 24), but a front end restricted that way
// is 33 % slower at r = 4 (0.65 -> 0.86 ms per 16-frame step) and the bundle adjustment beside it does not get faster (its
// workgroups may still land on shared CUs): step 4.19 -> 4.30 ms.  Not used.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 1; } } while (0)

__global__ void k_where(unsigned* out)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
    }
}
// short workgroups, many of them: what the front end looks like to the dispatcher
__global__ __launch_bounds__(256) void k_filler(float* sink, int iters)
{
    __shared__ float s[1536];
    float a = threadIdx.x * 0.001f;
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    s[threadIdx.x] = a; __syncthreads();
    if (a == 12345.f) sink[blockIdx.x] = s[(threadIdx.x + 1) & 255];
}
// a few workgroups with 90 KB of LDS each: what a panel launch of the factorisation looks like
__global__ __launch_bounds__(256) void k_small(float* sink, int iters)
{
    extern __shared__ float big[];
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    big[threadIdx.x] = a; __syncthreads();
    if (a == 12345.f) sink[blockIdx.x] = big[(threadIdx.x + 1) & 255];
}

int main(int argc, char** argv)
{
    const int r = argc > 1 ? atoi(argv[1]) : 4;          // CUs of every XCC kept away from the masked stream
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs, reserve %d per XCC\n", p.name, p.multiProcessorCount, r);
    uint32_t mask[8]; for (auto& m : mask) m = 0xffffffffu;
    for (int b = 0; b < 8 * r; ++b) mask[b / 32] &= ~(1u << (b % 32));
    hipStream_t masked, plain, hi;
    CK(hipExtStreamCreateWithCUMask(&masked, 8, mask));
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    int lo_p = 0, hi_p = 0; CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, hi_p));
    unsigned* d_out; const int NW = 8192; CK(hipMalloc(&d_out, NW * 4));
    std::vector<unsigned> h(NW);
    for (int which = 0; which < 2; ++which) {
        hipStream_t s = which ? masked : plain;
        hipLaunchKernelGGL(k_where, dim3(NW), dim3(64), 0, s, d_out);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d_out, NW * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> cus; int per_xcc[16] = {0};
        for (unsigned v : h) cus.insert(v & 0xffffff00u & ~0xf0u ? (v >> 16) << 16 | (v & 0xff00) : (v >> 16) << 16 | (v & 0xff00));     // (xcc, se/sh/cu bits 8..15)
        std::set<unsigned> seen;
        for (unsigned v : h) { const unsigned key = (v >> 16) << 16 | (v & 0xff00); if (seen.insert(key).second) per_xcc[v >> 16]++; }
        printf("%s stream: %zu distinct CUs; per XCC:", which ? "masked" : "plain", seen.size());
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
        printf("\n");
    }
    float* sink; CK(hipMalloc(&sink, 1 << 22));
    CK(hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 3; ++which) {
        // which 0: nothing beside; 1: filler on the plain stream; 2: filler on the masked stream
        float sum = 0, worst = 0; const int reps = 20;
        for (int rep = 0; rep < reps; ++rep) {
            if (which) hipLaunchKernelGGL(k_filler, dim3(60000), dim3(256), 0, which == 1 ? plain : masked, sink, 2000);
            // a chain of 10 dependent small launches, like a factorisation
            CK(hipEventRecord(e0, hi));
            for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(k_small, dim3(24), dim3(256), 90 * 1024, hi, sink, 3000);
            CK(hipEventRecord(e1, hi));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            sum += ms; worst = ms > worst ? ms : worst;
            CK(hipDeviceSynchronize());
        }
        printf("10 dependent 24-WG / 90-KB-LDS launches on the high-priority stream, %s: mean %.1f us, worst %.1f us\n",
               which == 0 ? "alone" : which == 1 ? "beside a filler on an ordinary stream" : "beside a filler on the CU-masked stream", 1e3 * sum / reps, 1e3 * worst);
    }
    // how much the mask costs the filler itself
    for (int which = 0; which < 2; ++which) {
        hipStream_t s = which ? masked : plain;
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_filler, dim3(60000), dim3(256), 0, s, sink, 2000);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("filler alone on the %s stream: %.1f us\n", which ? "masked" : "plain", 1e3 * ms);
    }
    return 0;
}
