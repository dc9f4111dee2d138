// Which XCC does workgroup 0 of a launch run on?  (MI355X; the panel chain keeps its workgroups on the XCC of blocks b % 8 == 0.)
// 400 small launches on a high-priority stream, alone and beside a chip-filling kernel on another stream; prints the histogram of
// the XCC of block 0 and whether blocks b and b + 8 always share an XCC.
// build: hipcc --offload-arch=gfx950 -O2 -o xcc0_probe xcc0_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 1; } } while (0)
__global__ void k_where(unsigned* out)
{
    if (threadIdx.x == 0) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); out[blockIdx.x] = xcc & 15; }
}
__global__ void k_fill(float* p, int n)
{
    __shared__ float s[4096];
    float a = threadIdx.x;
    for (int i = 0; i < 2000; ++i) a = a * 1.0001f + 0.5f;
    s[threadIdx.x] = a;
    __syncthreads();
    if (blockIdx.x * 256 + threadIdx.x < n) p[blockIdx.x * 256 + threadIdx.x] = s[(threadIdx.x + 1) & 255];
}
int main()
{
    const int L = 400, G = 64;
    unsigned* d; float* f;
    CK(hipMalloc(&d, sizeof(unsigned) * L * G)); CK(hipMalloc(&f, sizeof(float) * 256 * 20000));
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s, t; CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking));
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(k_fill, dim3(20000), dim3(256), 0, t, f, 256 * 20000);
        for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, s, d + l * G);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned> h(L * G);
        CK(hipMemcpy(h.data(), d, sizeof(unsigned) * L * G, hipMemcpyDeviceToHost));
        int hist[16] = {0}, broken = 0;
        for (int l = 0; l < L; ++l) {
            hist[h[l * G]]++;
            for (int b = 0; b + 8 < G; ++b) if (h[l * G + b] != h[l * G + b + 8]) { ++broken; break; }
            for (int b = 0; b < 8; ++b) if (h[l * G + b] != (h[l * G] + b) % 8) { /* not consecutive */ }
        }
        printf("%s: XCC of block 0 over %d launches:", pass ? "beside a chip-filling kernel" : "alone", L);
        for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
        printf("; launches where b and b + 8 differ: %d\n", broken);
        CK(hipDeviceSynchronize());
    }
    return 0;
}
