// micro-benchmark: time per dependent tiny kernel in a chain, direct stream launches vs a captured hipGraph
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_tiny(double* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1e-9; }
int main()
{
    double* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    hipStream_t s; CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, -1));
    const int N = 200;
    for (int blocks : {1, 32, 256}) {
        // direct
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, d, blocks * 256);
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("%3d blocks: direct launches %.2f us per kernel\n", blocks, us / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, d, blocks * 256);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("%3d blocks: graph            %.2f us per kernel\n", blocks, us / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
