// development: does hipExtAnyOrderLaunch let a kernel start while an earlier kernel of the SAME stream is still running (gfx950)?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_spin(unsigned long long ticks, unsigned long long* out) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); if (threadIdx.x == 0) out[0] = wall_clock64(); }
__global__ void k_stamp(unsigned long long* out) { if (threadIdx.x == 0) out[0] = wall_clock64(); }
int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long* d; hipMalloc(&d, 16);
    unsigned long long h[2];
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 20000ull /* 200 us of the 100 MHz clock */, d);
            if (mode == 0) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, d + 1);
            else hipExtLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d + 1);
            hipStreamSynchronize(s);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("%s: the second kernel stamped %.1f us %s the end of the first\n", mode ? "any-order launch" : "plain launch    ", 0.01 * (double)(h[1] > h[0] ? h[1] - h[0] : h[0] - h[1]), h[1] > h[0] ? "after" : "BEFORE");
        }
    }
    return 0;
}
