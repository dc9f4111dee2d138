#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    const double a = __builtin_amdgcn_rsq(d);
    y0[i] = a;
    const double t = d * a, e = fma(-t, a, 1.0), pp = fma(0.375, e, 0.5), ye = a * e;
    y1[i] = fma(ye, pp, a);
    double g = d * a, h = 0.5 * a; double r = fma(-h, g, 0.5); g = fma(g, r, g); h = fma(h, r, h); r = fma(-h, g, 0.5); h = fma(h, r, h);
    y2[i] = h + h;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    srand(3);
    for (int i = 0; i < n; ++i) { const double m = 1.0 + rand() / (double)RAND_MAX; const int ex = rand() % 200 - 100; x[i] = ldexp(m, ex); }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, n / 256, 256, 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double ex = 1.0L / sqrtl((long double)x[i]);
        e0 = fmax(e0, (double)fabsl(((long double)a[i] - ex) / ex));
        e1 = fmax(e1, (double)fabsl(((long double)b[i] - ex) / ex));
        e2 = fmax(e2, (double)fabsl(((long double)c[i] - ex) / ex));
    }
    printf("v_rsq_f64 max rel err %.3e (2^%.1f); cubic step %.3e; two Goldschmidt steps %.3e\n", e0, log2(e0), e1, e2);
    return 0;
}
