// Dev check: relative error of v_rcp_f64 and of the refined reciprocal used by the kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* r0, double* r1, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i];
    double r = __builtin_amdgcn_rcp(a);
    r0[i] = r;
    double e = fma(-a, r, 1.0);
    double p = fma(e, e, e);
    r1[i] = fma(r, p, r);
}
int main()
{
    const int n = 1 << 20;
    double *hx = new double[n], *h0 = new double[n], *h1 = new double[n];
    for (int i = 0; i < n; ++i) hx[i] = std::exp((i / double(n)) * 60.0 - 30.0) * (1.0 + 0.37 * ((i * 2654435761u) % 1000) / 1000.0);
    double *dx, *d0, *d1;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, n);
    hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)hx[i];
        m0 = fmax(m0, (double)fabsl(((long double)h0[i] - t) / t));
        m1 = fmax(m1, (double)fabsl(((long double)h1[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err %.3e (2^%.1f); refined max rel err %.3e (%.2f ulp)\n", m0, log2(m0), m1, m1 / 1.11e-16);
    return 0;
}
