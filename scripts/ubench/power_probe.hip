// Dev microbenchmark (MI355X): is a stream of f64 FMAs limited by issue or by POWER (clock throttling)?
// Same per-wavefront instruction stream on 256 / 64 / 16 CUs, with all 64 lanes or 40 of 64 lanes active (exec),
// wall time per instruction and the shader clock seen by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o power_probe power_probe.hip && ./power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define DPPM " row_mask:0xf bank_mask:0xf"
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(double *out, long long *cyc, int iters, double m, unsigned long long lanes)
{
    double a0 = threadIdx.x * 0.001, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, s = threadIdx.x * 0.5 + 1.0;
    long long t0 = __builtin_readcyclecounter();
    if ((lanes >> (threadIdx.x & 63)) & 1ull) {
        for (int it = 0; it < iters; ++it) {
            if (MODE == 0) asm volatile(REP8("v_fmac_f64_e32 %0, %4, %5\n\tv_fmac_f64_e32 %1, %4, %5\n\tv_fmac_f64_e32 %2, %4, %5\n\tv_fmac_f64_e32 %3, %4, %5\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "v"(m));
            if (MODE == 1) asm volatile(REP8("v_fmac_f64_dpp %0, %4, %5 row_newbcast:1" DPPM "\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:2" DPPM "\n\tv_fmac_f64_dpp %2, %4, %5 row_newbcast:3" DPPM "\n\tv_fmac_f64_dpp %3, %4, %5 row_newbcast:4" DPPM "\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "v"(m));
            if (MODE == 2) asm volatile(REP8("v_fmac_f64_dpp %0, %4, %5 row_newbcast:1" DPPM "\n\tv_mov_b32 %6, %6\n\tv_fmac_f64_dpp %2, %4, %5 row_newbcast:3" DPPM "\n\tv_mov_b32 %6, %6\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "v"(m), "v"(iters));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}
template <int MODE>
void run(const char *name, double *d, long long *c, int blocks, int wpb, unsigned long long lanes)
{
    const int iters = 40000; // ~5 ms kernels: long enough for the power controller to settle
    k<MODE><<<blocks, 64 * wpb>>>(d, c, iters / 10, 1e-9, lanes);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 64 * wpb>>>(d, c, iters, 1e-9, lanes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    const double ninst = double(iters) * 32;
    printf("%-22s CUs=%3d waves/SIMD=%d lanes=%2d  %8.3f ms  %5.2f ns/inst/SIMD  %5.2f cyc/inst/wave  clock %.2f GHz\n", name, blocks,
           wpb / 4, __builtin_popcountll(lanes), ms, ms * 1e6 / ninst / (wpb / 4.0), double(h[0]) / ninst, double(h[0]) / (ms * 1e6));
}
int main()
{
    double *d;
    long long *c;
    hipMalloc(&d, 256 * 1024 * 8);
    hipMalloc(&c, 256 * 16 * 8);
    const unsigned long long all = ~0ull, ten = 0x03ff03ff03ff03ffull;
    for (int wpb : {4, 8}) {
        for (int blocks : {256, 64, 16}) {
            run<0>("v_fmac_f64", d, c, blocks, wpb, all);
            run<1>("v_fmac_f64_dpp", d, c, blocks, wpb, all);
        }
        run<0>("v_fmac_f64", d, c, 256, wpb, ten);
        run<1>("v_fmac_f64_dpp", d, c, 256, wpb, ten);
        run<2>("dpp + v_mov_b32", d, c, 256, wpb, all);
        run<2>("dpp + v_mov_b32", d, c, 256, wpb, ten);
    }
    return 0;
}
