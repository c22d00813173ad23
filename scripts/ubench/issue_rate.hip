// Dev microbenchmark: cycles per wave-instruction (s_memtime) for a lone wave per SIMD vs 2 and 4 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#define DPPM " row_mask:0xf bank_mask:0xf"
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(double* out, long long* cyc, int iters, double m)
{
    double a0 = threadIdx.x * 0.001, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, s = threadIdx.x * 0.5 + 1.0;
    int i0 = threadIdx.x, i1 = 7;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(REP8("v_fmac_f64_e32 %0, %4, %5\n\tv_fmac_f64_e32 %1, %4, %5\n\tv_fmac_f64_e32 %2, %4, %5\n\tv_fmac_f64_e32 %3, %4, %5\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "v"(m));
        if (MODE == 1) asm volatile(REP8("v_fmac_f64_dpp %0, %4, %5 row_newbcast:1" DPPM "\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:2" DPPM "\n\tv_fmac_f64_dpp %2, %4, %5 row_newbcast:3" DPPM "\n\tv_fmac_f64_dpp %3, %4, %5 row_newbcast:4" DPPM "\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "v"(m));
        if (MODE == 2) asm volatile(REP8("v_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\t") : "+v"(i0) : "v"(i1));
        if (MODE == 3) asm volatile(REP8("s_nop 1\n\ts_nop 1\n\ts_nop 1\n\ts_nop 1\n\t"));
        if (MODE == 4) asm volatile(REP8("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"));
        if (MODE == 5) asm volatile(REP8("v_fmac_f64_e32 %0, %4, %5\n\tv_mov_b32 %6, %7\n\tv_fmac_f64_e32 %1, %4, %5\n\tv_mov_b32 %6, %7\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "v"(m), "v"(i0), "v"(i1));
        if (MODE == 6) asm volatile(REP8("v_fmac_f64_e32 %0, %4, %5\n\ts_nop 1\n\tv_fmac_f64_e32 %1, %4, %5\n\ts_nop 1\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "v"(m));
        if (MODE == 7) asm volatile(REP8("v_mov_b64_dpp %0, %4 row_newbcast:1" DPPM "\n\tv_mov_b64_dpp %1, %4 row_newbcast:2" DPPM "\n\tv_mov_b64_dpp %2, %4 row_newbcast:3" DPPM "\n\tv_mov_b64_dpp %3, %4 row_newbcast:4" DPPM "\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));
        if (MODE == 8) asm volatile(REP8("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s), "v"(m));
        if (MODE == 9) asm volatile(REP8("v_fmac_f32_e32 %0, %1, %1\n\tv_fmac_f32_e32 %0, %1, %1\n\tv_fmac_f32_e32 %0, %1, %1\n\tv_fmac_f32_e32 %0, %1, %1\n\t") : "+v"(i0) : "v"(i1));
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + i0;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}
template <int MODE> void run(const char* name, double* d, long long* c, int wpb)
{
    const int iters = 4000;
    k<MODE><<<256, 64 * wpb>>>(d, c, iters, 1e-9);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, 64 * wpb>>>(d, c, iters, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double per = double(h[0]) / (double(iters) * 32);
    printf("%-34s waves/SIMD=%d  %8.3f ms  %6.2f cyc/inst/wave (s_memtime) -> %5.2f cyc/inst per SIMD, clock ~%.2f GHz\n", name, wpb / 4, ms, per, per / (wpb / 4.0), double(h[0]) / (ms * 1e6));
}
int main()
{
    double* d; long long* c; hipMalloc(&d, 256 * 1024 * 8); hipMalloc(&c, 256 * 16 * 8);
    for (int wpb : {4, 8, 16}) {
        run<0>("v_fmac_f64 x4 indep", d, c, wpb);
        run<8>("v_fma_f64 (VOP3) x4 indep", d, c, wpb);
        run<1>("v_fmac_f64_dpp x4 indep", d, c, wpb);
        run<7>("v_mov_b64_dpp", d, c, wpb);
        run<2>("v_mov_b32 (dependent)", d, c, wpb);
        run<9>("v_fmac_f32 (dependent)", d, c, wpb);
        run<3>("s_nop 1", d, c, wpb);
        run<4>("s_nop 0", d, c, wpb);
        run<5>("fmac_f64 + v_mov_b32 alternating", d, c, wpb);
        run<6>("fmac_f64 + s_nop 1 alternating", d, c, wpb);
    }
    return 0;
}
