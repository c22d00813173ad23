// Dev microbenchmark: issue rate of v_fma_f64 vs v_fmac_f64_dpp vs v_mov_b64_dpp (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define DPPM " row_mask:0xf bank_mask:0xf"
template <int MODE>
__global__ void k(double* out, int iters, double m)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001 + i;
    double s = threadIdx.x * 0.5 + 1.0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(m));
        } else if (MODE == 1) {
            asm volatile("s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0" DPPM "\n\t"
                "v_fmac_f64_dpp %1, %8, %9 row_newbcast:1" DPPM "\n\t"
                "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2" DPPM "\n\t"
                "v_fmac_f64_dpp %3, %8, %9 row_newbcast:3" DPPM "\n\t"
                "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4" DPPM "\n\t"
                "v_fmac_f64_dpp %5, %8, %9 row_newbcast:5" DPPM "\n\t"
                "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6" DPPM "\n\t"
                "v_fmac_f64_dpp %7, %8, %9 row_newbcast:7" DPPM "\n\t"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                : "v"(s), "v"(m));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3" DPPM : "=v"(a[i]) : "v"(s));
        } else if (MODE == 4) {  // dependent chain fma
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[0]) : "v"(s), "v"(m));
        } else if (MODE == 5) {  // dependent chain fmac dpp
            asm volatile("s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0" DPPM "\n\t"
                "v_fmac_f64_dpp %0, %1, %2 row_newbcast:1" DPPM "\n\t"
                "v_fmac_f64_dpp %0, %1, %2 row_newbcast:2" DPPM "\n\t"
                "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3" DPPM "\n\t"
                "v_fmac_f64_dpp %0, %1, %2 row_newbcast:4" DPPM "\n\t"
                "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5" DPPM "\n\t"
                "v_fmac_f64_dpp %0, %1, %2 row_newbcast:6" DPPM "\n\t"
                "v_fmac_f64_dpp %0, %1, %2 row_newbcast:7" DPPM "\n\t"
                : "+v"(a[0]) : "v"(s), "v"(m));
        } else if (MODE == 6) {  // f32 fma for reference
#pragma unroll
            for (int i = 0; i < 8; ++i) { float f = (float)a[i]; asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(f) : "v"((float)s)); a[i] = f; }
        } else if (MODE == 7) {  // v_rcp_f64 throughput
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f64_e32 %0, %1" : "=v"(a[i]) : "v"(s));
        } else if (MODE == 8) {  // v_mul_f64
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "v"(s), "v"(m));
        }
    }
    double r = 0; for (int i = 0; i < 8; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, double* d, int wpb)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE><<<256, 64 * wpb>>>(d, 100, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, 64 * wpb>>>(d, iters, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = wpb/4; instructions per wave = iters*8
    double ns_per_inst = ms * 1e6 / (double(iters) * 8) / (wpb / 4.0);
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f clk @2.4GHz)\n", name, wpb / 4, ms, ns_per_inst, ns_per_inst * 2.4);
}
int main()
{
    double* d; hipMalloc(&d, 256 * 1024 * 8);
    for (int wpb : {4, 8}) {
        run<0>("v_fmac_f64 (indep)", d, wpb);
        run<1>("v_fmac_f64_dpp (indep)", d, wpb);
        run<2>("v_mov_b64_dpp", d, wpb);
        run<4>("v_fmac_f64 (dependent)", d, wpb);
        run<5>("v_fmac_f64_dpp (dependent)", d, wpb);
        run<7>("v_rcp_f64", d, wpb);
        run<8>("v_mul_f64", d, wpb);
    }
    return 0;
}
