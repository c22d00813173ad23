// Dev microbenchmark: cost of wavefront-wide broadcast mechanisms (per CU, all four SIMDs busy).
//   0: ds_read_b128 at a wavefront-uniform address (LDS broadcast)     1: ds_read_b64 uniform
//   2: ds_read_b128 uniform + the 2 FMAs that consume it               3: v_readlane pair + FMA reading the SGPR pair
//   4: ds_read_b128 at per-lane addresses (row stride 288 B)           5: s_load_dwordx8 (scalar cache) + 4 FMAs on SGPRs
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(double* out, long long* cyc, int iters, const double* gsrc)
{
    __shared__ __attribute__((aligned(16))) double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 1e-3;
    __syncthreads();
    double a0 = threadIdx.x * 0.001, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, s = threadIdx.x * 0.5 + 1.0;
    v2d r0 = {0, 0}, r1 = {0, 0}, r2 = {0, 0}, r3 = {0, 0};
    const unsigned ubase = (unsigned)(size_t)lds + (threadIdx.x / 64) * 2048;          // uniform per wave
    const unsigned lbase = (unsigned)(size_t)lds + (threadIdx.x % 64) * 288 % 16384;   // per lane
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(REP8("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t") "s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(ubase));
        if (MODE == 1) asm volatile(REP8("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:16\n\tds_read_b64 %2, %4 offset:32\n\tds_read_b64 %3, %4 offset:48\n\t") "s_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(ubase));
        if (MODE == 2) {
            const v2d* p = reinterpret_cast<const v2d*>(lds + (threadIdx.x / 64) * 256) + (it & 31);
#pragma unroll
            for (int i = 0; i < 32; i += 4) {
                const v2d v0 = p[i], v1 = p[i + 1], v2 = p[i + 2], v3 = p[i + 3];
                a0 = fma(s, v0.x, a0); a1 = fma(s, v0.y, a1); a2 = fma(s, v1.x, a2); a3 = fma(s, v1.y, a3);
                a0 = fma(s, v2.x, a0); a1 = fma(s, v2.y, a1); a2 = fma(s, v3.x, a2); a3 = fma(s, v3.y, a3);
            }
        }
        if (MODE == 3) asm volatile(REP8("v_readlane_b32 s40, %4, 3\n\tv_readlane_b32 s41, %5, 3\n\tv_readlane_b32 s42, %4, 5\n\tv_readlane_b32 s43, %5, 5\n\t"
                                         "v_readlane_b32 s44, %4, 7\n\tv_readlane_b32 s45, %5, 7\n\tv_readlane_b32 s46, %4, 9\n\tv_readlane_b32 s47, %5, 9\n\t"
                                         "v_fmac_f64_e32 %0, s[40:41], %6\n\tv_fmac_f64_e32 %1, s[42:43], %6\n\tv_fmac_f64_e32 %2, s[44:45], %6\n\tv_fmac_f64_e32 %3, s[46:47], %6\n\t")
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(__double2loint(s)), "v"(__double2hiint(s)), "v"(s)
                                    : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
        if (MODE == 4) asm volatile(REP8("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t") "s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(lbase));
        if (MODE == 5) asm volatile(REP8("s_load_dwordx8 s[40:47], %4, 0x0\n\ts_waitcnt lgkmcnt(0)\n\t"
                                         "v_fmac_f64_e32 %0, s[40:41], %5\n\tv_fmac_f64_e32 %1, s[42:43], %5\n\tv_fmac_f64_e32 %2, s[44:45], %5\n\tv_fmac_f64_e32 %3, s[46:47], %5\n\t")
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(gsrc), "v"(s)
                                    : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + r0.x + r1.y + r2.x + r3.y;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}
template <int MODE> void run(const char* name, double* d, long long* c, int wpb, int per_iter, const double* g)
{
    const int iters = 2000;
    k<MODE><<<256, 64 * wpb>>>(d, c, iters, g);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, 64 * wpb>>>(d, c, iters, g);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double per = double(h[0]) / (double(iters) * per_iter);
    printf("%-44s waves/CU=%2d  %8.3f ms  %7.2f cyc per unit per wave -> %6.2f cyc per unit per CU\n", name, wpb, ms, per, per / wpb);
}
int main()
{
    double* d; long long* c; double* g; hipMalloc(&d, 256 * 1024 * 8); hipMalloc(&c, 256 * 16 * 8); hipMalloc(&g, 4096); hipMemset(g, 0, 4096);
    for (int wpb : {4, 8}) {
        run<0>("ds_read_b128 uniform (unit = 1 read)", d, c, wpb, 32, g);
        run<1>("ds_read_b64 uniform (unit = 1 read)", d, c, wpb, 32, g);
        run<4>("ds_read_b128 per-lane rows (unit = 1 read)", d, c, wpb, 32, g);
        run<2>("ds_read_b128 uniform + 2 FMA (unit = 1 FMA)", d, c, wpb, 64, g);
        run<3>("readlane pair + FMA (unit = 1 FMA)", d, c, wpb, 32, g);
        run<5>("s_load_dwordx8 + 4 FMA (unit = 1 FMA)", d, c, wpb, 32, g);
    }
    return 0;
}
