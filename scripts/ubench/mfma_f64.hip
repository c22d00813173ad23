// Dev microbenchmark (MI355X): what v_mfma_f64_16x16x4_f64 costs next to the f64 VALU, and what a
// wavefront-uniform LDS read (broadcast) and a v_readlane pair cost -- the three ways an operand of the
// wide (16 < n <= 64) smoother can travel.  Also checks the f64 MFMA C/D layout (guide: col = lane&15,
// row = (lane>>4) + 4*reg) with A = I against an ASYMMETRIC B.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64 mfma_f64.hip && ./mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// MODE 0: MFMA f64 only (4 independent accumulator tiles)   1: v_fma_f64 only (8 independent chains)
// MODE 2: waves 0-3 of the workgroup MFMA, waves 4-7 FMA (same SIMDs when 8 waves per workgroup)
// MODE 3: uniform-address ds_read_b128 feeding 2 FMAs each    4: readlane pair + FMA
// MODE 5: dependent MFMA chain (latency)
template <int MODE>
__global__ void __launch_bounds__(512) k(double *out, int iters, double m)
{
    __shared__ __attribute__((aligned(16))) double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    double a = threadIdx.x * 1e-3 + 1.0, b = threadIdx.x * 2e-3 + 0.5;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = a + i;
    const bool do_mfma = MODE == 0 || MODE == 5 || (MODE == 2 && wave < 4);
    const bool do_fma = MODE == 1 || (MODE == 2 && wave >= 4);
    if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
            if (MODE == 5) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            }
        }
    }
    if (do_fma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(f[i]) : "v"(a), "v"(m));
        }
    }
    if (MODE == 3) {
        const v2d *l2 = reinterpret_cast<const v2d *>(lds);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v2d v = l2[(it * 4 + i) & 1023]; // wavefront-uniform address
                f[2 * i] = fma(v.x, a, f[2 * i]);
                f[2 * i + 1] = fma(v.y, a, f[2 * i + 1]);
            }
        }
    }
    if (MODE == 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(f[(i + 1) & 7]), i);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(f[(i + 1) & 7]), i);
                f[i] = fma(__hiloint2double(hi, lo), m, f[i]);
            }
        }
    }
    double r = c0.x + c0.y + c0.z + c0.w + c1.x + c2.y + c3.z;
    for (int i = 0; i < 8; ++i) r += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void layout_check(double *D)
{
    const int l = threadIdx.x;
    double acc[16 * 16];
    (void)acc;
    v4d c = {0, 0, 0, 0};
    // A = I (16x16 over 4 k-steps), B[k][j] = 100 k + j (asymmetric)
    for (int kk = 0; kk < 4; ++kk) {
        const int i = l & 15, k = 4 * kk + (l >> 4);
        const double av = (i == k) ? 1.0 : 0.0;              // A[i][k]
        const double bv = 100.0 * k + (l & 15);               // B[k][j], j = l & 15
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r]; // guide: row = (l>>4) + 4 reg, col = l&15
}

template <int MODE>
void run(const char *name, double *d, int threads, int blocks, int iters, double flop_per_iter_per_wave, double ops_per_iter)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0000001);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64;
    const double clk = 2.4e9; // nominal; the ratio between the modes is what matters
    const double waves_per_simd = waves / 1024.0;
    printf("%-44s %7.3f ms  %8.2f TFLOP/s  %7.1f cycles per op per SIMD (%.1f waves/SIMD)\n", name, ms,
           flop_per_iter_per_wave * iters * waves / (ms * 1e-3) / 1e12, ms * 1e-3 * clk / (ops_per_iter * iters * waves_per_simd),
           waves_per_simd);
}

int main()
{
    double *d;
    hipMalloc(&d, sizeof(double) * 1024 * 1024 * 8);
    double *D;
    hipMalloc(&D, sizeof(double) * 256);
    hipLaunchKernelGGL(layout_check, dim3(1), dim3(64), 0, 0, D);
    std::vector<double> h(256);
    hipMemcpy(h.data(), D, sizeof(double) * 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) bad += h[i * 16 + j] != 100.0 * i + j;
    printf("f64 MFMA layout check (A = I, B[k][j] = 100k + j): %d mismatches\n", bad);
    const int it = 20000;
    const double mf = 4 * 2.0 * 16 * 16 * 4, ff = 8 * 2.0 * 64;
    run<0>("mfma f64 16x16x4, 1 wave/SIMD", d, 256, 256, it, mf, 4);
    run<0>("mfma f64 16x16x4, 2 waves/SIMD", d, 512, 256, it, mf, 4);
    run<5>("mfma f64 dependent chain, 1 wave/SIMD", d, 256, 256, it, mf, 4);
    run<1>("v_fma_f64, 1 wave/SIMD", d, 256, 256, it, ff, 8);
    run<1>("v_fma_f64, 2 waves/SIMD", d, 512, 256, it, ff, 8);
    run<2>("mfma (waves 0-3) + fma (waves 4-7), 2/SIMD", d, 512, 256, it, (mf + ff) / 2, 6);
    run<3>("uniform ds_read_b128 + 2 fma, 1 wave/SIMD", d, 256, 256, it, ff, 4);
    run<3>("uniform ds_read_b128 + 2 fma, 2 waves/SIMD", d, 512, 256, it, ff, 4);
    run<4>("2 readlane + fma, 1 wave/SIMD", d, 256, 256, it, ff, 8);
    run<4>("2 readlane + fma, 2 waves/SIMD", d, 512, 256, it, ff, 8);
    return 0;
}
