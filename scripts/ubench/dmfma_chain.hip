// Does a chain of dependent v_mfma_f64_4x4x4_4b_f64 compute exactly when TWO wavefronts share a SIMD?  (round 3: the
// 4x4x4 block smoother returned results off by ~1e-7, not reproducibly, only with two resident wavefronts per SIMD.)
// Integer-valued operands: every product and partial sum is exact in f64, so any deviation is a hardware/compiler hazard.
//   variant 0: CH interleaved accumulation chains, MFMAs back to back (what hipcc emits, its own wait states)
//   variant 1: the same with an LDS read (wavefront-private) feeding the A operand of every MFMA
//   variant 2: the same as 0 with the A operand rewritten by a VALU instruction right after each MFMA
// hipcc --offload-arch=gfx950 -O3 dmfma_chain.hip -o dmfma_chain && ./dmfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH, int VAR>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) chain(int iters, double *out, int pad_vgprs)
{
    __shared__ double lds[64 * 4];
    const int lane = threadIdx.x;
    // block b (slot): A_b = B_b = small integer matrices depending on lane; C accumulates
    double a = (double)((lane * 7 + blockIdx.x) % 5 - 2), b = (double)((lane * 3) % 7 - 3);
    for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = a + i;
    __syncthreads();
    double acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            double aa = a;
            if (VAR == 1) aa = lds[lane * 4 + ((it + c) & 3)] - (double)((it + c) & 3); // = a, through LDS
            acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(aa, b, acc[c], 0, 0, 0);
            if (VAR == 2) {
                asm volatile("v_add_f64 %0, %0, 1.0\n\tv_add_f64 %0, %0, -1.0" : "+v"(a)); // VALU (f64) writes of the A operand
            }
            if (VAR >= 10 && VAR < 30) { // the same behind VAR - 10 extra wait states
                asm volatile("s_nop %1\n\tv_add_f64 %0, %0, 1.0\n\tv_add_f64 %0, %0, -1.0" : "+v"(a) : "n"(VAR - 10 < 16 ? VAR - 10 : 15));
            }
            if (VAR == 3) { // 32-bit VALU writes (xor of the low mantissa word, twice)
                asm volatile("v_xor_b32 %0, 1, %0\n\tv_xor_b32 %0, 1, %0" : "+v"(reinterpret_cast<int *>(&a)[0]));
            }
            if (VAR == 4) { // a v_mov of the whole operand from a copy (what a register allocator's reuse looks like)
                double t = a;
                asm volatile("v_mov_b64 %0, 0\n\tv_mov_b64 %0, %1" : "+v"(a) : "v"(t));
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c] * (c + 1);
    out[(size_t)blockIdx.x * 64 + lane] = s;
}

template <int CH, int VAR>
static void run(int blocks, int iters, const char *what)
{
    double *d;
    hipMalloc(&d, sizeof(double) * 64 * blocks);
    std::vector<double> h(64 * (size_t)blocks), ref(64 * (size_t)blocks);
    // reference: one wavefront per SIMD at most (256 CUs x 4 SIMDs): launch in slices of 1024 blocks, serialised
    for (int b0 = 0; b0 < blocks; b0 += 512) {
        // a grid of 512 single-wave blocks never puts two on one SIMD (1024 SIMDs)
    }
    hipLaunchKernelGGL((chain<CH, VAR>), dim3(blocks), dim3(64), 0, 0, iters, d, 0);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // exact reference on the host: acc = iters * (A^T B) per slot; reproduce the layout algebra by running ONE block alone
    double *d1;
    hipMalloc(&d1, sizeof(double) * 64);
    long bad = 0;
    double worst = 0.0;
    for (int b = 0; b < blocks; b += blocks / 8 > 0 ? blocks / 8 : 1) { // spot blocks, each recomputed alone on the device
        // blockIdx enters `a`: launch a grid of b+1 blocks would re-create contention; instead run grid 1 with an offset kernel
    }
    // simpler: every block with the same (blockIdx % 5) has identical inputs -> identical exact outputs
    for (int b = 5; b < blocks; ++b)
        for (int l = 0; l < 64; ++l) {
            const double x = h[(size_t)b * 64 + l], y = h[(size_t)(b % 5) * 64 + l];
            if (x != y) {
                ++bad;
                const double e = fabs(x - y) / (fabs(y) + 1e-300);
                if (e > worst) worst = e;
            }
        }
    printf("%-46s blocks %5d iters %d: mismatching lanes %ld of %ld, worst rel %.3e\n", what, blocks, iters, bad, (long)(blocks - 5) * 64, worst);
    hipFree(d);
    hipFree(d1);
}

int main()
{
    for (int blocks : {1024, 4096}) {
        run<1, 0>(blocks, 2000, "1 chain, back to back");
        run<3, 0>(blocks, 2000, "3 chains, back to back");
        run<9, 0>(blocks, 2000, "9 chains, back to back");
        run<3, 1>(blocks, 2000, "3 chains, A operand through LDS");
        run<3, 2>(blocks, 2000, "3 chains, A rewritten by v_add_f64 right behind");
        run<3, 3>(blocks, 2000, "3 chains, A low word xor-ed twice (32-bit VALU)");
        run<3, 4>(blocks, 2000, "3 chains, A zeroed and restored by v_mov_b64");
        run<3, 10>(blocks, 2000, "3 chains, v_add_f64 behind s_nop 0");
        run<3, 11>(blocks, 2000, "3 chains, v_add_f64 behind s_nop 1");
        run<3, 13>(blocks, 2000, "3 chains, v_add_f64 behind s_nop 3");
        run<3, 17>(blocks, 2000, "3 chains, v_add_f64 behind s_nop 7");
        run<3, 25>(blocks, 2000, "3 chains, v_add_f64 behind s_nop 15");
    }
    return 0;
}
