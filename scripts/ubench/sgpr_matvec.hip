// Dev microbenchmark (round 6): how should the 36-term product w_a = sum_c N[a][c] x_c of smoother_dk_kernel get its broadcast
// operand x_c?  (a) v_fmac_f64_dpp row_newbcast from three DPP-replicated registers read from LDS (what the kernel does),
// (b) plain v_fmac_f64 with x_c in an SGPR pair, the vector fetched by scalar loads (s_load_dwordx16) straight from the tape
// in global memory, (c) plain v_fmac_f64 on VGPR operands (no broadcast: the floor).  One model per wavefront, two
// wavefronts per SIMD, every SIMD busy (4096 wavefronts), 36 row registers, two accumulators -- the kernel's shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define DPPM " row_mask:0xf bank_mask:0xf"
typedef double v8d __attribute__((ext_vector_type(8)));
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k(const double *tape, double *out, int entries, long stride)
{
    __shared__ double lds[64 * 4];
    const int lane = threadIdx.x;
    double Nr[36];
#pragma unroll
    for (int c = 0; c < 36; ++c) Nr[c] = 1e-3 * (lane + c);
    double tot = 0.0;
    const double *base = tape + (long)blockIdx.x * stride;
    for (int c = 0; c < 40; ++c) lds[c] = 1e-6 * c;
    __syncthreads();
    for (int e = 0; e < entries; ++e) {
        double acc0 = 0.0, acc1 = 0.0;
        const double *p = base + (long)e * 40; // 36 doubles + 4 scalars per entry (320 B)
        if constexpr (MODE == 0) {
            // DPP-replicated operands from LDS: lane 16q+i holds x[16m+i]
            const double xb0 = lds[(lane & 15) + (e & 1)], xb1 = lds[16 + (lane & 15) + (e & 1)], xf = lds[32 + (lane & 3) + (e & 1)];
#define F(a, x, v, j) "v_fmac_f64_dpp " a ", " x ", " v " row_newbcast:" #j DPPM "\n\t"
            asm volatile("s_nop 1\n\t" F("%0", "%2", "%3", 0) F("%1", "%2", "%4", 1) F("%0", "%2", "%5", 2) F("%1", "%2", "%6", 3) F("%0", "%2", "%7", 4)
                             F("%1", "%2", "%8", 5) F("%0", "%2", "%9", 6) F("%1", "%2", "%10", 7) F("%0", "%2", "%11", 8) F("%1", "%2", "%12", 9)
                                 F("%0", "%2", "%13", 10) F("%1", "%2", "%14", 11) F("%0", "%2", "%15", 12) F("%1", "%2", "%16", 13)
                                     F("%0", "%2", "%17", 14) F("%1", "%2", "%18", 15)
                         : "+v"(acc0), "+v"(acc1)
                         : "v"(xb0), "v"(Nr[0]), "v"(Nr[1]), "v"(Nr[2]), "v"(Nr[3]), "v"(Nr[4]), "v"(Nr[5]), "v"(Nr[6]), "v"(Nr[7]), "v"(Nr[8]),
                           "v"(Nr[9]), "v"(Nr[10]), "v"(Nr[11]), "v"(Nr[12]), "v"(Nr[13]), "v"(Nr[14]), "v"(Nr[15]));
            asm volatile("s_nop 1\n\t" F("%0", "%2", "%3", 0) F("%1", "%2", "%4", 1) F("%0", "%2", "%5", 2) F("%1", "%2", "%6", 3) F("%0", "%2", "%7", 4)
                             F("%1", "%2", "%8", 5) F("%0", "%2", "%9", 6) F("%1", "%2", "%10", 7) F("%0", "%2", "%11", 8) F("%1", "%2", "%12", 9)
                                 F("%0", "%2", "%13", 10) F("%1", "%2", "%14", 11) F("%0", "%2", "%15", 12) F("%1", "%2", "%16", 13)
                                     F("%0", "%2", "%17", 14) F("%1", "%2", "%18", 15)
                         : "+v"(acc0), "+v"(acc1)
                         : "v"(xb1), "v"(Nr[16]), "v"(Nr[17]), "v"(Nr[18]), "v"(Nr[19]), "v"(Nr[20]), "v"(Nr[21]), "v"(Nr[22]), "v"(Nr[23]),
                           "v"(Nr[24]), "v"(Nr[25]), "v"(Nr[26]), "v"(Nr[27]), "v"(Nr[28]), "v"(Nr[29]), "v"(Nr[30]), "v"(Nr[31]));
            asm volatile("s_nop 1\n\t" F("%0", "%2", "%3", 0) F("%1", "%2", "%4", 1) F("%0", "%2", "%5", 2) F("%1", "%2", "%6", 3)
                         : "+v"(acc0), "+v"(acc1)
                         : "v"(xf), "v"(Nr[32]), "v"(Nr[33]), "v"(Nr[34]), "v"(Nr[35]));
#undef F
        } else if constexpr (MODE == 1) {
            // the entry's vector by scalar loads (uniform address: blockIdx and the loop counter), x_c as the SGPR operand
            const __attribute__((address_space(4))) v8d *sp = (const __attribute__((address_space(4))) v8d *)(uintptr_t)p;
            const v8d x0 = sp[0], x1 = sp[1], x2 = sp[2], x3 = sp[3];
            const v4d x4 = *(const __attribute__((address_space(4))) v4d *)(uintptr_t)(p + 32);
#define G(a, x, v) asm volatile("v_fmac_f64_e64 %0, %1, %2" : "+v"(a) : "s"(x), "v"(v));
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                G(c & 1 ? acc1 : acc0, x0[c], Nr[c])
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                G(c & 1 ? acc1 : acc0, x1[c], Nr[8 + c])
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                G(c & 1 ? acc1 : acc0, x2[c], Nr[16 + c])
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                G(c & 1 ? acc1 : acc0, x3[c], Nr[24 + c])
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                G(c & 1 ? acc1 : acc0, x4[c], Nr[32 + c])
            }
#undef G
        } else {
            const double xv = lds[(lane & 15) + (e & 1)];
#pragma unroll
            for (int c = 0; c < 36; ++c) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(c & 1 ? acc1 : acc0) : "v"(xv), "v"(Nr[c]));
        }
        const double w = acc0 + acc1;
        tot += w;
        Nr[e % 36 == 0 ? 0 : 1] -= 1e-9 * w; // (keeps the rows live and dependent on the product, like the column update)
    }
    out[(long)blockIdx.x * 64 + lane] = tot + Nr[0] + Nr[1];
}

template <int MODE>
void run(const char *name, const double *tape, double *out, int waves, long stride)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int entries = 20000;
    k<MODE><<<waves, 64>>>(tape, out, 200, stride);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<waves, 64>>>(tape, out, entries, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double rounds = waves / 2048.0; // two wavefronts per SIMD, 1024 SIMDs
    printf("%-34s waves=%5d  %8.3f ms  -> %7.1f ns of SIMD time per entry (36 multiply-adds), %.2f ns per multiply-add\n", name, waves, ms,
           ms * 1e6 / entries / rounds / 2.0, ms * 1e6 / entries / rounds / 2.0 / 36);
}

int main()
{
    const long stride = 20200L * 40;
    double *tape, *out;
    const int maxw = 4096;
    hipMalloc(&tape, sizeof(double) * stride * maxw);
    hipMemset(tape, 0, sizeof(double) * stride * maxw);
    hipMalloc(&out, sizeof(double) * 64 * maxw);
    for (int waves : {2048, 4096}) {
        run<0>("v_fmac_f64_dpp, operands from LDS", tape, out, waves, stride);
        run<1>("v_fmac_f64 sgpr, operands s_load", tape, out, waves, stride);
        run<2>("v_fmac_f64 vgpr (no broadcast)", tape, out, waves, stride);
    }
    return 0;
}
