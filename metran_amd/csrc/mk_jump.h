// mk_jump.h -- run-time column access to a lane's register row through jump tables, and wavefront sums: shared by the wide
// kernels (mk_split.hip: split filter, wide adjoint; mk_dk.hip: inverse-free smoother).
#pragma once
#include "mk_prims.h"

namespace mk {



// dst (lanes of MASK only) = p[j - BASE] for a wavefront-uniform j in [BASE, BASE + 16): a jump table in place of the
// decision tree hipcc builds for `switch (j)` (five levels of compare / structurised "Flow" blocks, ~40 scalar
// instructions and ~10 branches per pick; there are 64 / H picks per scalar update).  Every case is 8 bytes
// (v_mov_b64 + s_branch), the target is computed from the program counter; j outside the range falls through.
template <int BASE, unsigned long long MASK>
__device__ __forceinline__ void pick16(double &dst, int j, double p0, double p1, double p2, double p3, double p4, double p5,
                                       double p6, double p7, double p8, double p9, double p10, double p11, double p12,
                                       double p13, double p14, double p15)
{
    int t;
    unsigned long long saved;
    asm volatile("s_sub_i32 %[t], %[j], %[base]\n\t"
                 "s_cmp_lt_u32 %[t], 16\n\t"
                 "s_cbranch_scc0 .Lpick_end_%=\n\t"
                 "s_lshl_b32 %[t], %[t], 3\n\t"
                 "s_add_u32 %[t], %[t], 12\n\t"
                 "s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b32 exec_lo, %[mlo]\n\t"
                 "s_mov_b32 exec_hi, %[mhi]\n\t"
                 "s_getpc_b64 vcc\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, %[t]\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_setpc_b64 vcc\n\t"
                 "v_mov_b64 %[d], %[p0]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p1]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p2]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p3]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p4]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p5]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p6]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p7]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p8]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p9]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p10]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p11]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p12]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p13]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p14]\n\ts_branch .Lpick_done_%=\n\t"
                 "v_mov_b64 %[d], %[p15]\n\t"
                 ".Lpick_done_%=:\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 ".Lpick_end_%=:"
                 : [d] "+v"(dst), [t] "=&s"(t), [sv] "=&s"(saved)
                 : [j] "s"(j), [base] "n"(BASE), [mlo] "n"((int)(unsigned)(MASK & 0xffffffffull)), [mhi] "n"((int)(unsigned)(MASK >> 32)),
                   [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4), [p5] "v"(p5), [p6] "v"(p6), [p7] "v"(p7),
                   [p8] "v"(p8), [p9] "v"(p9), [p10] "v"(p10), [p11] "v"(p11), [p12] "v"(p12), [p13] "v"(p13), [p14] "v"(p14),
                   [p15] "v"(p15)
                 : "vcc", "scc");
}
// element j (wavefront-uniform, < N) of the lane's row, written to the lanes of MASK
template <int N, int n, unsigned long long MASK>
__device__ __forceinline__ void pick_column(double &dst, int j, const double (&P)[n])
{
#define MK_PE(i) P[(i) < N ? (i) : N - 1]
    pick16<0, MASK>(dst, j, MK_PE(0), MK_PE(1), MK_PE(2), MK_PE(3), MK_PE(4), MK_PE(5), MK_PE(6), MK_PE(7), MK_PE(8), MK_PE(9),
                    MK_PE(10), MK_PE(11), MK_PE(12), MK_PE(13), MK_PE(14), MK_PE(15));
    if constexpr (N > 16)
        pick16<16, MASK>(dst, j, MK_PE(16), MK_PE(17), MK_PE(18), MK_PE(19), MK_PE(20), MK_PE(21), MK_PE(22), MK_PE(23), MK_PE(24),
                         MK_PE(25), MK_PE(26), MK_PE(27), MK_PE(28), MK_PE(29), MK_PE(30), MK_PE(31));
#undef MK_PE
}

template <int CTRL>
__device__ __forceinline__ double dpp_perm_f64(double v)
{
    // (bound_ctrl: every lane of these permutations has a source, so no "old" value has to be materialised)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, result in every lane: xor-1, xor-2 inside the quads, half-row and row mirrors, then the four
// row sums through readlanes
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_perm_f64<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_perm_f64<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_perm_f64<0x141>(v); // row_half_mirror
    v += dpp_perm_f64<0x140>(v); // row_mirror
    using G64 = Group<64>;
    return (G64::bcast<0>(v) + G64::bcast<16>(v)) + (G64::bcast<32>(v) + G64::bcast<48>(v));
}

// Sum over the 64 lanes, result in every lane, on the matrix pipe: v_mfma_f64_4x4x4_4b_f64 contracts over the four DPP rows
// (lane = j + 4 b + 16 k: element j of block b in row k), so with a matrix of ones as the other operand one instruction adds
// the four rows for every lane position and a second one (operand roles swapped: the first result comes back transposed
// between quad position and row) completes the sum over each 16-lane set {quad b of every row}; the four quad totals of a
// row are then two rotate-and-add steps.  2 MFMA (16 cycles of the f64 pipe each) + 6 vector instructions against the 8
// permutes, 8 readlanes and 11 additions of wave_sum_f64.
__device__ __forceinline__ double wave_sum_mfma(double v)
{
    double t = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, t, 0.0, 0, 0, 0);
    t += dpp_perm_f64<0x124>(t); // row_ror:4
    t += dpp_perm_f64<0x128>(t); // row_ror:8
    return t;
}

// p[j - BASE] += h for a wavefront-uniform j in [BASE, BASE + 16) (jump table, 12 bytes per case: v_add_f64 + s_branch)
template <int BASE>
__device__ __forceinline__ void add16(int j, double h, double &p0, double &p1, double &p2, double &p3, double &p4, double &p5,
                                      double &p6, double &p7, double &p8, double &p9, double &p10, double &p11, double &p12,
                                      double &p13, double &p14, double &p15)
{
    int t;
    asm volatile("s_sub_i32 %[t], %[j], %[base]\n\t"
                 "s_cmp_lt_u32 %[t], 16\n\t"
                 "s_cbranch_scc0 .Ladd_end_%=\n\t"
                 "s_mul_i32 %[t], %[t], 12\n\t"
                 "s_add_u32 %[t], %[t], 12\n\t"
                 "s_getpc_b64 vcc\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, %[t]\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_setpc_b64 vcc\n\t"
                 "v_add_f64 %[p0], %[p0], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p1], %[p1], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p2], %[p2], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p3], %[p3], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p4], %[p4], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p5], %[p5], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p6], %[p6], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p7], %[p7], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p8], %[p8], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p9], %[p9], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p10], %[p10], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p11], %[p11], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p12], %[p12], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p13], %[p13], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p14], %[p14], %[h]\n\ts_branch .Ladd_end_%=\n\t"
                 "v_add_f64 %[p15], %[p15], %[h]\n\t"
                 ".Ladd_end_%=:"
                 : [t] "=&s"(t), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4), [p5] "+v"(p5),
                   [p6] "+v"(p6), [p7] "+v"(p7), [p8] "+v"(p8), [p9] "+v"(p9), [p10] "+v"(p10), [p11] "+v"(p11),
                   [p12] "+v"(p12), [p13] "+v"(p13), [p14] "+v"(p14), [p15] "+v"(p15)
                 : [j] "s"(j), [base] "n"(BASE), [h] "v"(h)
                 : "vcc", "scc");
}
template <int N, int n>
__device__ __forceinline__ void add_column(int j, double h, double (&P)[n])
{
    static_assert(N <= 64 && n >= 16, "wide models");
    double dump = 0.0; // cases beyond N - 1 are never selected (j < N)
#define MK_AE(i) ((i) < N ? P[(i) < N ? (i) : 0] : dump)
    sfor<0, (N + 15) / 16>(MK_LAMBDA(bb) {
        constexpr int B0 = 16 * decltype(bb)::value;
        add16<B0>(j, h, MK_AE(B0 + 0), MK_AE(B0 + 1), MK_AE(B0 + 2), MK_AE(B0 + 3), MK_AE(B0 + 4), MK_AE(B0 + 5), MK_AE(B0 + 6),
                  MK_AE(B0 + 7), MK_AE(B0 + 8), MK_AE(B0 + 9), MK_AE(B0 + 10), MK_AE(B0 + 11), MK_AE(B0 + 12), MK_AE(B0 + 13),
                  MK_AE(B0 + 14), MK_AE(B0 + 15));
    });
#undef MK_AE
}
// all lanes: dst = P[j], wavefront-uniform j < N <= 64
template <int N, int n>
__device__ __forceinline__ void pick_column_all(double &dst, int j, const double (&P)[n])
{
#define MK_PE(i) P[(i) < N ? (i) : N - 1]
    sfor<0, (N + 15) / 16>(MK_LAMBDA(bb) {
        constexpr int B0 = 16 * decltype(bb)::value;
        pick16<B0, ~0ull>(dst, j, MK_PE(B0 + 0), MK_PE(B0 + 1), MK_PE(B0 + 2), MK_PE(B0 + 3), MK_PE(B0 + 4), MK_PE(B0 + 5),
                          MK_PE(B0 + 6), MK_PE(B0 + 7), MK_PE(B0 + 8), MK_PE(B0 + 9), MK_PE(B0 + 10), MK_PE(B0 + 11),
                          MK_PE(B0 + 12), MK_PE(B0 + 13), MK_PE(B0 + 14), MK_PE(B0 + 15));
    });
#undef MK_PE
}


} // namespace mk
