// mk_prims.h -- device primitives shared by the kernel translation units (mk_kernels.hip, mk_wide.hip):
// compile-time loops, the cross-lane Group<G> primitives (fused DPP broadcast-FMA for G = 16, readlane for G = 64),
// row / column-run / record addressing helpers, the fused projection epilogue, reciprocal and logarithm kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mk_internal.h"

namespace mk {

// ---------------------------------------------------------------- compile-time loops
template <int I, int E, class F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, E>(static_cast<F &&>(f));
    }
}
template <int I, int E, class F>
__device__ __forceinline__ void sfor_down(F &&f) // I = E-1 .. 0 handled as (E-1-I)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, E - 1 - I>{});
        sfor_down<I + 1, E>(static_cast<F &&>(f));
    }
}
#define MK_LAMBDA(arg) [&](auto arg) __attribute__((always_inline))

// ---------------------------------------------------------------- cross-lane primitives
// All cross-lane traffic of the hot kernels goes through three primitives:
//   bcast<J>(v)                          value of lane J of the group, in every lane
//   axpy_lane<J,C0,C1>(acc, src, mul)    acc[c] (+|-)= bcast<J>(src[c]) * mul      c in [C0,C1)
//   axpy_col<C0,C1>(acc, src, mul)       acc[c] (+|-)= bcast<c>(src)    * mul      c in [C0,C1)
//
// G = 16 (four models per wavefront, one per 16-lane DPP row): gfx950 has DPP64
// `row_newbcast` on v_mov_b64 and v_fmac_f64, so a broadcast-multiply-accumulate is ONE
// instruction (v_fmac_f64_dpp; negation is a free source modifier) with no LDS and no extra
// move.  hipcc has no builtin for the fused form, so it is emitted as inline asm.
// HAZARD: "VALU writes a VGPR -> DPP reads it as src0" needs 2 wait states and neither the
// assembler nor hipcc's hazard recogniser pads it around inline asm.  A lone wavefront issues one
// instruction per ~5 cycles and pays ~9 for an `s_nop 1`, so no blanket nops are emitted;
// instead scripts/check_dpp_hazards.py statically verifies the generated assembly at build time
// (see __graft_entry__.build) and `dpp_guard()` is placed where it reports a producer too close.
// G = 64 (one model per wavefront): v_readlane -> SGPR pair, plain v_fma with a scalar operand.
template <int G>
struct Group;

#define MK_DPPMASK " row_mask:0xf bank_mask:0xf"

// Two wait states between the producers of `arr` and the DPP reads that follow: the empty asm
// statements pin every element (its producer cannot be scheduled below them), the nop follows.
template <int n>
__device__ __forceinline__ void dpp_guard(double (&arr)[n])
{
#pragma unroll
    for (int c = 0; c < n; ++c) asm volatile("" : "+v"(arr[c]));
    asm volatile("s_nop 1");
}
__device__ __forceinline__ void dpp_pin(double &v) { asm volatile("" : "+v"(v)); }
// two wait states before DPP reads of one or two freshly produced scalars
__device__ __forceinline__ void dpp_guard1(double &u, double &v)
{
    asm volatile("s_nop 1" : "+v"(u), "+v"(v));
}

template <>
struct Group<16> {
    template <int J>
    static __device__ __forceinline__ double bcast(double v)
    {
        return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, true); // v_mov_b64_dpp row_newbcast:J
    }
    // acc += bcast<J>(src) * mul   /   acc -= bcast<J>(src) * mul
    // GUARD (here and below): two wait states INSIDE the statement, ahead of its first DPP read.  The compiler does not see the
    // DPP reads of an asm statement; a kernel that spills (AGPR reloads are VALU writes placed right before the use) asks for it.
    template <int J, bool NEG = false, bool GUARD = false>
    static __device__ __forceinline__ void fmac(double &acc, double src, double mul)
    {
        if constexpr (GUARD)
            asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3" MK_DPPMASK : "+v"(acc) : "v"(src), "v"(NEG ? -mul : mul), "n"(J));
        else if constexpr (NEG)
            asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3" MK_DPPMASK : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
        else
            asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3" MK_DPPMASK : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
    }
    // two broadcast-multiply-adds in ONE asm statement (hipcc pads every asm statement that follows another with an
    // `s_nop 0`; neither instruction reads what the other writes)
    template <int J0, int J1, bool NEG = false>
    static __device__ __forceinline__ void fmac2(double &acc0, double src0, double mul0, double &acc1, double src1, double mul1)
    {
        if constexpr (NEG)
            asm volatile("v_fmac_f64_dpp %0, %2, -%3 row_newbcast:%6" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %4, -%5 row_newbcast:%7" MK_DPPMASK
                         : "+v"(acc0), "+v"(acc1) : "v"(src0), "v"(mul0), "v"(src1), "v"(mul1), "n"(J0), "n"(J1));
        else
            asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:%6" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:%7" MK_DPPMASK
                         : "+v"(acc0), "+v"(acc1) : "v"(src0), "v"(mul0), "v"(src1), "v"(mul1), "n"(J0), "n"(J1));
    }
    // four in one statement, two accumulators alternating (acc0, acc1, acc0, acc1): the dot-form sweeps of the wide smoother
    template <int J0, int J1, int J2, int J3, bool NEG = false, bool GUARD = false>
    static __device__ __forceinline__ void fmac4(double &acc0, double &acc1, double s0, double m0, double s1, double m1, double s2,
                                                 double m2, double s3, double m3)
    {
        static_assert(!(GUARD && NEG), "guarded form: positive only");
        if constexpr (GUARD)
            asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:%10" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:%11" MK_DPPMASK
                         "\n\tv_fmac_f64_dpp %0, %6, %7 row_newbcast:%12" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:%13" MK_DPPMASK
                         : "+v"(acc0), "+v"(acc1)
                         : "v"(s0), "v"(m0), "v"(s1), "v"(m1), "v"(s2), "v"(m2), "v"(s3), "v"(m3), "n"(J0), "n"(J1), "n"(J2), "n"(J3));
        else if constexpr (NEG)
            asm volatile("v_fmac_f64_dpp %0, %2, -%3 row_newbcast:%10" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %4, -%5 row_newbcast:%11" MK_DPPMASK
                         "\n\tv_fmac_f64_dpp %0, %6, -%7 row_newbcast:%12" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %8, -%9 row_newbcast:%13" MK_DPPMASK
                         : "+v"(acc0), "+v"(acc1)
                         : "v"(s0), "v"(m0), "v"(s1), "v"(m1), "v"(s2), "v"(m2), "v"(s3), "v"(m3), "n"(J0), "n"(J1), "n"(J2), "n"(J3));
        else
            asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:%10" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:%11" MK_DPPMASK
                         "\n\tv_fmac_f64_dpp %0, %6, %7 row_newbcast:%12" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:%13" MK_DPPMASK
                         : "+v"(acc0), "+v"(acc1)
                         : "v"(s0), "v"(m0), "v"(s1), "v"(m1), "v"(s2), "v"(m2), "v"(s3), "v"(m3), "n"(J0), "n"(J1), "n"(J2), "n"(J3));
    }
    // sixteen in ONE statement (19 operands): the sixteen lanes of one replicated source register against sixteen
    // multipliers, two accumulators alternating -- a 16-column slice of a row-per-lane matrix-vector product (mk_dk.hip)
    template <bool GUARD = false>
    static __device__ __forceinline__ void fmac16(double &acc0, double &acc1, double src, double m0, double m1, double m2, double m3,
                                                  double m4, double m5, double m6, double m7, double m8, double m9, double m10,
                                                  double m11, double m12, double m13, double m14, double m15)
    {
        if constexpr (GUARD)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:0" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %8 row_newbcast:5" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %9 row_newbcast:6" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %10 row_newbcast:7" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %11 row_newbcast:8" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %12 row_newbcast:9" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %13 row_newbcast:10" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %14 row_newbcast:11" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %15 row_newbcast:12" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %16 row_newbcast:13" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %17 row_newbcast:14" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %18 row_newbcast:15" MK_DPPMASK
                     : "+v"(acc0), "+v"(acc1)
                     : "v"(src), "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(m6), "v"(m7), "v"(m8), "v"(m9), "v"(m10),
                       "v"(m11), "v"(m12), "v"(m13), "v"(m14), "v"(m15));
        else
        asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:0" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %8 row_newbcast:5" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %9 row_newbcast:6" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %10 row_newbcast:7" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %11 row_newbcast:8" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %12 row_newbcast:9" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %13 row_newbcast:10" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %14 row_newbcast:11" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %15 row_newbcast:12" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %16 row_newbcast:13" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %0, %2, %17 row_newbcast:14" MK_DPPMASK "\n\t"
                     "v_fmac_f64_dpp %1, %2, %18 row_newbcast:15" MK_DPPMASK
                     : "+v"(acc0), "+v"(acc1)
                     : "v"(src), "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(m6), "v"(m7), "v"(m8), "v"(m9), "v"(m10),
                       "v"(m11), "v"(m12), "v"(m13), "v"(m14), "v"(m15));
    }
    // four independent accumulators, one broadcast lane, one multiplier: acc_i += bcast<J>(s_i) * mul
    template <int J, bool GUARD = false>
    static __device__ __forceinline__ void fmac4x(double &a0, double &a1, double &a2, double &a3, double s0, double s1, double s2,
                                                  double s3, double mul)
    {
        if constexpr (GUARD)
            asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %4, %8 row_newbcast:%9" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %5, %8 row_newbcast:%9" MK_DPPMASK
                         "\n\tv_fmac_f64_dpp %2, %6, %8 row_newbcast:%9" MK_DPPMASK "\n\tv_fmac_f64_dpp %3, %7, %8 row_newbcast:%9" MK_DPPMASK
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                         : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(mul), "n"(J));
        else
        asm volatile("v_fmac_f64_dpp %0, %4, %8 row_newbcast:%9" MK_DPPMASK "\n\tv_fmac_f64_dpp %1, %5, %8 row_newbcast:%9" MK_DPPMASK
                     "\n\tv_fmac_f64_dpp %2, %6, %8 row_newbcast:%9" MK_DPPMASK "\n\tv_fmac_f64_dpp %3, %7, %8 row_newbcast:%9" MK_DPPMASK
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                     : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(mul), "n"(J));
    }
    template <int J, int C0, int C1, bool NEG, int n>
    static __device__ __forceinline__ void axpy_lane(double (&acc)[n], const double (&src)[n], double mul)
    {
        sfor<C0, C1>(MK_LAMBDA(c) { fmac<J, NEG>(acc[decltype(c)::value], src[decltype(c)::value], mul); });
    }
    template <int C0, int C1, bool NEG, int n>
    static __device__ __forceinline__ void axpy_col(double (&acc)[n], double src, double mul)
    {
        sfor<C0, C1>(MK_LAMBDA(c) { fmac<decltype(c)::value, NEG>(acc[decltype(c)::value], src, mul); });
    }
    // per-group mask of lanes whose predicate holds (bit l = lane l of this group)
    static __device__ __forceinline__ unsigned group_bits(unsigned long long m)
    {
        return (unsigned)(m >> ((threadIdx.x & 63) & ~15)) & 0xffffu;
    }
    // ballot pattern with bits [0,N) set in every group
    static constexpr unsigned long long full_mask(int N)
    {
        const unsigned long long g = (1ull << N) - 1ull;
        return g | (g << 16) | (g << 32) | (g << 48);
    }
};

template <>
struct Group<64> {
    template <int J>
    static __device__ __forceinline__ double bcast(double v)
    {
        int lo = __builtin_amdgcn_readlane(__double2loint(v), J);
        int hi = __builtin_amdgcn_readlane(__double2hiint(v), J);
        return __hiloint2double(hi, lo);
    }
    template <int J, bool NEG = false>
    static __device__ __forceinline__ void fmac(double &acc, double src, double mul)
    {
        acc = fma(NEG ? -bcast<J>(src) : bcast<J>(src), mul, acc);
    }
    // A broadcast is two v_readlane_b32 into an SGPR pair that the FMA reads as a constant, with 2 wait
    // states between them: issued one after the other every FMA pays an `s_nop 1` (~9 cycles for a lone
    // wavefront).  The sweeps therefore run in batches of BATCH elements, all broadcasts of a batch first
    // (source order is what the scheduler keeps), then the FMAs; the scheduling fence after each batch
    // bounds the live SGPRs at 2*BATCH -- left alone the compiler hoists every readlane of a sweep to
    // its top and spills the SGPRs into VGPR lanes and those VGPRs into scratch (measured at n = 36).
    static constexpr int BATCH = 6;
    template <int J, int C0, int C1, bool NEG, int n>
    static __device__ __forceinline__ void axpy_lane(double (&acc)[n], const double (&src)[n], double mul)
    {
        if constexpr (C0 < C1) {
            constexpr int CE = C0 + BATCH < C1 ? C0 + BATCH : C1;
            double b[CE - C0];
            sfor<C0, CE>(MK_LAMBDA(c) { b[decltype(c)::value - C0] = bcast<J>(src[decltype(c)::value]); });
            sfor<C0, CE>(MK_LAMBDA(c) {
                constexpr int cc = decltype(c)::value;
                acc[cc] = fma(NEG ? -b[cc - C0] : b[cc - C0], mul, acc[cc]);
            });
            __builtin_amdgcn_sched_barrier(0);
            axpy_lane<J, CE, C1, NEG, n>(acc, src, mul);
        }
    }
    template <int C0, int C1, bool NEG, int n>
    static __device__ __forceinline__ void axpy_col(double (&acc)[n], double src, double mul)
    {
        if constexpr (C0 < C1) {
            constexpr int CE = C0 + BATCH < C1 ? C0 + BATCH : C1;
            double b[CE - C0];
            sfor<C0, CE>(MK_LAMBDA(c) { b[decltype(c)::value - C0] = bcast<decltype(c)::value>(src); });
            sfor<C0, CE>(MK_LAMBDA(c) {
                constexpr int cc = decltype(c)::value;
                acc[cc] = fma(NEG ? -b[cc - C0] : b[cc - C0], mul, acc[cc]);
            });
            __builtin_amdgcn_sched_barrier(0);
            axpy_col<CE, C1, NEG, n>(acc, src, mul);
        }
    }
    static __device__ __forceinline__ unsigned long long group_bits(unsigned long long m) { return m; }
    static constexpr unsigned long long full_mask(int N) { return N >= 64 ? ~0ull : (1ull << N) - 1ull; }
};

#include "mk_sweeps.h" // Sweeps<n>: whole sweeps as single asm statements (generated, n <= 10)

// value of `v` in lane `l`, l wavefront-uniform at run time (v_readlane_b32 with a scalar lane select)
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

typedef double v2d __attribute__((ext_vector_type(2))); // native <2 x double> (stays in VGPRs)

template <int n>
__device__ __forceinline__ void store_row(double *dst, const double (&row)[n])
{
    if constexpr (n % 2 == 0) {
        v2d *d2 = reinterpret_cast<v2d *>(dst); // row offset is a multiple of 16 B
#pragma unroll
        for (int c = 0; c < n / 2; ++c) d2[c] = v2d{row[2 * c], row[2 * c + 1]};
    } else {
#pragma unroll
        for (int c = 0; c < n; ++c) dst[c] = row[c];
    }
}

template <int n>
__device__ __forceinline__ void load_row(const double *src, double (&row)[n])
{
    if constexpr (n % 2 == 0) {
        const v2d *s2 = reinterpret_cast<const v2d *>(src);
#pragma unroll
        for (int c = 0; c < n / 2; ++c) {
            const v2d v = s2[c];
            row[2 * c] = v.x;
            row[2 * c + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int c = 0; c < n; ++c) row[c] = src[c];
    }
}

// single v_min_f64 (fmin() adds a canonicalising v_max); NaN operands are ignored, which is fine:
// a NaN variance propagates into the outputs by itself
__device__ __forceinline__ double min_f64(double a, double b)
{
    double o;
    asm("v_min_f64 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}

// 1/x to ~1 ulp: v_rcp_f64 (measured relative error < 2^-25 on gfx950, tests/test_hip_parity.py::
// test_rcp_accuracy) + one cubically convergent step (3 dependent FMAs instead of the 4 of two Newton
// steps; no scaling/fix-up: x is an O(1) variance here, never denormal).
__device__ __forceinline__ double rcp_nr(double x)
{
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r0, 1.0);   // relative error of r0
    const double p = fma(e, e, e);        // e + e^2
    return fma(r0, p, r0);                // r0 (1 + e + e^2): error e^3
}

// ---------------------------------------------------------------- state I/O: symmetric column runs
// Lane r holds ROW r of a covariance; its HBM image is row-major, i.e. lane r's data is n doubles at
// stride 1 -- stored directly, every store instruction would touch ~n scattered pieces per model
// (measured in v1: store-issue bound at 2.5 TB/s).  The covariances are SYMMETRIC, so the kernels
// store the transpose instead: for column index c = 0..n-1 every lane writes its element P[r][c] to
// position (c, r), i.e. one instruction writes a contiguous n*8-byte run per model and n instructions
// cover the block.  What lands in HBM is (P)^T = P up to the rounding-level asymmetry of the rank-1
// updates (< 1e-14 relative).  No LDS staging, no waits, and the smoother re-reading "row r" as
// column r gets back exactly the filter's own row r.  Lanes >= n (and the groups of a partial last
// workgroup) are exact REPLICAS of lane n-1 (of the last model): same inputs, same instruction
// stream, identical bytes to identical addresses -- no store in the hot loops needs an exec mask.
//
// PACKED RECORDS (fast path, mk_outputs.record_stride): measured on MI355X
// (scripts/ubench/store_pattern2.hip), 6.5 GB of block stores per launch are almost free next to a
// busy VALU, but 80-byte mean vectors and 8-byte sigma/detf scalars written as SEPARATE arrays cost
// ~8x more per byte: they leave every cache line partially written when it is evicted, and
// partial-line writes throttle HBM.  A record keeps one (model, step) moment set together,
//     [ mean(n) | covariance(n*n) | sigma, detf (filtered set only) | zero pad ]     RS doubles,
// with RS*8 a multiple of 128 bytes; the reference-shaped arrays are strided VIEWS of the record
// arrays on the host.
__device__ __forceinline__ void wave_lds_sync()
{
    // LDS-only ("local") fences: a generic fence would also pin private arrays to scratch memory
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// LDS stores executed by the lanes of a COMPILE-TIME mask only: exec is swapped around the store inside one asm block
// (scalar moves with literal masks; an `if (lane ...)` costs a compare on the vector ALU and, worse, splits the basic block -- hipcc
// then sinks whole dependency chains across the branch and spills what they keep alive).  The stores are untracked by
// the compiler's lgkmcnt bookkeeping, which stays correct: LDS operations of a wavefront complete in order, so an
// extra operation in the queue can only make a counted wait more conservative.
typedef __attribute__((address_space(3))) double lds_double;
template <unsigned long long MASK, int BYTE_OFFSET>
__device__ __forceinline__ void lds_store_masked(double *p, double v)
{
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, %3\n\ts_mov_b32 exec_hi, %4\n\tds_write_b64 %1, %2 offset:%5\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "v"((unsigned)(unsigned long)(lds_double *)p), "v"(v), "n"((int)(unsigned)(MASK & 0xffffffffull)),
                   "n"((int)(unsigned)(MASK >> 32)), "n"(BYTE_OFFSET)
                 : "memory");
}
template <unsigned long long MASK, int BYTE_OFFSET>
__device__ __forceinline__ void lds_store_masked(double *p, v2d v)
{
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, %3\n\ts_mov_b32 exec_hi, %4\n\tds_write_b128 %1, %2 offset:%5\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "v"((unsigned)(unsigned long)(lds_double *)p), "v"(v), "n"((int)(unsigned)(MASK & 0xffffffffull)),
                   "n"((int)(unsigned)(MASK >> 32)), "n"(BYTE_OFFSET)
                 : "memory");
}

// offset of row c of a lower triangle (diagonal excluded) packed by rows, every row padded to an even count
constexpr int tri_off(int c)
{
    int s = 0;
    for (int i = 0; i < c; ++i) s += (i + 1) & ~1;
    return s;
}

constexpr int record_payload(int n) { return n + n * n; }
constexpr int record_stride_c(int n) { return ((record_payload(n) + 2 + 15) / 16) * 16; }
// packed-symmetric records (mk_outputs.flags & MK_OUT_PACKED_SYM): mean + upper triangle by rows
constexpr int record_payload_sym(int n) { return n + n * (n + 1) / 2; }
constexpr int record_stride_sym_c(int n) { return ((record_payload_sym(n) + 2 + 15) / 16) * 16; }
// offset of row c of the packed upper triangle (elements (c, c) .. (c, n-1))
constexpr int sym_row_offset(int n, int c) { return c * n - c * (c - 1) / 2; }

// pointer pair addressing one moment set of one model: element r of the mean vector and element
// (0, r) of the covariance's column runs; `advance` moves both to the next (or previous) time step
struct MomentPtr {
    double *vec, *mat;
    long vstep, mstep;
    __device__ __forceinline__ void advance(long dir) // a null (skipped) output stays null
    {
        if (vec) vec += dir * vstep;
        if (mat) mat += dir * mstep;
    }
    __device__ __forceinline__ void advance_nn(long dir) // both pointers known to be non-null
    {
        vec += dir * vstep;
        mat += dir * mstep;
    }
};

// inst/r already clamped; rs > 0: packed records (V is the record array), else dense arrays V [.,n], M [.,n,n]
template <int n>
__device__ __forceinline__ MomentPtr moment_ptr(double *V, double *M, long blk, long ts, long rs, int r)
{
    MomentPtr p;
    if (rs > 0) {
        p.vec = V ? V + blk * rs + r : nullptr; // projection-only smoothing has no smoothed record array
        p.mat = V ? V + blk * rs + n + r : nullptr;
        p.vstep = p.mstep = ts * rs;
    } else {
        p.vec = V ? V + blk * n + r : nullptr;
        p.mat = M ? M + blk * n * n + r : nullptr;
        p.vstep = ts * n;
        p.mstep = ts * n * n;
    }
    return p;
}

template <int n>
__device__ __forceinline__ void store_cols(double *mat, const double (&row)[n])
{
#pragma unroll
    for (int c = 0; c < n; ++c) mat[c * n] = row[c]; // (c, r) <- P[r][c]: contiguous over the lanes
}
template <int n>
__device__ __forceinline__ void load_cols(const double *mat, double (&row)[n])
{
#pragma unroll
    for (int c = 0; c < n; ++c) row[c] = mat[c * n];
}

// packed-symmetric variants (base = first double of the record's packed upper triangle): lane r writes its
// P[r][c] to position (c, r) of the triangle for c <= r -- for a fixed c the lanes r >= c write one contiguous
// run -- and reads row r back as (c, r) for c <= r, (r, c) for c > r.
template <int n>
__device__ __forceinline__ void store_cols_sym(double *base, const double (&row)[n], int r)
{
#pragma unroll
    for (int c = 0; c < n; ++c)
        if (r >= c) base[sym_row_offset(n, c) + (r - c)] = row[c];
}
template <int n>
__device__ __forceinline__ void load_cols_sym(const double *base, double (&row)[n], int r)
{
    const int offr = r * n - r * (r - 1) / 2 - r; // sym_row_offset(n, r) - r: element (r, c) sits at offr + c
#pragma unroll
    for (int c = 0; c < n; ++c) row[c] = base[r >= c ? sym_row_offset(n, c) + (r - c) : offr + c];
}

// ---------------------------------------------------------------- fused projection epilogue
// SPKalmanFilter.simulate (kalmanfilter.py:569-603) for Metran's scaled observation matrix
// Z~ = diag(s) [I | loadings] (metran/metran.py:944-961), evaluated on the moments a kernel holds
// in registers: lane j < N returns  mean_j = s_j (x_j + sum_k g_jk x_{N+k}) + offset_j  and
// var_j = max(s_j^2 (P_jj + 2 sum_k g_jk P_j,N+k + sum_kl g_jk g_jl P_N+k,N+l), 0).
// Writing these 2N doubles instead of the n + n^2 state moments is what Metran.get_simulation
// needs (metran/metran.py:831-883) and cuts the smoother's output traffic ~7x at n = 10, ~40x at n = 36.
template <int N, int K, int G>
__device__ __forceinline__ void project(double x, const double (&P)[N + K], const double (&gam)[K], double scale,
                                        double offset, int lane, double &mean, double &var)
{
    using Gp = Group<G>;
    double m = x, t = 0.0, diag = 0.0;
    double pf[K][K]; // factor block of the covariance, replicated
    sfor<0, K>(MK_LAMBDA(k) {
        constexpr int kk = decltype(k)::value;
        m = fma(gam[kk], Gp::template bcast<N + kk>(x), m);
        sfor<kk, K>(MK_LAMBDA(l) {
            constexpr int ll = decltype(l)::value;
            pf[kk][ll] = pf[ll][kk] = Gp::template bcast<N + kk>(P[N + ll]);
        });
    });
    sfor<0, N>(MK_LAMBDA(c) { diag = (decltype(c)::value == lane) ? P[decltype(c)::value] : diag; });
    sfor<0, K>(MK_LAMBDA(k) {
        constexpr int kk = decltype(k)::value;
        double u = 2.0 * P[N + kk];
        sfor<0, K>(MK_LAMBDA(l) { u = fma(gam[decltype(l)::value], pf[kk][decltype(l)::value], u); });
        t = fma(gam[kk], u, t);
    });
    mean = fma(scale, m, offset);
    const double v = scale * scale * (diag + t);
    var = v < 0.0 ? 0.0 : v; // :601-602 (np.maximum keeps a NaN)
}

// log of a frexp-normalised mantissa m in [0.5, 1) as log(m) = e ln2 + l with e in {-1, 0}:
// m' = m or 2m in [1/sqrt2, sqrt2), s = (m'-1)/(m'+1), l = 2 atanh(s) = 2s (1 + s^2/3 + ... + s^18/19)
// (|s| < 0.1716: the truncated tail is < 2.3e-17 relative; measured max abs error 1.4e-16 against
// 120-bit arithmetic, libm 0.6e-16).  ~27 instructions instead of the ~100 of the generic log(), which
// was 1/6 of the filter's per-step instruction stream when the per-step determinants are booked.
__device__ __forceinline__ double log_mant(double m, int &e)
{
    if (__builtin_expect(!(m > 0.0), 0)) { // f <= 0 or NaN somewhere in the product: let libm say so
        e = 0;
        return log(m);
    }
    const bool low = m < 0.70710678118654752440;
    e = low ? -1 : 0;
    const double m2 = low ? m + m : m;
    const double s = (m2 - 1.0) * rcp_nr(m2 + 1.0);
    const double w = s * s;
    double p = 1.0 / 19.0;
    p = fma(p, w, 1.0 / 17.0);
    p = fma(p, w, 1.0 / 15.0);
    p = fma(p, w, 1.0 / 13.0);
    p = fma(p, w, 1.0 / 11.0);
    p = fma(p, w, 1.0 / 9.0);
    p = fma(p, w, 1.0 / 7.0);
    p = fma(p, w, 1.0 / 5.0);
    p = fma(p, w, 1.0 / 3.0);
    p = fma(p, w, 1.0);
    return (s + s) * p;
}

constexpr double kLn2 = 0.693147180559945309417232121458;
constexpr double kLog2Pi = 1.837877066409345483560659472811; // log(2*pi)


// A = L D L^T of the predicted covariance, right-looking and distributed: lane c ends up holding L(c, j) in
// A[j] for j < c, dinv[j] = 1/d_j.  The reference inverts Pp with numpy.linalg.pinv (kalmanfilter.py:455),
// which drops (numerically) null directions; an LDL^T pivot d_j <= 0 is such a direction (q_j = 0 when a
// series has communality 1, metran.py:314-316, and its state variance has decayed to rounding level).
// GUARD = false is the hot path (no test per stage; the caller checks `pivmin` once per step);
// GUARD = true re-factorises with 1/d_j := 0 and L(., j) := 0 for d_j <= 0 -- the generalised inverse
// L^-T D^+ L^-1, which acts like pinv on range(Pp), where J's operands live.  Positive pivots, however
// small, are inverted: measured against 60-digit arithmetic (tests/golden/heywood.npz) that is MORE accurate
// than the reference's truncation (1e-14 vs 7e-8 on the smoothed means at cond(Pp) = 1e17).
template <int n, int G, bool GUARD>
__device__ __forceinline__ void ldlt_factor(double (&A)[n], double (&dinv)[n], double &pivmin)
{
    using Gp = Group<G>;
    sfor<0, n>(MK_LAMBDA(jc) {
        constexpr int j = decltype(jc)::value;
        const double piv = Gp::template bcast<j>(A[j]); // d_j
        pivmin = min_f64(pivmin, piv);
        double ij = rcp_nr(piv);
        if constexpr (GUARD) ij = piv > 0.0 ? ij : 0.0;
        dinv[j] = ij;
        const double lr = A[j] * ij; // L(r, j), valid for r > j (A symmetric)
        // trailing update A[r][c] -= L(r,j) * a_jc, a_jc broadcast from lane j
        if constexpr (G == 16 && n == 15 && j + 1 < n && n - (j + 1) <= 2) {
            // n = 15 (e.g. 14 series + 1 factor) with the projection epilogue keeps the last rows of A in AGPRs across the
            // final stages, and the reload -- a VALU write the compiler places right before the use it sees -- lands ONE wait
            // state ahead of the statement's DPP read (found by scripts/check_dpp_hazards.py when the shape grid was first
            // prebuilt, round 6): the last two stages take the guarded statement (two wait states inside it)
            sfor<j + 1, n>(MK_LAMBDA(c) { Gp::template fmac<j, false, true>(A[decltype(c)::value], A[decltype(c)::value], -lr); });
        } else {
            Gp::template axpy_lane<j, j + 1, n, true, n>(A, A, lr);
        }
        A[j] = lr;
    });
}
// status bits of a smoother instance from the smallest pivot it met (covariances of standardised series are
// O(1): -1e-8 is far outside rounding)
__device__ __forceinline__ unsigned pivot_flags(double pivmin)
{
    unsigned f = 0u;
    if (!(pivmin > 0.0)) f |= MK_FLAG_RANK_DEFICIENT;
    if (!(pivmin >= -1e-8)) f |= MK_FLAG_NOT_SPD;
    return f;
}

} // namespace mk
