// mk_kernels.hip -- hand-written CDNA4 (gfx950) kernels for Metran's Kalman hot path.
//
// Mapping: one GROUP of G lanes owns one independent dynamic-factor model;
// lane r of the group owns ROW r of every n x n covariance (n = N + K <= G) in VGPRs and
// element r of every state vector.  G = 16 packs four models into one 64-wide wavefront (one
// model per DPP row; cross-lane traffic is `row_newbcast` DPP, no LDS, no barriers);
// G = 64 gives one model per wavefront (cross-lane traffic is v_readlane).
// The time recursion is sequential per model; parallelism comes from the batch.
// No MFMA: n is tiny and the path is bound by HBM traffic of the state outputs.
//
// Reference semantics restated here (file:line in /root/reference):
//   filter_kernel   : seqkalmanfilter  metran/kalmanfilter.py:236-400  + get_mle :550-567
//   smoother_kernel : kalmansmoother   metran/kalmanfilter.py:403-476
//   simulate/decompose kernels : SPKalmanFilter.simulate/decompose :569-644
//   params kernel   : Metran._phi / get_transition_* metran/metran.py:246-322
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

template <>
struct Group<16> {
    template <int J>
    static __device__ __forceinline__ double bcast(double v)
    {
        return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, true); // v_mov_b64_dpp row_newbcast:J
    }
    // acc += bcast<J>(src) * mul   /   acc -= bcast<J>(src) * mul
    template <int J, bool NEG = false>
    static __device__ __forceinline__ void fmac(double &acc, double src, double mul)
    {
        if constexpr (NEG)
            asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3" MK_DPPMASK : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
        else
            asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3" MK_DPPMASK : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
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
    static __device__ __forceinline__ unsigned long long group_bits(unsigned long long m) { return m; }
    static constexpr unsigned long long full_mask(int N) { return N >= 64 ? ~0ull : (1ull << N) - 1ull; }
};

template <int n>
__device__ __forceinline__ void store_row(double *dst, const double (&row)[n])
{
    if constexpr (n % 2 == 0) {
        double2 *d2 = reinterpret_cast<double2 *>(dst); // row offset is a multiple of 16 B
#pragma unroll
        for (int c = 0; c < n / 2; ++c) d2[c] = make_double2(row[2 * c], row[2 * c + 1]);
    } else {
#pragma unroll
        for (int c = 0; c < n; ++c) dst[c] = row[c];
    }
}

template <int n>
__device__ __forceinline__ void load_row(const double *src, double (&row)[n])
{
    if constexpr (n % 2 == 0) {
        const double2 *s2 = reinterpret_cast<const double2 *>(src);
#pragma unroll
        for (int c = 0; c < n / 2; ++c) {
            double2 v = s2[c];
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

// ---------------------------------------------------------------- coalesced n x n block I/O
// A model's covariance at one time step is one contiguous n*n*8-byte block in HBM, but lane r
// holds row r.  Storing rows directly makes every store instruction touch ~n scattered 16-byte
// pieces per model (measured: the filter ran at 2.5 TB/s, store-issue bound).  Instead the
// group's block is transposed through LDS (wave-private, no workgroup barrier): lanes write
// their rows, then lane l moves the 16-byte chunks l, l+G, l+2G, ... so that one store/load
// instruction covers G*16 contiguous bytes per model.
__device__ __forceinline__ void wave_lds_sync()
{
    // LDS-only ("local") fences: a generic fence would also pin private arrays to scratch memory
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

typedef double v2d __attribute__((ext_vector_type(2))); // native <2 x double> (stays in VGPRs)

template <int n, int G>
struct BlockIO {
    static constexpr int NN = n * n;
    static constexpr bool V2 = (NN % 2 == 0);            // 16-byte chunks when the block allows it
    static constexpr int CH = V2 ? NN / 2 : NN;          // chunks per block
    static constexpr int PER = (CH + G - 1) / G;         // chunks per lane
    static constexpr int STRIDE = ((NN * 8 + 255) / 256) * 256 / 8; // doubles between groups' LDS buffers
    static constexpr int LDS_DOUBLES = (256 / G) * STRIDE;
    using chunk_t = typename std::conditional<V2, v2d, double>::type;

    // Lanes >= n (and the groups of a partial last workgroup) are exact REPLICAS of lane n-1 (of the
    // last model): same inputs, same instruction stream.  They therefore write identical bytes to
    // identical addresses, and no store in the hot loops needs an exec mask.
    // rows (registers) -> LDS -> HBM block; r = min(lane, n-1)
    static __device__ __forceinline__ void store(double *lds, double *gblock, const double (&row)[n], int lane, int r)
    {
        wave_lds_sync();
        store_row<n>(lds + r * n, row);
        wave_lds_sync();
        chunk_t *g = reinterpret_cast<chunk_t *>(gblock);
        const chunk_t *l = reinterpret_cast<const chunk_t *>(lds);
        chunk_t tmp[PER];
        int qi[PER];
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            qi[m] = lane + m * G;
            if (qi[m] > CH - 1) qi[m] = CH - 1; // duplicates of the last chunk: same data, same address
            tmp[m] = l[qi[m]];
        }
#pragma unroll
        for (int m = 0; m < PER; ++m) g[qi[m]] = tmp[m];
    }
    // two blocks (predicted + filtered covariance of one step) through ONE LDS round trip;
    // lds must hold 2 * STRIDE doubles for this group
    static __device__ __forceinline__ void store2(double *lds, double *g0, const double (&row0)[n], double *g1,
                                                  const double (&row1)[n], int lane, int r)
    {
        wave_lds_sync();
        store_row<n>(lds + r * n, row0);
        store_row<n>(lds + STRIDE + r * n, row1);
        wave_lds_sync();
        const chunk_t *l0 = reinterpret_cast<const chunk_t *>(lds);
        const chunk_t *l1 = reinterpret_cast<const chunk_t *>(lds + STRIDE);
        chunk_t t0[PER], t1[PER];
        int qi[PER];
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            qi[m] = lane + m * G;
            if (qi[m] > CH - 1) qi[m] = CH - 1;
            t0[m] = l0[qi[m]];
            t1[m] = l1[qi[m]];
        }
#pragma unroll
        for (int m = 0; m < PER; ++m) reinterpret_cast<chunk_t *>(g0)[qi[m]] = t0[m];
#pragma unroll
        for (int m = 0; m < PER; ++m) reinterpret_cast<chunk_t *>(g1)[qi[m]] = t1[m];
    }
    // Split form of store2 for kernels that spread their global stores over the following compute:
    // a CU's vector-memory path moves only ~64 B/clk and is shared by its wavefronts, so a burst of
    // back-to-back 1 KB stores stalls the issuing wavefront (measured: +0.8 ms on the filter);
    // issued one at a time between the scalar updates of the NEXT step they cost almost nothing.
    static __device__ __forceinline__ int chunk_index(int lane, int m)
    {
        const int qi = lane + m * G;
        return qi > CH - 1 ? CH - 1 : qi;
    }
    static __device__ __forceinline__ void stage2(double *lds, const double (&row0)[n], const double (&row1)[n],
                                                  chunk_t (&t0)[PER], chunk_t (&t1)[PER], int lane, int r)
    {
        wave_lds_sync();
        store_row<n>(lds + r * n, row0);
        store_row<n>(lds + STRIDE + r * n, row1);
        wave_lds_sync();
        const chunk_t *l0 = reinterpret_cast<const chunk_t *>(lds);
        const chunk_t *l1 = reinterpret_cast<const chunk_t *>(lds + STRIDE);
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            const int qi = chunk_index(lane, m);
            t0[m] = l0[qi];
            t1[m] = l1[qi];
        }
    }
    // HBM block -> registers (issue early; the data is consumed one time step later)
    static __device__ __forceinline__ void load_issue(const double *gblock, chunk_t (&buf)[PER], int lane)
    {
        const chunk_t *g = reinterpret_cast<const chunk_t *>(gblock);
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            const int qi = lane + m * G;
            buf[m] = g[qi < CH ? qi : CH - 1]; // clamped, branch-free (keeps buf in registers)
        }
    }
    // registers -> LDS -> row r
    static __device__ __forceinline__ void load_finish(double *lds, const chunk_t (&buf)[PER], double (&row)[n],
                                                       int lane, int r)
    {
        wave_lds_sync();
        chunk_t *l = reinterpret_cast<chunk_t *>(lds);
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            const int qi = lane + m * G;
            l[qi < CH ? qi : CH - 1] = buf[m];
        }
        wave_lds_sync();
        load_row<n>(lds + r * n, row);
    }
};

constexpr double kLn2 = 0.693147180559945309417232121458;
constexpr double kLog2Pi = 1.837877066409345483560659472811; // log(2*pi)

// =====================================================================================
// Sequential-processing Kalman filter + -2 log L            (kalmanfilter.py:236-400, 550-567)
//   OUT  : 0 = no state outputs (solver objective, mk_loglik), 1 = F, Pf, Xp, Pp all written,
//          2 = any subset (runtime null checks)
//   BOOK : per-step sigmas/detfs arrays are written (needs one log per step); otherwise the
//          log-determinant is accumulated as a normalised product with ONE log at the end
// =====================================================================================
template <int N, int K, int G, int OUT, bool BOOK>
__global__ void __launch_bounds__(256) filter_kernel(FilterArgs a)
{
    constexpr int n = N + K;
    static_assert(n <= G, "state dimension must fit the lane group");
    using Gp = Group<G>;
    constexpr int GPB = 256 / G; // models per 256-thread workgroup
    constexpr bool HOIST = (N * K <= 32); // keep Z's loading block replicated in registers
    const int lane = threadIdx.x % G;
    long inst = (long)blockIdx.x * GPB + threadIdx.x / G;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1; // surplus groups replicate the last model (identical stores, see BlockIO)
    const long rec = inst % a.R;
    const bool rowok = lane < n;
    const int r = rowok ? lane : n - 1;
    const bool lead = lane == 0;
    const long T = a.T;

    // per-model constants: lane r holds phi_r, q_r; lane j < N holds loadings[j,:] and obsvar[j]
    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    double pp[n], qd[n]; // row r of Phi (x) Phi and of Q = diag(q)
    sfor<0, n>(MK_LAMBDA(c) {
        pp[decltype(c)::value] = phi_r * Gp::template bcast<decltype(c)::value>(phi_r);
        qd[decltype(c)::value] = (decltype(c)::value == r) ? q_r : 0.0;
    });
    const int jr = lane < N ? lane : N - 1;
    double gam[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gam[k] = a.loadings[(rec * N + jr) * K + k];
    const double rvar = a.obsvar ? a.obsvar[rec * N + jr] : 0.0;
    double Gh[HOIST ? N : 1][K]; // Gh[j][k] = loadings[j,k] in every lane
    if constexpr (HOIST) {
        sfor<0, N>(MK_LAMBDA(j) {
            sfor<0, K>(MK_LAMBDA(k) {
                Gh[decltype(j)::value][decltype(k)::value] = Gp::template bcast<decltype(j)::value>(gam[decltype(k)::value]);
            });
        });
    }

    // initial state (run_filter defaults, kalmanfilter.py:747-750)
    double x = a.x0 ? a.x0[inst * n + r] : 0.0;
    double P[n];
#pragma unroll
    for (int c = 0; c < n; ++c) P[c] = a.P0 ? a.P0[(inst * n + r) * n + c] : (c == r ? 1.0 : 0.0);

    using BIO = BlockIO<n, G>;
    constexpr int LDS_PER_GROUP = (OUT == 1 ? 2 : 1) * BIO::STRIDE;
    __shared__ __attribute__((aligned(16))) double lds_io[OUT ? (256 / G) * LDS_PER_GROUP : 2];
    double *lds = lds_io + (OUT ? (threadIdx.x / G) * LDS_PER_GROUP : 0); // this model's staging buffer

    const double *yp = a.obs + rec * a.obs_bs * N + jr; // lane j streams series j
    const long ystep = a.obs_ts * N;
    // observation ring: y of step t was requested YQ steps earlier (HBM latency under the store
    // traffic of this kernel exceeds one time step when a SIMD holds a single wavefront)
    constexpr int YQ = 2;
    double yq[YQ];
#pragma unroll
    for (int i = 0; i < YQ; ++i) yq[i] = yp[(i < T ? i : T - 1) * ystep];
    double sum_sig = 0.0, sum_det = 0.0;
    double run_mant = 1.0; // !BOOK: prod of f over the counted steps, normalised
    long run_exp = 0;
    long nobs = 0, sc = 0;
    double fmin_seen = 1.0;
    // (b, t) lives at block index b * bs + t * ts: (bs, ts) = (T, 1) model-major [B,T,..] or (1, B) time-major [T,B,..]
    const long vstep = a.ts * n, bstep = a.ts * n * n;
    const long vec0 = inst * a.bs * n + r;
    double *pXp = (OUT && a.Xp) ? a.Xp + vec0 : nullptr;
    double *pF = (OUT && a.F) ? a.F + vec0 : nullptr;
    double *pPp = (OUT && a.Pp) ? a.Pp + inst * a.bs * n * n : nullptr; // block of step t
    double *pPf = (OUT && a.Pf) ? a.Pf + inst * a.bs * n * n : nullptr;

    // OUT == 1: the covariance blocks of step t are transposed to 16-byte chunks at the end of step t and
    // written during step t+1, one store after each scalar update.  The staging registers start as
    // zeros aimed at block 0, which the real step-0 data overwrites later (same wavefront, in order).
    constexpr int PERC = OUT == 1 ? BIO::PER : 1;
    constexpr int NSLOT = 2 * PERC;                 // stores per step
    constexpr int SPO = (NSLOT + N - 1) / N;        // stores issued after each scalar update
    typename BIO::chunk_t stg0[PERC], stg1[PERC];
#pragma unroll
    for (int m = 0; m < PERC; ++m) stg0[m] = stg1[m] = typename BIO::chunk_t(0.0);
    double *sPp = pPp, *sPf = pPf;
    auto emit = MK_LAMBDA(sc) {
        constexpr int slot = decltype(sc)::value;
        if constexpr (slot < PERC)
            reinterpret_cast<typename BIO::chunk_t *>(sPp)[BIO::chunk_index(lane, slot)] = stg0[slot];
        else
            reinterpret_cast<typename BIO::chunk_t *>(sPf)[BIO::chunk_index(lane, slot - PERC)] = stg1[slot - PERC];
    };

    for (long t = 0; t < T; ++t) {
        const double y = yq[0];
#pragma unroll
        for (int i = 0; i + 1 < YQ; ++i) yq[i] = yq[i + 1];
        yq[YQ - 1] = yp[(t + YQ < T ? t + YQ : T - 1) * ystep]; // clamped: branch-free prefetch
        // which series are observed at this step (NaN / inf = missing, kalmanfilter.py:657)
        const unsigned long long ball = __ballot(lane < N && isfinite(y));
        const auto vm = Gp::group_bits(ball);

        // ---- predict (:318-331; Phi diagonal) ----
        x = phi_r * x;
#pragma unroll
        for (int c = 0; c < n; ++c) P[c] = fma(P[c], pp[c], qd[c]);
        double Ppred[OUT == 1 ? n : 1]; // OUT == 1: keep the predicted row, store it with the filtered one
        if constexpr (OUT == 1) {
            *pXp = x; // :332 (replica lanes rewrite lane n-1's element)
            pXp += vstep;
#pragma unroll
            for (int c = 0; c < n; ++c) Ppred[c] = P[c]; // :333, written at the end of the step
        } else if constexpr (OUT == 2) {
            if (pXp) {
                *pXp = x;
                pXp += vstep;
            }
            if (pPp) {
                BIO::store(lds, pPp, P, lane, r);
                pPp += bstep;
            }
        }

        // ---- sequential scalar updates (:341-378), observations in ascending series order ----
        double sigma = 0.0, fmant = 1.0;
        int fexp = 0;
        auto update = MK_LAMBDA(jc) {
            constexpr int j = decltype(jc)::value;
            // innovation (:344-347): every lane l < N forms v_l = y_l - Z_l x with ITS loadings;
            // lane j's value is the one used
            double vl = y - x;
            sfor<0, K>(MK_LAMBDA(k) {
                constexpr int kk = decltype(k)::value;
                vl = fma(-gam[kk], Gp::template bcast<N + kk>(x), vl);
            });
            const double v = Gp::template bcast<j>(vl);
            // d = P Z_j^T : lane r computes d_r from its own row (:349-357)
            double dr = P[j];
            sfor<0, K>(MK_LAMBDA(k) {
                constexpr int kk = decltype(k)::value;
                double g;
                if constexpr (HOIST) g = Gh[j][kk];
                else g = Gp::template bcast<j>(gam[kk]);
                dr = fma(P[N + kk], g, dr);
            });
            // innovation variance f = R_j + Z_j d (:359-362), formed at lane j from d_j, d_{N+k}
            double fl = rvar + dr;
            sfor<0, K>(MK_LAMBDA(k) {
                constexpr int kk = decltype(k)::value;
                fl = fma(gam[kk], Gp::template bcast<N + kk>(dr), fl);
            });
            const double f = Gp::template bcast<j>(fl);
            const double rf = rcp_nr(f);
            const double kr = dr * rf; // Kalman gain element r (:364-366)
            // P -= k k^T f (:368-372): P[r][c] -= d_c * k_r, d_c broadcast from lane c
            Gp::template axpy_col<0, n, true, n>(P, dr, kr);
            x = fma(kr, v, x);             // :374-375
            sigma = fma(v * v, rf, sigma); // :377
            // detf += log f (:378): accumulate prod f as mantissa * 2^exp
            fmant *= f;
            if constexpr ((j & 3) == 3 || j == N - 1) {
                fexp += __builtin_amdgcn_frexp_exp(fmant);
                fmant = __builtin_amdgcn_frexp_mant(fmant);
            }
            fmin_seen = min_f64(fmin_seen, f);
        };
        auto stores_after = MK_LAMBDA(jc) { // previous step's staged chunks, SPO per scalar update
            if constexpr (OUT == 1) {
                constexpr int s0 = decltype(jc)::value * SPO;
                constexpr int s1 = (s0 + SPO) < NSLOT ? (s0 + SPO) : NSLOT;
                sfor<s0, s1>(emit);
            }
        };
        if (ball == Gp::full_mask(N)) { // every model of this wavefront observes all N series: no masking
            sfor<0, N>(MK_LAMBDA(jc) {
                update(jc);
                stores_after(jc);
            });
        } else {
            sfor<0, N>(MK_LAMBDA(jc) {
                if ((vm >> decltype(jc)::value) & 1) { // uniform within the model's lane group
                    update(jc);
                } else if constexpr ((decltype(jc)::value & 3) == 3 || decltype(jc)::value == N - 1) {
                    fexp += __builtin_amdgcn_frexp_exp(fmant); // keep the product normalised
                    fmant = __builtin_amdgcn_frexp_mant(fmant);
                }
                stores_after(jc);
            });
        }

        const int cnt = __popcll((unsigned long long)vm);
        if (cnt > 0) { // :380-382 compressed bookkeeping
            if constexpr (BOOK) {
                const double detf = fma((double)fexp, kLn2, log(fmant));
                if (a.sigmas && lead) a.sigmas[inst * a.bs + sc * a.ts] = sigma;
                if (a.detfs && lead) a.detfs[inst * a.bs + sc * a.ts] = detf;
                if (sc >= a.warmup) { // get_mle: detfs[warmup:], sigmas[warmup:] are COMPRESSED indices (:563-564)
                    sum_det += detf;
                    sum_sig += sigma;
                }
            } else {
                if (sc >= a.warmup) {
                    sum_sig += sigma;
                    run_mant *= fmant;
                    run_exp += fexp + __builtin_amdgcn_frexp_exp(run_mant);
                    run_mant = __builtin_amdgcn_frexp_mant(run_mant);
                }
            }
            ++sc;
        }
        if (t >= a.warmup) nobs += cnt; // observation_count[warmup:] is a TIME index (:565)

        if constexpr (OUT == 1) {
            *pF = x; // :389
            pF += vstep;
            BIO::stage2(lds, Ppred, P, stg0, stg1, lane, r); // :333, :390 -> written during the next step
            sPp = pPp;
            sPf = pPf;
            pPp += bstep;
            pPf += bstep;
        } else if constexpr (OUT == 2) {
            if (pF) {
                *pF = x;
                pF += vstep;
            }
            if (pPf) {
                BIO::store(lds, pPf, P, lane, r);
                pPf += bstep;
            }
        }
    }

    if constexpr (OUT == 1) sfor<0, NSLOT>(emit); // flush the last step's blocks

    // zero tail of the compressed arrays (np.zeros init, :307-308)
    if (BOOK) {
        for (long i = sc + lane; i < T; i += G) {
            if (a.sigmas) a.sigmas[inst * a.bs + i * a.ts] = 0.0;
            if (a.detfs) a.detfs[inst * a.bs + i * a.ts] = 0.0;
        }
    }
    if (lead) {
        if (!BOOK) sum_det = fma((double)run_exp, kLn2, log(run_mant));
        if (a.mle) a.mle[inst] = ((double)nobs * kLog2Pi + sum_det) + sum_sig; // :566
        if (a.sigmacount) a.sigmacount[inst] = sc;
        if (a.status) a.status[inst] = (fmin_seen > 0.0) ? 0u : MK_FLAG_NONPOSITIVE_F; // NaN f also flags
    }
}

// =====================================================================================
// RTS smoother                                               (kalmanfilter.py:403-476)
//   Pp[t+1] = Phi Pf[t] Phi + Q and Xp[t+1] = Phi F[t] are recomputed (Phi diagonal), so only
//   F and Pf are re-read.  J = Pf Phi^T Pp^{-1} through an LDL^T factorisation of Pp (SPD
//   whenever q > 0, where the reference's pinv (:455) is the inverse); lane i solves for ROW i of J.
//   The factor is DISTRIBUTED: lane c keeps L(c, 0..c-1) in place of its row of A.
// =====================================================================================
template <int n, int G>
__global__ void __launch_bounds__(256) smoother_kernel(SmootherArgs a)
{
    static_assert(n <= G, "state dimension must fit the lane group");
    using Gp = Group<G>;
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    long inst = (long)blockIdx.x * GPB + threadIdx.x / G;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1;
    const bool rowok = lane < n;
    const int r = rowok ? lane : n - 1;
    const long T = a.T;

    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    double phic[n];
    sfor<0, n>(MK_LAMBDA(c) { phic[c] = Gp::template bcast<decltype(c)::value>(phi_r); });

    using BIO = BlockIO<n, G>;
    __shared__ __attribute__((aligned(16))) double lds_io[BIO::LDS_DOUBLES];
    double *lds = lds_io + (threadIdx.x / G) * BIO::STRIDE;

    double qd[n]; // row r of Q = diag(q)
#pragma unroll
    for (int c = 0; c < n; ++c) qd[c] = (c == r) ? q_r : 0.0;

    // last step: smoothed = filtered (:450-451)
    const long vstep = a.ts * n, bstep = a.ts * n * n;
    const long blkT = inst * a.bs + (T - 1) * a.ts; // block index of (inst, T-1)
    const long vecT = blkT * n + r;
    const double *pF = a.F + vecT;
    const double *pPf = a.Pf + blkT * n * n; // block of step t
    double *pS = a.S ? a.S + vecT : nullptr;
    double *pPs = a.Ps ? a.Ps + blkT * n * n : nullptr;
    double xs = *pF;
    double Psn[n];
    typename BIO::chunk_t pre[BIO::PER];
    BIO::load_issue(pPf, pre, lane);
    BIO::load_finish(lds, pre, Psn, lane, r);
    if (pS) *pS = xs;
    if (pPs) BIO::store(lds, pPs, Psn, lane, r);
    double pivmin = 1.0;

    // Software pipeline: at the top of iteration t the ROWS of Pf[t] (Pfc) and F[t] (xfc) are already in
    // registers and the 16-byte chunks of Pf[t-1] (pre) are in flight from HBM; their LDS transposition
    // (pre -> Pfn) is issued in the middle of the iteration so that its latency hides behind the
    // Ps sweep, and the HBM loads for t-2 are issued right after it.
    double Pfc[n], xfc = 0.0, xfn = 0.0;
    if (T >= 2) {
        pF -= vstep;
        pPf -= bstep;
        BIO::load_issue(pPf, pre, lane);
        xfc = *pF;
        BIO::load_finish(lds, pre, Pfc, lane, r);
        if (T >= 3) {
            pF -= vstep;
            pPf -= bstep;
            BIO::load_issue(pPf, pre, lane);
            xfn = *pF;
        }
    }

    for (long t = T - 2; t >= 0; --t) {
        // W = Pf Phi (column scaling); A = Pp[t+1] = Phi Pf Phi + Q (row r); D = Ps[t+1] - Pp[t+1]
        double A[n], z[n], D[n];
#pragma unroll
        for (int c = 0; c < n; ++c) {
            z[c] = Pfc[c] * phic[c]; // W, the right-hand side of the solve
            A[c] = fma(phi_r, z[c], qd[c]);
            D[c] = Psn[c] - A[c];
        }

        double delta = xs - phi_r * xfc; // xs[t+1] - Xp[t+1]; formed early so that it is "old" when DPP-read
        if constexpr (G == 16) dpp_pin(delta);

        // ---- A = L D L^T, right-looking; lane c ends up holding L(c, j) in A[j] for j < c ----
        double dinv[n];
        sfor<0, n>(MK_LAMBDA(jc) {
            constexpr int j = decltype(jc)::value;
            const double piv = Gp::template bcast<j>(A[j]); // d_j
            pivmin = min_f64(pivmin, piv);
            const double ij = rcp_nr(piv);
            dinv[j] = ij;
            const double lr = A[j] * ij; // L(r, j), valid for r > j (A symmetric)
            // trailing update A[r][c] -= L(r,j) * a_jc, a_jc broadcast from lane j
            Gp::template axpy_lane<j, j + 1, n, true, n>(A, A, lr);
            A[j] = lr;
        });

        // ---- lane i solves A z = W_i  (row i of J = Pf Phi^T A^{-1}, :458-460) ----
        sfor<0, n>(MK_LAMBDA(kc) { // forward: L y = b;  z[c] -= L(c,k) y_k, L(c,k) lives in lane c
            constexpr int k = decltype(kc)::value;
            Gp::template axpy_col<k + 1, n, true, n>(z, A[k], z[k]);
        });
#pragma unroll
        for (int c = 0; c < n; ++c) z[c] *= dinv[c]; // D^{-1}
        sfor_down<0, n>(MK_LAMBDA(kc) { // backward: L^T z = y;  z[c] -= L(k,c) z_k, L(k,c) lives in lane k
            constexpr int k = decltype(kc)::value;
            Gp::template axpy_lane<k, 0, k, true, n>(z, A, z[k]);
        });
        // z = J[r, :]

        // ---- smoothed mean (:461-464): xs[t] = F[t] + J (xs[t+1] - Phi F[t]) ----
        double acc0 = xfc, acc1 = 0.0;
        sfor<0, n>(MK_LAMBDA(cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c % 2 == 0) Gp::template fmac<c>(acc0, delta, z[c]);
            else Gp::template fmac<c>(acc1, delta, z[c]);
        });
        xs = acc0 + acc1;

        // ---- smoothed covariance (:465-474): Ps[t] = Pf[t] + J (Ps[t+1] - Pp[t+1]) J^T ----
        double V[n]; // V = J D (row r):  V[c] += J[r][k] * D[k][c], D[k][:] broadcast from lane k
#pragma unroll
        for (int c = 0; c < n; ++c) V[c] = 0.0;
        if constexpr (G == 16) dpp_guard(D); // D is compiler-produced (build-time hazard check)
        sfor<0, n>(MK_LAMBDA(kc) {
            constexpr int k = decltype(kc)::value;
            Gp::template axpy_lane<k, 0, n, false, n>(V, D, z[k]);
        });
#pragma unroll
        for (int c = 0; c < n; ++c) Psn[c] = Pfc[c];

        // mid-iteration: transpose the chunks of Pf[t-1] into rows (LDS latency hides behind the
        // sweep below) and issue the HBM loads of Pf[t-2]
        const double xf_next = xfn;
        if (t >= 1) {
            BIO::load_finish(lds, pre, Pfc, lane, r);
            if (t >= 2) {
                pF -= vstep;
                pPf -= bstep;
                BIO::load_issue(pPf, pre, lane);
                xfn = *pF;
            }
        }

        // Ps[r][c] = Pf[r][c] + sum_k V[r][k] J[c][k], J[c][k] broadcast from lane c
        sfor<0, n>(MK_LAMBDA(kc) {
            constexpr int k = decltype(kc)::value;
            Gp::template axpy_col<0, n, false, n>(Psn, z[k], V[k]);
        });

        if (pS) {
            pS -= vstep;
            *pS = xs;
        }
        if (pPs) {
            pPs -= bstep;
            BIO::store(lds, pPs, Psn, lane, r);
        }
        xfc = xf_next;
    }
    if (a.status && live && lane == 0 && !(pivmin > 0.0)) atomicOr(a.status + inst, MK_FLAG_NOT_SPD);
}

// =====================================================================================
// Small helper kernels
// =====================================================================================
// Metran._phi / get_transition_matrix / get_transition_covariance diagonals (metran.py:246-322)
__global__ void params_kernel(long B, long R, int N, int K, const double *alpha, const double *loadings,
                              double dt, double *phi, double *q)
{
    const int n = N + K;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n) return;
    const long b = i / n;
    const int s = (int)(i % n);
    const double ph = exp(-dt / alpha[i]);
    double qq = 1.0 - ph * ph;
    if (s < N) {
        const double *g = loadings + ((b % R) * N + s) * K;
        double comm = 0.0;
        for (int k = 0; k < K; ++k) comm += g[k] * g[k];
        qq *= (1.0 - comm);
    }
    phi[i] = ph;
    q[i] = qq;
}

// SPKalmanFilter.simulate (kalmanfilter.py:596-602): one thread per (b, t, j)
__global__ void simulate_kernel(long B, long RZ, long T, int N, int n, const double *Z, const double *means,
                                const double *covs, double *sim_means, double *sim_vars)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T * N) return;
    const int j = (int)(i % N);
    const long bt = i / N;
    const long b = bt / T;
    const double *z = Z + ((b % RZ) * N + j) * n;
    const double *x = means + bt * n;
    double m = 0.0;
    for (int c = 0; c < n; ++c) m += z[c] * x[c];
    if (sim_means) sim_means[i] = m;
    if (sim_vars && covs) {
        const double *P = covs + bt * n * n;
        double v = 0.0;
        for (int rr = 0; rr < n; ++rr) {
            double s = 0.0;
            for (int c = 0; c < n; ++c) s += P[rr * n + c] * z[c];
            v += z[rr] * s;
        }
        sim_vars[i] = v > 0.0 ? v : 0.0;
    }
}

// SPKalmanFilter.decompose (kalmanfilter.py:633-643)
__global__ void decompose_kernel(long B, long RZ, long T, int N, int n, const double *Z, const double *means,
                                 double *sdf, double *cdf)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T * N) return;
    const int j = (int)(i % N);
    const long bt = i / N;
    const long b = bt / T, t = bt % T;
    const int K = n - N;
    const double *z = Z + ((b % RZ) * N + j) * n;
    const double *x = means + bt * n;
    double s = 0.0;
    for (int c = 0; c < N; ++c) s += z[c] * x[c];
    if (sdf) sdf[i] = s;
    if (cdf)
        for (int k = 0; k < K; ++k) cdf[((b * K + k) * T + t) * N + j] = z[N + k] * x[N + k];
}

// deterministic single-workgroup tree sum (fixed order -> identical on every run / rank count)
__global__ void sum_kernel(long count, const double *v, double *out)
{
    __shared__ double sh[1024];
    double s = 0.0;
    for (long i = threadIdx.x; i < count; i += 1024) s += v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// =====================================================================================
// Shape registry and launchers
// =====================================================================================
template <int N, int K>
static hipError_t launch_filter_nk(const FilterArgs &a, hipStream_t s)
{
    constexpr int n = N + K;
    constexpr int G = n <= 16 ? 16 : 64;
    constexpr int GPB = 256 / G;
    const unsigned grid = (unsigned)((a.B + GPB - 1) / GPB);
    const bool book = a.sigmas || a.detfs;
    const bool any = a.F || a.Pf || a.Xp || a.Pp;
    const bool all = a.F && a.Pf && a.Xp && a.Pp;
    if (!any && !book)
        hipLaunchKernelGGL((filter_kernel<N, K, G, 0, false>), dim3(grid), dim3(256), 0, s, a);
    else if (!any)
        hipLaunchKernelGGL((filter_kernel<N, K, G, 0, true>), dim3(grid), dim3(256), 0, s, a);
    else if (all)
        hipLaunchKernelGGL((filter_kernel<N, K, G, 1, true>), dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((filter_kernel<N, K, G, 2, true>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int n>
static hipError_t launch_smoother_n(const SmootherArgs &a, hipStream_t s)
{
    constexpr int G = n <= 16 ? 16 : 64;
    constexpr int GPB = 256 / G;
    const unsigned grid = (unsigned)((a.B + GPB - 1) / GPB);
    hipLaunchKernelGGL((smoother_kernel<n, G>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

#define MK_CASE_FILTER(NN, KK) \
    if (N == NN && K == KK) return launch_filter_nk<NN, KK>(a, s);
#define MK_CASE_SMOOTH(NN, KK) \
    if (N + K == NN + KK) return launch_smoother_n<NN + KK>(a, s);
#define MK_CASE_LIST(NN, KK) {NN, KK},

hipError_t launch_filter(int N, int K, const FilterArgs &a, hipStream_t s)
{
    MK_SHAPES(MK_CASE_FILTER)
    return hipErrorInvalidValue;
}

hipError_t launch_smoother(int N, int K, const SmootherArgs &a, hipStream_t s)
{
    MK_SHAPES(MK_CASE_SMOOTH)
    return hipErrorInvalidValue;
}

static const int kShapes[][2] = {MK_SHAPES(MK_CASE_LIST)};

int num_shapes() { return (int)(sizeof(kShapes) / sizeof(kShapes[0])); }
void get_shape(int i, int *N, int *K)
{
    *N = kShapes[i][0];
    *K = kShapes[i][1];
}

hipError_t launch_params(long B, long R, int N, int K, const double *alpha, const double *loadings, double dt,
                         double *phi, double *q, hipStream_t s)
{
    const long tot = B * (N + K);
    hipLaunchKernelGGL(params_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, B, R, N, K, alpha,
                       loadings, dt, phi, q);
    return hipGetLastError();
}

hipError_t launch_simulate(long B, long RZ, long T, int N, int n, const double *Z, const double *means,
                           const double *covs, double *sm, double *sv, hipStream_t s)
{
    const long tot = B * T * N;
    hipLaunchKernelGGL(simulate_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, B, RZ, T, N, n, Z,
                       means, covs, sm, sv);
    return hipGetLastError();
}

hipError_t launch_decompose(long B, long RZ, long T, int N, int n, const double *Z, const double *means,
                            double *sdf, double *cdf, hipStream_t s)
{
    const long tot = B * T * N;
    hipLaunchKernelGGL(decompose_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, B, RZ, T, N, n, Z,
                       means, sdf, cdf);
    return hipGetLastError();
}

hipError_t launch_sum(long count, const double *v, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, s, count, v, out);
    return hipGetLastError();
}

} // namespace mk
