// mk_kernels.hip -- hand-written CDNA4 (gfx950) kernels for Metran's Kalman hot path.
//
// Mapping: one GROUP of G lanes owns one independent dynamic-factor model;
// lane r of the group owns ROW r of every n x n covariance (n = N + K <= G) in VGPRs and
// element r of every state vector.  G = 16 packs four models into one 64-wide wavefront (one
// model per DPP row; cross-lane traffic is `row_newbcast` DPP, no LDS, no barriers);
// G = 64 gives one model per wavefront (cross-lane traffic is v_readlane).
// The time recursion is sequential per model; parallelism comes from the batch.
// No MFMA in the default kernels: n is tiny, the broadcast-FMA count is the floor of this mapping and the f64 vector
// pipe -- not HBM, not issue -- is what bounds them; smoother_blk_kernel (opt-in) runs the two
// n^3 products of the smoother as 4x4x4 f64 MFMA blocks and measures slower.
//
// Reference semantics restated here (file:line in /root/reference):
//   filter_kernel   : seqkalmanfilter  metran/kalmanfilter.py:236-400  + get_mle :550-567
//   smoother_kernel : kalmansmoother   metran/kalmanfilter.py:403-476
//   simulate/decompose kernels : SPKalmanFilter.simulate/decompose :569-644
//   params kernel   : Metran._phi / get_transition_* metran/metran.py:246-322
#include <cstdlib>
#include <cstring>

#include "mk_prims.h"

namespace mk {

// =====================================================================================
// Sequential-processing Kalman filter + -2 log L            (kalmanfilter.py:236-400, 550-567)
//   OUT  : 0 = no state outputs (solver objective, mk_loglik), 1 = predicted + filtered packed records,
//          3 = filtered packed record only (input of a projecting smoother), 2 = any dense subset,
//          4 = the BACKWARD TAPE of the inverse-free smoother (mk_dk.hip; G = 64 only: the shapes with more than 32 series,
//              which the split layout of mk_split.hip -- the tape's writer for N <= 32 -- does not serve; same entries)
//   BOOK : per-step sigmas/detfs are written (needs one log per step); otherwise the
//          log-determinant is accumulated as a normalised product with ONE log at the end
// =====================================================================================
template <int n, int G, bool SYM>
struct RecordIO;
// packed records of the 16-lane kernels leave through wave-private LDS images as whole 16-byte chunks
// (8 global_store_dwordx4 per step instead of 24 column-run dwordx2 stores: 1.52 -> 1.47 ms at B=4096)
// resident wavefronts per SIMD asked of the compiler for the wide (one model per wavefront) filter
#ifndef MK_WIDE_FILTER_WAVES
#define MK_WIDE_FILTER_WAVES 2
#endif
// wide filter: scalar updates as a run-time loop over the observed series (1) or fully unrolled (0)
#ifndef MK_WIDE_FILTER_LOOP
#define MK_WIDE_FILTER_LOOP 1
#endif
#ifndef MK_FILTER_LDS_STORES
#define MK_FILTER_LDS_STORES 1
#endif

// (two resident wavefronts per SIMD are asked for only while the wide variant's two n-double row arrays fit
// 256 VGPRs: compile-checked at n = 64, the constraint spilled 1.1 KB per lane)
template <int N, int K, int G, int OUT, bool BOOK, bool SYM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(G == 64 && N + K <= 40 ? MK_WIDE_FILTER_WAVES : 1))) filter_kernel(FilterArgs a)
{
    static_assert(!SYM || OUT == 1 || OUT == 3, "packed-symmetric layout applies to record outputs");
    constexpr int n = N + K;
    static_assert(n <= G, "state dimension must fit the lane group");
    constexpr bool TAPE = (OUT == 4);
    static_assert(!TAPE || (G == 64 && MK_WIDE_FILTER_LOOP && N * K > 32), "the tape leaves the one-model-per-wavefront loop filter");
    using Gp = Group<G>;
    constexpr int GPB = 256 / G; // models per 256-thread workgroup
    constexpr bool HOIST = (N * K <= 32); // keep Z's loading block replicated in registers
    constexpr int NV = record_payload(n);
    const int lane = threadIdx.x % G;
    long inst = (long)blockIdx.x * GPB + threadIdx.x / G;
    if (inst > a.B - 1) inst = a.B - 1; // surplus groups replicate the last model (identical stores)
    const long rec = inst % a.R;
    const int r = lane < n ? lane : n - 1; // lanes >= n replicate lane n-1
    const bool lead = lane == 0;
    const long T = a.T;

    // per-model constants: lane r holds phi_r, q_r; lane j < N holds loadings[j,:] and obsvar[j]
    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    // row r of Phi (x) Phi and of Q = diag(q) in registers -- except for one model per wavefront
    // (n > 16), where those 4n VGPRs are what stands between one and two resident wavefronts per SIMD:
    // there diag(Phi) is read back from LDS every step (wavefront-uniform address) and Q's row is re-selected
    constexpr bool WIDE = (G == 64);
    double pp[WIDE ? 1 : n], qd[WIDE ? 1 : n];
    constexpr int NP = n + (n & 1); // even: rows of 16-byte pieces
    __shared__ __attribute__((aligned(16))) double lds_phi[WIDE ? (256 / G) * 3 * NP : 1];
    double *phim = lds_phi + (WIDE ? (threadIdx.x / G) * 3 * NP : 0); // diag(Phi)
    double *dvec = phim + (WIDE ? NP : 0);                             // d = P Z_j^T of the current update
    double *dvec2 = dvec;                                              // (loop filter: two buffers, alternating)
    if constexpr (WIDE) {
        phim[r] = phi_r;
        wave_lds_sync();
    } else {
        sfor<0, n>(MK_LAMBDA(c) {
            pp[decltype(c)::value] = phi_r * Gp::template bcast<decltype(c)::value>(phi_r);
            qd[decltype(c)::value] = (decltype(c)::value == r) ? q_r : 0.0;
        });
    }
    const int jr = lane < N ? lane : N - 1;
    double gam[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gam[k] = a.loadings[(rec * N + jr) * K + k];
    const double rvar = a.obsvar ? a.obsvar[rec * N + jr] : 0.0;
    // WIDE and too many loadings to replicate in registers: a wave-private LDS copy read at wavefront-uniform
    // addresses (two 16-byte reads per update for K = 4) instead of 2 K v_readlane per update
    constexpr bool GTAB = WIDE && !HOIST;
    __shared__ __attribute__((aligned(16))) double lds_gam[GTAB ? (256 / G) * N * K : 1];
    double *gtab = lds_gam + (GTAB ? (threadIdx.x / G) * N * K : 0);
    if constexpr (GTAB) {
        if (lane < N) {
#pragma unroll
            for (int k = 0; k < K; ++k) gtab[lane * K + k] = gam[k];
        }
        wave_lds_sync();
    }
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

    // wide loop filter: the K common-factor states (and their phi) replicated in every lane, updated with the gain
    // elements broadcast at the END of an update (next to the rank-one update) -- the innovation of the next update
    // then starts from registers instead of K dependent readlane pairs at the head of its chain
    constexpr bool XREP = WIDE && MK_WIDE_FILTER_LOOP;
    double xk[XREP ? K : 1], phik[XREP ? K : 1];
    if constexpr (XREP) {
        sfor<0, K>(MK_LAMBDA(k) {
            xk[decltype(k)::value] = Gp::template bcast<N + decltype(k)::value>(x);
            phik[decltype(k)::value] = Gp::template bcast<N + decltype(k)::value>(phi_r);
        });
    }

    // outputs: (b, t) lives at block index b*bs + t*ts -- (T, 1) model-major or (1, B) time-major
    constexpr bool RECF = (OUT == 1 || OUT == 3); // filtered moments go to packed records
    MomentPtr oP = moment_ptr<n>(a.Xp, a.Pp, inst * a.bs, a.ts, OUT == 1 ? a.rs : 0, r); // predicted moments
    MomentPtr oF = moment_ptr<n>(a.F, a.Pf, inst * a.bs, a.ts, RECF ? a.rs : 0, r);      // filtered moments
    // OUT == 1 (packed records): the RS - NV pad doubles of both records are written too (whole cache
    // lines): lane l owns pad slot min(l, PADN-1); filtered record: slot 0 = sigma, slot 1 = detf, rest 0
    // (RS / NVO: stride and payload of the record in HBM -- full-square or packed-symmetric)
    constexpr int RS = SYM ? record_stride_sym_c(n) : record_stride_c(n), NVO = SYM ? record_payload_sym(n) : NV;
    constexpr int PADN = RS - NVO;
    const int pslot = lane < PADN ? lane : PADN - 1;
    double *padF = RECF ? a.F + inst * a.bs * RS + NVO + pslot : nullptr;
    double *padP = OUT == 1 ? a.Xp + inst * a.bs * RS + NVO + pslot : nullptr;
    // records leave through wave-private LDS images as whole 16-byte chunks (as in the smoother)
    constexpr bool LDSOUT = MK_FILTER_LDS_STORES && RECF && G == 16;
    using RIO = RecordIO<n, G, SYM>;
    __shared__ __attribute__((aligned(16))) double lds_rec[LDSOUT ? 4 * 2 * RIO::LDS_PER_WAVE : 1];
    double *imgP = lds_rec + (LDSOUT ? (threadIdx.x / 64) * 2 * RIO::LDS_PER_WAVE : 0);
    double *imgF = imgP + (LDSOUT ? RIO::LDS_PER_WAVE : 0);
    const int lane64 = threadIdx.x & 63, gw = lane64 / G;
    typename RIO::Map rmap;
    double *recP = a.Xp, *recF = a.F;
    const long rstep = a.ts * RS;
    if constexpr (LDSOUT) {
        rmap = RIO::make_map(lane64, (long)blockIdx.x * GPB + (threadIdx.x / 64) * RIO::GW, a.B, a.bs);
        RIO::clear_tail(imgP, lane64);
        RIO::clear_tail(imgF, lane64);
        RIO::put_pad(imgP, gw, 0.0, 0.0);
    }
    double pad0 = 0.0, pad1 = 0.0;
    // TAPE: block of (model, step) = N (state tape: N + K) entries [ vector in the observable basis (n) | s0 s1 s2 0 ] of
    // XS = n + 4 doubles (mk_internal.h); lane r < n holds element r of an entry's vector
    [[maybe_unused]] constexpr int XS = tape_xs_c(N, K);
    [[maybe_unused]] double *trec = TAPE ? a.F + inst * a.bs * a.rs : nullptr;
    [[maybe_unused]] const long tstep = a.ts * a.rs;

    // ---- observation stream: tiles of 16 time steps through LDS ----
    // vmcnt retires vector-memory operations IN ORDER, so consuming a load makes the wavefront wait for
    // every older store to be acknowledged by the memory system (~2 us while HBM writes are queued: a
    // per-step observation load cost +0.35 ms even in a pure-store microbenchmark,
    // scripts/ubench/store_pattern2.hip).  The observations are therefore fetched 16 steps at a time
    // (lane l loads the N values of step t0 + l), one tile ahead of use, parked in LDS, and the
    // per-step value comes from LDS (lgkmcnt, independent of the store queue).
    constexpr int TS = 16;
    // The time loop is a nest: tile t0 is parked (registers -> LDS) and the loads of tile t0 + TS are issued
    // at the top of the OUTER iteration, so they are in flight for the 16 steps of the inner loop and no
    // load result is carried through a conditional inside it.  (An earlier version issued the tile two
    // steps before parking it from inside a single loop: the phi copies of the staging registers on the
    // 14 other steps each came with an `s_waitcnt vmcnt(0)` -- a full drain of the store queue EVERY step,
    // 23 % of the filter's cycles in SQ_WAIT_INST_ANY.)
    const int lrow = lane < TS ? lane : TS - 1; // lanes >= TS (G = 64) duplicate row TS-1
    constexpr bool OV2 = (N % 2 == 0);          // 16-byte row pieces when rows are 16-byte aligned
    constexpr int ONC = OV2 ? N / 2 : N;
    using ochunk_t = typename std::conditional<OV2, v2d, double>::type;
    __shared__ __attribute__((aligned(16))) double lds_obs[(256 / G) * TS * N];
    double *otile = lds_obs + (threadIdx.x / G) * TS * N; // this model's tile
    const double *obase = a.obs + rec * a.obs_bs * N;
    const long ostep = a.obs_ts * N;
    ochunk_t oreg[WIDE ? 1 : ONC];
    // WIDE (one model per wavefront): a step takes ~10 us, the once-per-tile wait for the store queue is
    // noise, and the N staging registers held across two steps are not: fetch and park in one go
    long otr = 0;
    auto obs_issue = [&](long t0) __attribute__((always_inline)) { // HBM -> registers, row min(t0+lane, T-1)
        long tr = t0 + lrow;
        if (tr > T - 1) tr = T - 1;
        if constexpr (WIDE) {
            otr = tr;
        } else {
            const ochunk_t *src = reinterpret_cast<const ochunk_t *>(obase + tr * ostep);
#pragma unroll
            for (int i = 0; i < ONC; ++i) oreg[i] = src[i];
        }
    };
    auto obs_park = [&]() __attribute__((always_inline)) { // registers -> the LDS tile
        ochunk_t *dst = reinterpret_cast<ochunk_t *>(otile + lrow * N);
        wave_lds_sync(); // the previous tile's reads are complete
        if constexpr (WIDE) {
            const ochunk_t *src = reinterpret_cast<const ochunk_t *>(obase + otr * ostep);
#pragma unroll
            for (int i = 0; i < ONC; ++i) dst[i] = src[i];
        } else {
#pragma unroll
            for (int i = 0; i < ONC; ++i) dst[i] = oreg[i];
        }
        wave_lds_sync();
    };
    obs_issue(0);
    double ynext = 0.0;

    double sum_sig = 0.0, sum_det = 0.0;
    double run_mant = 1.0; // !BOOK: prod of f over the counted steps, normalised
    long run_exp = 0;
    long nobs = 0, sc = 0;
    double fmin_seen = 1.0;

    for (long t0 = 0; t0 < T; t0 += TS) {
    obs_park();           // tile t0: loads issued one outer iteration (16 steps) ago
    obs_issue(t0 + TS);   // tile t0 + TS (rows clamped to T-1)
    ynext = otile[jr];
    const long tend = t0 + TS < T ? t0 + TS : T;
    for (long t = t0; t < tend; ++t) {
        const double y = ynext;
        {
            // next step's observation from LDS now (latency hidden by this step); the last step of a tile
            // re-reads its own row, the outer loop supplies the next tile's first
            const int s1 = (int)(t - t0) + 1;
            ynext = otile[(s1 < TS ? s1 : TS - 1) * N + jr];
        }
        // which series are observed at this step (NaN / inf = missing, kalmanfilter.py:657)
        const unsigned long long ball = __ballot(lane < N && isfinite(y));
        const auto vm = Gp::group_bits(ball);

        // ---- predict (:318-331; Phi diagonal) ----
        x = phi_r * x;
        if constexpr (XREP) {
#pragma unroll
            for (int k = 0; k < K; ++k) xk[k] = phik[k] * xk[k];
        }
        if constexpr (WIDE) {
            int rv = r; // opaque copies: keeps the n selects inside the loop (hoisted, they are 2n VGPRs)
            double qv = q_r;
            asm volatile("" : "+v"(rv), "+v"(qv));
            double phc[n];
            load_row<n>(phim, phc);
#pragma unroll
            for (int c = 0; c < n; ++c) P[c] = fma(P[c] * phi_r, phc[c], c == rv ? qv : 0.0);
        } else {
#pragma unroll
            for (int c = 0; c < n; ++c) P[c] = fma(P[c], pp[c], qd[c]);
        }
        if constexpr (LDSOUT && OUT == 1) {
            RIO::put(imgP, gw, r, x, P);
        } else if constexpr (OUT == 1) {
            *oP.vec = x;               // :332
            if constexpr (SYM) store_cols_sym<n>(oP.mat - r, P, r);
            else store_cols<n>(oP.mat, P);  // :333
            *padP = 0.0;
            oP.advance_nn(1);
            padP += a.ts * RS;
        } else if constexpr (OUT == 2) {
            if (oP.vec) *oP.vec = x;
            if (oP.mat) store_cols<n>(oP.mat, P);
            oP.advance(1);
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
                Gp::template fmac<N + kk, true>(vl, x, gam[kk]); // vl -= x_{N+k} * gam_k
            });
            const double v = Gp::template bcast<j>(vl);
            // d = P Z_j^T : lane r computes d_r from its own row (:349-357)
            double dr = P[j];
            sfor<0, K>(MK_LAMBDA(k) {
                constexpr int kk = decltype(k)::value;
                double g;
                if constexpr (HOIST) g = Gh[j][kk];
                else if constexpr (GTAB) g = gtab[j * K + kk];
                else g = Gp::template bcast<j>(gam[kk]);
                dr = fma(P[N + kk], g, dr);
            });
            // innovation variance f = R_j + Z_j d (:359-362), formed at lane j from d_j, d_{N+k}
            double fl = rvar + dr;
            if constexpr (G == 16) dpp_pin(dr);
            sfor<0, K>(MK_LAMBDA(k) {
                constexpr int kk = decltype(k)::value;
                Gp::template fmac<N + kk, false>(fl, dr, gam[kk]); // fl += d_{N+k} * gam_k
            });
            const double f = Gp::template bcast<j>(fl);
            const double rf = rcp_nr(f);
            const double kr = dr * rf; // Kalman gain element r (:364-366)
            // P -= k k^T f (:368-372): P[r][c] -= d_c * k_r, d_c broadcast from lane c
            if constexpr (WIDE) {
                // n broadcasts through readlane are 2n VALU + n FMA issues; through LDS (every lane stores
                // its d_r, then reads the whole vector at a wavefront-uniform address) the VALU sees only
                // the n FMAs -- these kernels are VALU-issue bound (one f64 wave instruction per 4 cycles)
                dvec[r] = dr;
                wave_lds_sync();
                double dc[n];
                load_row<n>(dvec, dc);
#pragma unroll
                for (int c = 0; c < n; ++c) P[c] = fma(-dc[c], kr, P[c]);
            } else {
                Gp::template axpy_col<0, n, true, n>(P, dr, kr);
            }
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
        if constexpr (WIDE && MK_WIDE_FILTER_LOOP) {
            // One model per wavefront: the mask of observed series is wavefront-uniform, so the updates run as a
            // RUN-TIME loop over its set bits with ONE update body (the series index lives in an SGPR: v_readlane with a
            // scalar lane select, loadings at a scalar LDS offset, P[j] through a uniform switch) instead of 2 x 32
            // unrolled bodies -- 64 KB of code whose merged live ranges spilled the covariance row around the masked
            // path (PMC: 399 GB of traffic per launch against 89 GB algorithmic at configs[3]).
            unsigned long long m = (unsigned long long)vm;
            int nupd = 0;
            while (m) {
                const int j = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(m));
                m &= m - 1;
                double vl = y - x, vl2 = 0.0;
                sfor<0, K>(MK_LAMBDA(k) { // vl = y_l - x_l - sum_k gam_k x_{N+k}, two chains, factor states from registers
                    constexpr int kk = decltype(k)::value;
                    if constexpr (kk % 2 == 0) vl = fma(-gam[kk], xk[kk], vl);
                    else vl2 = fma(-gam[kk], xk[kk], vl2);
                });
                vl += vl2;
                const double v = readlane_f64(vl, j);
                double dr = 0.0;
                switch (j) {
#define MK_CASE_P(c) \
    case c:          \
        if constexpr (c < N) dr = P[c]; \
        break;
                    MK_CASE_P(0) MK_CASE_P(1) MK_CASE_P(2) MK_CASE_P(3) MK_CASE_P(4) MK_CASE_P(5) MK_CASE_P(6) MK_CASE_P(7)
                    MK_CASE_P(8) MK_CASE_P(9) MK_CASE_P(10) MK_CASE_P(11) MK_CASE_P(12) MK_CASE_P(13) MK_CASE_P(14) MK_CASE_P(15)
                    MK_CASE_P(16) MK_CASE_P(17) MK_CASE_P(18) MK_CASE_P(19) MK_CASE_P(20) MK_CASE_P(21) MK_CASE_P(22) MK_CASE_P(23)
                    MK_CASE_P(24) MK_CASE_P(25) MK_CASE_P(26) MK_CASE_P(27) MK_CASE_P(28) MK_CASE_P(29) MK_CASE_P(30) MK_CASE_P(31)
                    MK_CASE_P(32) MK_CASE_P(33) MK_CASE_P(34) MK_CASE_P(35) MK_CASE_P(36) MK_CASE_P(37) MK_CASE_P(38) MK_CASE_P(39)
                    MK_CASE_P(40) MK_CASE_P(41) MK_CASE_P(42) MK_CASE_P(43) MK_CASE_P(44) MK_CASE_P(45) MK_CASE_P(46) MK_CASE_P(47)
                    MK_CASE_P(48) MK_CASE_P(49) MK_CASE_P(50) MK_CASE_P(51) MK_CASE_P(52) MK_CASE_P(53) MK_CASE_P(54) MK_CASE_P(55)
                    MK_CASE_P(56) MK_CASE_P(57) MK_CASE_P(58) MK_CASE_P(59) MK_CASE_P(60) MK_CASE_P(61) MK_CASE_P(62) MK_CASE_P(63)
#undef MK_CASE_P
                default: break;
                }
                const double *gj = (GTAB ? gtab : dvec) + j * K; // loadings of series j (wavefront-uniform address)
                sfor<0, K>(MK_LAMBDA(k) {
                    constexpr int kk = decltype(k)::value;
                    double g;
                    if constexpr (GTAB) g = gj[kk];
                    else g = readlane_f64(gam[kk], j);
                    dr = fma(P[N + kk], g, dr);
                });
                double *dv = dvec2 + (nupd & 1) * NP; // two buffers: no second fence per update
                dv[r] = dr;
                double fl = rvar + dr, fl2 = 0.0;
                double dk[K];
                sfor<0, K>(MK_LAMBDA(k) { dk[decltype(k)::value] = Gp::template bcast<N + decltype(k)::value>(dr); });
                sfor<0, K>(MK_LAMBDA(k) {
                    constexpr int kk = decltype(k)::value;
                    if constexpr (kk % 2 == 0) fl = fma(dk[kk], gam[kk], fl);
                    else fl2 = fma(dk[kk], gam[kk], fl2);
                });
                fl += fl2;
                const double f = readlane_f64(fl, j);
                const double rf = rcp_nr(f);
                const double kr = dr * rf;
                if constexpr (OUT == 3 && !SYM) {
                    // the recording pass of the adjoint gradient: (d, 1/f, v) of this update into slot nupd of the step's block of the
                    // update tape -- what the backward walk would otherwise recompute from the filtered record of step t - 1 (round 6)
                    if (a.upd) {
                        double *ub = a.upd + (inst * a.bs + t * a.ts) * a.us + (long)nupd * adjoint_update_slot_c(N, K);
                        ub[r] = dr;                                  // one 8 n-byte run per update
                        if (lead) *reinterpret_cast<v2d *>(ub + NP) = v2d{rf, v};
                    }
                }
                wave_lds_sync();
                {
                    double dc[n];
                    load_row<n>(dv, dc);
#pragma unroll
                    for (int c = 0; c < n; ++c) P[c] = fma(-dc[c], kr, P[c]);
                }
                if constexpr (TAPE) {
                    // tape entry of the observed series j: [ kt = T k | v/f | 1/f | y_j | 0 ], the gain in the observable basis
                    // (kt_l = k_l + sum_k g_lk k_{N+k} on the series lanes, the factor gains themselves on the factor lanes:
                    // one 8 n-byte run), as filter_split_kernel writes it
                    double kt = 0.0;
                    sfor<0, K>(MK_LAMBDA(k) { kt = fma(gam[decltype(k)::value], Gp::template bcast<N + decltype(k)::value>(kr), kt); });
                    kt = lane < N ? kr + kt : kr;
                    trec[j * XS + r] = kt;
                    const double yj = readlane_f64(y, j);
                    if (lead) {
                        double *sd = trec + j * XS + n;
                        sd[0] = v * rf;
                        sd[1] = rf;
                        sd[2] = yj;
                        sd[3] = 0.0;
                    }
                }
                sfor<0, K>(MK_LAMBDA(k) { // factor-state replicas: gain elements of lanes N+k (off the critical path)
                    constexpr int kk = decltype(k)::value;
                    xk[kk] = fma(Gp::template bcast<N + kk>(kr), v, xk[kk]);
                });
                x = fma(kr, v, x);
                sigma = fma(v * v, rf, sigma);
                fmant *= f;
                if ((++nupd & 3) == 0) { // keep the product of innovation variances normalised (exact: powers of two)
                    fexp += __builtin_amdgcn_frexp_exp(fmant);
                    fmant = __builtin_amdgcn_frexp_mant(fmant);
                }
                fmin_seen = min_f64(fmin_seen, f);
            }
            fexp += __builtin_amdgcn_frexp_exp(fmant);
            fmant = __builtin_amdgcn_frexp_mant(fmant);
        } else if (ball == Gp::full_mask(N)) { // every model of this wavefront observes all N series: no masking
            sfor<0, N>(update);
        } else {
            sfor<0, N>(MK_LAMBDA(jc) {
                if ((vm >> decltype(jc)::value) & 1) { // uniform within the model's lane group
                    update(jc);
                } else if constexpr ((decltype(jc)::value & 3) == 3 || decltype(jc)::value == N - 1) {
                    fexp += __builtin_amdgcn_frexp_exp(fmant); // keep the product normalised
                    fmant = __builtin_amdgcn_frexp_mant(fmant);
                }
            });
        }

        const int cnt = __popcll((unsigned long long)vm);
        double pad = 0.0; // records: what this lane writes into its pad slot of the filtered record
        if (cnt > 0) { // :380-382 compressed bookkeeping
            if constexpr (BOOK) {
                int le;
                const double lm = log_mant(fmant, le);
                const double detf = fma((double)(fexp + le), kLn2, lm);
                if constexpr (RECF) {
                    // compressed entry sc lives in the pad of filtered record sc; sc == t unless an earlier
                    // step of this model was empty (then: one scattered 16-byte store, rare)
                    if (sc == t) {
                        pad = pslot == 0 ? sigma : (pslot == 1 ? detf : 0.0);
                        pad0 = sigma;
                        pad1 = detf;
                    } else if (lead && a.sigmas)
                        *reinterpret_cast<v2d *>(a.F + (inst * a.bs + sc * a.ts) * RS + NVO) = v2d{sigma, detf};
                } else {
                    if (a.sigmas && lead) a.sigmas[(inst * a.bs + sc * a.ts) * a.sig_stride] = sigma;
                    if (a.detfs && lead) a.detfs[(inst * a.bs + sc * a.ts) * a.sig_stride] = detf;
                }
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

        if constexpr (TAPE) {
            // entries of the series NOT observed at this step, from the filtered moments (mk_split.hip writes the same):
            //     [ pt = T Pf z_u' (n) | z_u x_f | z_u Pf z_u' | NaN | 0 ],   d = Pf z_u' as in an update (no rank-one update),
            //     pt_l = d_l + sum_k g_lk d_{N+k} on the series lanes (pt_u IS z_u Pf z_u'), d_{N+k} on the factor lanes
            const double qnan = __builtin_nan("");
            if (a.tape == 2) {
                // STATE tape: entry N + k = column N + k of T Pf T' -- [ Pf[l][N+k] + sum_k' g_lk' Pf[N+k'][N+k] (series lanes),
                // Pf[N+k'][N+k] (factor lanes) | x_f[N+k] | Pf[N+k][N+k] | NaN | 0 ]
                sfor<0, K>(MK_LAMBDA(kc) {
                    constexpr int k = decltype(kc)::value;
                    double qk = 0.0;
                    sfor<0, K>(MK_LAMBDA(k2) { qk = fma(gam[decltype(k2)::value], Gp::template bcast<N + decltype(k2)::value>(P[N + k]), qk); });
                    qk = lane < N ? P[N + k] + qk : P[N + k];
                    trec[(N + k) * XS + r] = qk;
                    const double pkk = Gp::template bcast<N + k>(P[N + k]);
                    if (lead) {
                        double *sd = trec + (N + k) * XS + n;
                        sd[0] = xk[k];
                        sd[1] = pkk;
                        sd[2] = qnan;
                        sd[3] = 0.0;
                    }
                });
            }
            unsigned long long um = ~(unsigned long long)vm & Gp::full_mask(N);
            double yh = x; // the filtered observable z_l x_f of every series lane
            sfor<0, K>(MK_LAMBDA(k) { yh = fma(gam[decltype(k)::value], xk[decltype(k)::value], yh); });
            while (um) {
                const int u = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(um));
                um &= um - 1;
                double dr = 0.0;
                switch (u) {
#define MK_CASE_PU(c) \
    case c:           \
        if constexpr (c < N) dr = P[c]; \
        break;
                    MK_CASE_PU(0) MK_CASE_PU(1) MK_CASE_PU(2) MK_CASE_PU(3) MK_CASE_PU(4) MK_CASE_PU(5) MK_CASE_PU(6) MK_CASE_PU(7) MK_CASE_PU(8) MK_CASE_PU(9) MK_CASE_PU(10) MK_CASE_PU(11) MK_CASE_PU(12) MK_CASE_PU(13) MK_CASE_PU(14) MK_CASE_PU(15) MK_CASE_PU(16) MK_CASE_PU(17) MK_CASE_PU(18) MK_CASE_PU(19) MK_CASE_PU(20) MK_CASE_PU(21) MK_CASE_PU(22) MK_CASE_PU(23) MK_CASE_PU(24) MK_CASE_PU(25) MK_CASE_PU(26) MK_CASE_PU(27) MK_CASE_PU(28) MK_CASE_PU(29) MK_CASE_PU(30) MK_CASE_PU(31) MK_CASE_PU(32) MK_CASE_PU(33) MK_CASE_PU(34) MK_CASE_PU(35) MK_CASE_PU(36) MK_CASE_PU(37) MK_CASE_PU(38) MK_CASE_PU(39) MK_CASE_PU(40) MK_CASE_PU(41) MK_CASE_PU(42) MK_CASE_PU(43) MK_CASE_PU(44) MK_CASE_PU(45) MK_CASE_PU(46) MK_CASE_PU(47) MK_CASE_PU(48) MK_CASE_PU(49) MK_CASE_PU(50) MK_CASE_PU(51) MK_CASE_PU(52) MK_CASE_PU(53) MK_CASE_PU(54) MK_CASE_PU(55) MK_CASE_PU(56) MK_CASE_PU(57) MK_CASE_PU(58) MK_CASE_PU(59) MK_CASE_PU(60) MK_CASE_PU(61) MK_CASE_PU(62) MK_CASE_PU(63)
#undef MK_CASE_PU
                default: break;
                }
                const double *gu = gtab + u * K;
                sfor<0, K>(MK_LAMBDA(k) { dr = fma(P[N + decltype(k)::value], gu[decltype(k)::value], dr); });
                double pt = 0.0;
                sfor<0, K>(MK_LAMBDA(k) { pt = fma(gam[decltype(k)::value], Gp::template bcast<N + decltype(k)::value>(dr), pt); });
                pt = lane < N ? dr + pt : dr;
                trec[u * XS + r] = pt;
                const double s0 = readlane_f64(yh, u), s1 = readlane_f64(pt, u);
                if (lead) {
                    double *sd = trec + u * XS + n;
                    sd[0] = s0;
                    sd[1] = s1;
                    sd[2] = qnan;
                    sd[3] = 0.0;
                }
            }
            trec += tstep;
        }

        if constexpr (LDSOUT) {
            RIO::put(imgF, gw, r, x, P);
            RIO::put_pad(imgF, gw, pad0, pad1);
            pad0 = pad1 = 0.0;
            if constexpr (OUT == 1) {
                RIO::emit2(imgP, recP, imgF, recF, rmap);
                recP += rstep;
            } else {
                RIO::emit(imgF, recF, rmap);
            }
            recF += rstep;
        } else if constexpr (RECF) {
            *oF.vec = x;               // :389
            if constexpr (SYM) store_cols_sym<n>(oF.mat - r, P, r);
            else store_cols<n>(oF.mat, P);  // :390
            *padF = pad;               // sigma | detf | zeros
            oF.advance_nn(1);
            padF += a.ts * RS;
        } else if constexpr (OUT == 2) {
            if (oF.vec) *oF.vec = x;
            if (oF.mat) store_cols<n>(oF.mat, P);
            oF.advance(1);
        }
    }
    }

    // zero tail of the compressed arrays (np.zeros init, :307-308); record pads were written as zeros
    if (BOOK && !RECF) {
        for (long i = sc + lane; i < T; i += G) {
            if (a.sigmas) a.sigmas[(inst * a.bs + i * a.ts) * a.sig_stride] = 0.0;
            if (a.detfs) a.detfs[(inst * a.bs + i * a.ts) * a.sig_stride] = 0.0;
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
// ---------------------------------------------------------------- packed-record I/O through LDS (smoother)
// The smoother both reads and writes a record per step; moving whole 16-byte chunks of the
// wavefront's GW adjacent records (lane l <-> chunks l, l+64, ...) and transposing through a
// wave-private LDS image measured faster there (1.65 ms) than symmetric column runs (2.1 ms).
// SYM (mk_outputs.flags & MK_OUT_PACKED_SYM): the record in HBM is PACKED-SYMMETRIC,
//     [ mean(n) | upper triangle by rows: (0,0..n-1) (1,1..n-1) ... | sigma, detf | zero pad ]   RSO doubles,
// n + n(n+1)/2 (+2) doubles padded to 128 bytes (n = 10: 640 B instead of 896 B).  The LDS image keeps the full
// square layout (the lanes still put / read whole rows); only the chunk maps change: a 16-byte chunk of the packed
// record is two doubles gathered from (scattered to) two image positions, and a loaded off-diagonal element is
// written to its mirror position as well.
// Record traffic is a pure stream (every byte written once, read at most once by the next kernel): the 16-byte record
// stores / loads are marked non-temporal (MK_NT_IO=0 builds the plain form; measured at configs[1] / 8192 models:
// filter 1.44 -> 1.39 ms / 2.55 -> 2.53 ms, smoother 1.76 -> 1.73 ms / 3.2 -> 3.2 ms).
#ifndef MK_NT_IO
#define MK_NT_IO 1
#endif
template <class V>
__device__ __forceinline__ void rec_store(V *p, V v)
{
#if MK_NT_IO
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <class V>
__device__ __forceinline__ V rec_load(const V *p)
{
#if MK_NT_IO
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

template <int n, int G, bool SYM>
struct RecordIO {
    static constexpr int NV = n + n * n;                    // payload doubles of the IMAGE (full square)
    static constexpr int RS = record_stride_c(n);           // image stride per model, doubles
    static constexpr int NVO = SYM ? record_payload_sym(n) : NV;   // payload doubles of the HBM record
    static constexpr int RSO = SYM ? record_stride_sym_c(n) : RS;  // HBM record stride in doubles (128-byte multiple)
    static constexpr int RC = RSO / 2;                      // 16-byte chunks per record
    static constexpr int GW = 64 / G;                       // models per wavefront
    static constexpr int CW = GW * RC;                      // chunks per wavefront-step
    static constexpr int PER = (CW + 63) / 64;              // chunks per lane
    static constexpr int SPARE = GW * RS;                   // SYM: two spare doubles behind the image (zeros / dump)
    static constexpr int LDS_PER_WAVE = GW * RS + (SYM ? 2 : 0); // doubles of one wavefront image
    using chunk_t = v2d;

    struct Map {          // loop-invariant per-lane addressing of its PER chunks
        long off[PER];    // element offset of the chunk inside the record ARRAY at t = 0
        int lq[PER];      // !SYM: chunk index inside the wavefront's LDS image
        int a0[SYM ? PER : 1], a1[SYM ? PER : 1]; // SYM: image position of the chunk's two doubles (store side)
        int b0[SYM ? PER : 1], b1[SYM ? PER : 1]; // SYM: the same for the load side (pads go to the dump slot)
        int m0[SYM ? PER : 1], m1[SYM ? PER : 1]; // SYM: mirror positions (c,r) of loaded off-diagonal elements
    };
    // packed index p of a record -> position inside ONE model's image block; kind: 0 payload, 1 sigma/detf, 2 zero pad
    static __device__ __forceinline__ void locate(int p, int &pos, int &mirror, int &kind)
    {
        kind = 0;
        if (p < n) {
            pos = mirror = p;
        } else if (p < NVO) {
            int u = p - n, r = 0;
            while (u >= n - r) {
                u -= n - r;
                ++r;
            }
            const int c = r + u;
            pos = n + r * n + c;
            mirror = n + c * n + r;
        } else if (p < NVO + 2) {
            pos = mirror = NV + (p - NVO);
            kind = 1;
        } else {
            pos = mirror = 0;
            kind = 2;
        }
    }
    static __device__ __forceinline__ Map make_map(int lane64, long inst0, long B, long bs)
    {
        Map mp;
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            int q = lane64 + 64 * m;
            if (q > CW - 1) q = CW - 1; // surplus lanes duplicate the last chunk (same bytes, same address)
            const int g = q / RC, w = q - g * RC;
            long ig = inst0 + g;
            if (ig > B - 1) ig = B - 1; // surplus groups replicate the last model
            mp.off[m] = ig * bs * RSO + 2 * w;
            mp.lq[m] = q;
            if constexpr (SYM) {
                int pos, mir, kind;
                locate(2 * w, pos, mir, kind);
                mp.a0[m] = kind == 2 ? SPARE : g * RS + pos;
                mp.b0[m] = kind == 0 ? g * RS + pos : SPARE;
                mp.m0[m] = kind == 0 ? g * RS + mir : SPARE;
                locate(2 * w + 1, pos, mir, kind);
                mp.a1[m] = kind == 2 ? SPARE + 1 : g * RS + pos;
                mp.b1[m] = kind == 0 ? g * RS + pos : SPARE + 1;
                mp.m1[m] = kind == 0 ? g * RS + mir : SPARE + 1;
            }
        }
        return mp;
    }
    // registers -> this model's record in the wavefront's LDS image
    static __device__ __forceinline__ void put(double *img, int g, int r, double x, const double (&row)[n])
    {
        img[g * RS + r] = x;
        store_row<n>(img + g * RS + n + r * n, row);
    }
    static __device__ __forceinline__ void put_pad(double *img, int g, double s0, double s1)
    {
        *reinterpret_cast<v2d *>(img + g * RS + NV) = v2d{s0, s1};
    }
    // zero the pad tail [NV+2, RS) of every record once (it is never rewritten); SYM: and the two spare doubles
    static __device__ __forceinline__ void clear_tail(double *img, int lane64)
    {
        for (int i = lane64; i < GW * (RS - NV - 2); i += 64) {
            const int g = i / (RS - NV - 2 > 0 ? RS - NV - 2 : 1), w = i % (RS - NV - 2 > 0 ? RS - NV - 2 : 1);
            img[g * RS + NV + 2 + w] = 0.0;
        }
        if constexpr (SYM)
            if (lane64 < 2) img[SPARE + lane64] = 0.0;
    }
    static __device__ __forceinline__ chunk_t gather(const double *img, const Map &mp, int m)
    {
        if constexpr (SYM) return chunk_t{img[mp.a0[m]], img[mp.a1[m]]};
        else return reinterpret_cast<const chunk_t *>(img)[mp.lq[m]];
    }
    // LDS image -> HBM records of this step (base = record array + t*ts*RSO)
    static __device__ __forceinline__ void emit(const double *img, double *base, const Map &mp)
    {
        chunk_t tmp[PER];
        gather_all(img, mp, tmp);
        store_all(base, mp, tmp);
    }
    // the two halves of emit, for callers that put useful work between the LDS reads and the stores they feed
    static __device__ __forceinline__ void gather_all(const double *img, const Map &mp, chunk_t (&tmp)[PER])
    {
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < PER; ++m) tmp[m] = gather(img, mp, m);
    }
    static __device__ __forceinline__ void store_all(double *base, const Map &mp, const chunk_t (&tmp)[PER])
    {
#pragma unroll
        for (int m = 0; m < PER; ++m) rec_store(reinterpret_cast<chunk_t *>(base + mp.off[m]), tmp[m]);
    }
    static __device__ __forceinline__ void emit2(const double *img0, double *base0, const double *img1, double *base1,
                                                 const Map &mp)
    {
        wave_lds_sync();
        chunk_t t0[PER], t1[PER];
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            t0[m] = gather(img0, mp, m);
            t1[m] = gather(img1, mp, m);
        }
#pragma unroll
        for (int m = 0; m < PER; ++m) rec_store(reinterpret_cast<chunk_t *>(base0 + mp.off[m]), t0[m]);
#pragma unroll
        for (int m = 0; m < PER; ++m) rec_store(reinterpret_cast<chunk_t *>(base1 + mp.off[m]), t1[m]);
    }
    // HBM records -> registers (issued one step ahead of use)
    static __device__ __forceinline__ void load_issue(const double *base, const Map &mp, chunk_t (&buf)[PER])
    {
#pragma unroll
        for (int m = 0; m < PER; ++m) buf[m] = rec_load(reinterpret_cast<const chunk_t *>(base + mp.off[m]));
    }
    // registers -> LDS image (the wavefront's records of one step, full-square layout)
    static __device__ __forceinline__ void to_image(double *img, const chunk_t (&buf)[PER], const Map &mp)
    {
        wave_lds_sync();
        if constexpr (SYM) {
#pragma unroll
            for (int m = 0; m < PER; ++m) {
                img[mp.b0[m]] = buf[m].x;
                img[mp.b1[m]] = buf[m].y;
                img[mp.m0[m]] = buf[m].x; // mirror of an off-diagonal element; means and diagonals rewrite themselves
                img[mp.m1[m]] = buf[m].y;
            }
        } else {
            chunk_t *l = reinterpret_cast<chunk_t *>(img);
#pragma unroll
            for (int m = 0; m < PER; ++m) l[mp.lq[m]] = buf[m];
        }
        wave_lds_sync();
    }
    // registers -> LDS image -> x element r and row r of this model
    static __device__ __forceinline__ void load_finish(double *img, const chunk_t (&buf)[PER], const Map &mp, int g,
                                                       int r, double &x, double (&row)[n])
    {
        to_image(img, buf, mp);
        x = img[g * RS + r];
        load_row<n>(img + g * RS + n + r * n, row);
    }
};

// EPI: 0 = smoothed records only; 1 = + fused projection (sim_means / sim_vars, section f2); 2 = + state means and
// VARIANCES (mk_outputs.flags & MK_OUT_VAR_ONLY: what get_state_means / get_state_variances / get_state consume)
template <int N, int K, int G, int EPI, bool SYM>
__global__ void __launch_bounds__(256) smoother_record_kernel(SmootherArgs a)
{
    constexpr bool PROJ = (EPI == 1), VAR = (EPI == 2);
    constexpr int n = N + K;
    static_assert(n <= G, "state dimension must fit the lane group");
    using Gp = Group<G>;
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    long inst = (long)blockIdx.x * GPB + threadIdx.x / G;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1; // surplus groups replicate the last model (identical stores)
    const int r = lane < n ? lane : n - 1; // lanes >= n replicate lane n-1
    const long T = a.T;

    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    double phic[n], qd[n]; // diag(Phi) replicated; row r of Q = diag(q)
    sfor<0, n>(MK_LAMBDA(c) {
        phic[decltype(c)::value] = Gp::template bcast<decltype(c)::value>(phi_r);
        qd[decltype(c)::value] = (decltype(c)::value == r) ? q_r : 0.0;
    });

    // REC: packed records (RecordIO): one load image and one store image per wavefront;
    // !REC: dense arrays with arbitrary strides (BlockIO), one staging buffer per model
    constexpr bool REC = true;
    using RIO = RecordIO<n, G, SYM>;
    constexpr int LDS_DOUBLES = 4 * 2 * RIO::LDS_PER_WAVE;
    __shared__ __attribute__((aligned(16))) double lds_io[LDS_DOUBLES];
    const int lane64 = threadIdx.x & 63;
    const int gw = lane64 / G;
    double *imgL = lds_io + (threadIdx.x / 64) * 2 * RIO::LDS_PER_WAVE;
    double *imgS = imgL + RIO::LDS_PER_WAVE;
    typename RIO::Map rmap;
    if constexpr (REC) {
        rmap = RIO::make_map(lane64, (long)blockIdx.x * GPB + (threadIdx.x / 64) * RIO::GW, a.B, a.bs);
        RIO::clear_tail(imgS, lane64);
        RIO::put_pad(imgS, gw, 0.0, 0.0);
    }
    typename RIO::chunk_t prer[RIO::PER]; // chunks of the next filtered record

    // addressing: (b, t) at block index b*bs + t*ts
    const long rstep = a.ts * RIO::RSO;
    const double *recF = a.F + (T - 1) * rstep; // record array positioned at step t
    double *recS = a.S ? a.S + (T - 1) * rstep : nullptr; // smoothed records are optional when projecting

    // PROJ: fused projection epilogue (simulate) of the smoothed moments, lanes j < N
    const long rec_id = inst % a.R;
    const int jr = lane < N ? lane : N - 1;
    double gam[K], pscale = 1.0, poffset = 0.0;
    double *pM = nullptr, *pV = nullptr;
    if constexpr (PROJ) {
#pragma unroll
        for (int k = 0; k < K; ++k) gam[k] = a.loadings[(rec_id * N + jr) * K + k];
        if (a.scale) pscale = a.scale[rec_id * N + jr];
        if (a.offset) poffset = a.offset[rec_id * N + jr];
        const long pidx = (inst * a.bs + (T - 1) * a.ts) * N + jr;
        pM = a.sim_means ? a.sim_means + pidx : nullptr;
        pV = a.sim_vars ? a.sim_vars + pidx : nullptr;
    }
    double *sM = nullptr, *sV = nullptr; // VAR: state means / variances [., n]
    if constexpr (VAR) {
        const long sidx = (inst * a.bs + (T - 1) * a.ts) * n + r;
        sM = a.state_means + sidx;
        sV = a.state_vars + sidx;
    }

    // fetch helpers: issue the HBM loads of one step / turn them into (x_r, row r)
    auto issue = [&](double &xnext) __attribute__((always_inline)) {
        recF -= rstep;
        RIO::load_issue(recF, rmap, prer);
    };
    auto finish = [&](double &xv, double(&row)[n]) __attribute__((always_inline)) {
        RIO::load_finish(imgL, prer, rmap, gw, r, xv, row);
    };
    // DEFER (records only, EPI = 0 -- the full-output path): the smoothed record of a step goes into the store image at
    // the END of its iteration and leaves it in the MIDDLE of the next one, and the next filtered record is transposed
    // through LDS right after the factorisation: the images' write -> read round trips and the reads' latency (~250 cycles
    // a step each, exposed with one wavefront per SIMD) hide behind VALU work, and the stores are still a full iteration
    // older than the next consumed load (the vmcnt rule below).  With an epilogue (projection / variances) the extra
    // live registers of that schedule cost more than the hidden latency is worth (measured: 1.56 -> 1.63 ms): those
    // variants emit at the end of the iteration.
    constexpr bool DEFER = (EPI == 0);
    auto store = [&](double xv, const double(&row)[n]) __attribute__((always_inline)) {
        if (recS) {
            RIO::put(imgS, gw, r, xv, row);
            if constexpr (!DEFER) {
                RIO::emit(imgS, recS, rmap);
                recS -= rstep;
            }
        }
        if constexpr (PROJ) {
            double mean, var;
            project<N, K, G>(xv, row, gam, pscale, poffset, lane, mean, var);
            if (lane < N && live) { // factor rows hold other values: mask (one exec region per step)
                if (pM) *pM = mean;
                if (pV) *pV = var;
            }
            if (pM) pM -= a.ts * N;
            if (pV) pV -= a.ts * N;
        }
        if constexpr (VAR) {
            double diag = 0.0;
            sfor<0, n>(MK_LAMBDA(c) { diag = (decltype(c)::value == lane) ? row[decltype(c)::value] : diag; });
            if (lane < n && live) {
                *sM = xv;
                *sV = diag;
            }
            sM -= a.ts * n;
            sV -= a.ts * n;
        }
    };
    typename RIO::chunk_t etmp[RIO::PER]; // chunks of the record being emitted

    // last step: smoothed = filtered (:450-451)
    double xs, Psn[n];
    RIO::load_issue(recF, rmap, prer);
    finish(xs, Psn);
    store(xs, Psn);
    double pivmin = 1.0;

    // Software pipeline: at the top of iteration t the ROWS of Pf[t] (Pfc) and F[t] (xfc) are already in
    // registers and the 16-byte chunks of step t-1 are in flight from HBM; their LDS transposition is issued
    // in the middle of the iteration so that its latency hides behind the Ps sweep, and the HBM loads for
    // t-2 are issued right after it.
    double Pfc[n], xfc = 0.0, xfn = 0.0;
    if (T >= 2) {
        issue(xfc);
        finish(xfc, Pfc);
        if (T >= 3) issue(xfn);
    }

    for (long t = T - 2; t >= 0; --t) {
        // W = Pf Phi (column scaling); A = Pp[t+1] = Phi Pf Phi + Q (row r); D = Ps[t+1] - Pp[t+1]
        double A[n], z[n], D[n];
#pragma unroll
        for (int c = 0; c < n; ++c) {
            z[c] = Pfc[c] * phic[c]; // W, the right-hand side of the solve
            A[c] = fma(phi_r, z[c], qd[c]);
            D[c] = Psn[c] - A[c];
            if constexpr (DEFER) Psn[c] = Pfc[c]; // the accumulator of the second product; Pfc is refilled in mid-iteration
        }
        double delta = xs - phi_r * xfc; // xs[t+1] - Xp[t+1]; formed early so that it is "old" when DPP-read
        if constexpr (G == 16) dpp_pin(delta);

        if constexpr (DEFER)
            if (recS) RIO::gather_all(imgS, rmap, etmp); // the record of step t+1, put there at the end of its iteration

        // ---- A = L D L^T, right-looking; lane c ends up holding L(c, j) in A[j] for j < c ----
        if constexpr (G == 16) dpp_guard(A); // A is compiler-produced (build-time hazard check)
        double dinv[n];
        ldlt_factor<n, G, false>(A, dinv, pivmin);
        if (__builtin_expect(__ballot(!(pivmin > 0.0)) != 0ull, 0)) { // a null direction (cold): pinv-like redo
#pragma unroll
            for (int c = 0; c < n; ++c) A[c] = fma(phi_r, z[c], qd[c]);
            if constexpr (G == 16) dpp_guard(A);
            ldlt_factor<n, G, true>(A, dinv, pivmin);
        }
        // Consume the chunks of step t-1 (requested one iteration ago) and request those of step t-2 BEFORE this
        // iteration's stores are issued.  Loads and stores share vmcnt and complete out of order with respect to each
        // other, so the compiler waits for vmcnt(0) whenever both are pending: here every pending operation is one full
        // iteration old.  The LDS transposition (chunks -> image -> row) has the substitutions and both products to
        // complete in.
        double xf_next = xfn;
        if constexpr (DEFER) {
            if (t >= 1) {
                finish(xf_next, Pfc);
                if (t >= 2) issue(xfn);
            }
            if (recS) {
                RIO::store_all(recS, rmap, etmp);
                recS -= rstep;
            }
        }

        // Sweeps<n>::fused (16-lane groups, n <= 10): every sweep below is ONE asm statement (mk_sweeps.h) -- written as one
        // statement per broadcast-FMA, hipcc pads every sweep boundary with an s_nop (31 issue slots per step at n = 10)
        constexpr bool FUSED = (G == 16) && Sweeps<n>::fused;
        // ---- lane i solves A z = W_i  (row i of J = Pf Phi^T A^{-1}, :458-460) ----
        if constexpr (FUSED) {
            Sweeps<n>::forward(z, A);
        } else {
            sfor<0, n>(MK_LAMBDA(kc) { // forward: L y = b;  z[c] -= L(c,k) y_k, L(c,k) lives in lane c
                constexpr int k = decltype(kc)::value;
                Gp::template axpy_col<k + 1, n, true, n>(z, A[k], z[k]);
            });
        }
#pragma unroll
        for (int c = 0; c < n; ++c) z[c] *= dinv[c]; // D^{-1}
        if constexpr (FUSED) {
            Sweeps<n>::backward(z, A);
        } else {
            sfor_down<0, n>(MK_LAMBDA(kc) { // backward: L^T z = y;  z[c] -= L(k,c) z_k, L(k,c) lives in lane k
                constexpr int k = decltype(kc)::value;
                Gp::template axpy_lane<k, 0, k, true, n>(z, A, z[k]);
            });
        }
        // z = J[r, :]

        // ---- smoothed mean (:461-464): xs[t] = F[t] + J (xs[t+1] - Phi F[t]) ----
        double acc0 = xfc, acc1 = 0.0;
        if constexpr (FUSED) {
            Sweeps<n>::mean(acc0, acc1, delta, z);
        } else {
            sfor<0, n>(MK_LAMBDA(cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (c % 2 == 0) Gp::template fmac<c>(acc0, delta, z[c]);
                else Gp::template fmac<c>(acc1, delta, z[c]);
            });
        }
        xs = acc0 + acc1;

        // ---- smoothed covariance (:465-474): Ps[t] = Pf[t] + J (Ps[t+1] - Pp[t+1]) J^T ----
        double V[n]; // V = J D (row r):  V[c] += J[r][k] * D[k][c], D[k][:] broadcast from lane k
        if constexpr (G == 16) dpp_guard(D); // D is compiler-produced (build-time hazard check)
        if constexpr (FUSED) {
            Sweeps<n>::jd(V, D, z);
        } else {
#pragma unroll
            for (int c = 0; c < n; ++c) V[c] = 0.0;
            sfor<0, n>(MK_LAMBDA(kc) {
                constexpr int k = decltype(kc)::value;
                Gp::template axpy_lane<k, 0, n, false, n>(V, D, z[k]);
            });
        }
        if constexpr (!DEFER) {
#pragma unroll
            for (int c = 0; c < n; ++c) Psn[c] = Pfc[c];
        }
        // Ps[r][c] = Pf[r][c] + sum_k V[r][k] J[c][k], J[c][k] broadcast from lane c
        if constexpr (FUSED) {
            Sweeps<n>::vjt(Psn, z, V);
        } else {
            sfor<0, n>(MK_LAMBDA(kc) {
                constexpr int k = decltype(kc)::value;
                Gp::template axpy_col<0, n, false, n>(Psn, z[k], V[k]);
            });
        }

        if constexpr (!DEFER) { // consume the prefetched record BEFORE this iteration's stores are issued (vmcnt)
            if (t >= 1) {
                finish(xf_next, Pfc);
                if (t >= 2) issue(xfn);
            }
        }
        store(xs, Psn);
        xfc = xf_next;
    }
    if constexpr (DEFER)
        if (recS) RIO::emit(imgS, recS, rmap); // the record of step 0 (T = 1: of the only step)
    if (a.status && live && lane == 0 && pivot_flags(pivmin)) atomicOr(a.status + inst, pivot_flags(pivmin));
}


// ---------------------------------------------------------------------------------------------------------------
// smoother_blk_kernel: the record smoother with its two n^3 products on the matrix pipe (n <= 15, packed records).
// Measured on MI355X (scripts/ubench/power_probe.hip, mfma44.hip): at full-chip occupancy a v_fmac_f64_dpp costs ~3.3 ns
// of SIMD time (2.3 ns without DPP) whatever the number of resident wavefronts, and f64 MFMA shares that pipe (it does
// NOT overlap another wavefront's f64 vector work) -- but v_mfma_f64_4x4x4_4b_f64 retires 4 x 64 multiply-adds in 7.9 ns,
// 1.7x the DPP rate, and its four independent 4x4x4 products are exactly "one block of each of the wavefront's four
// models".  So the factorisation and the substitutions stay in the row-per-lane DPP form (their operands are
// broadcasts of single elements), and
//     D~ = Ps~[t+1] - (pp o Pf~ + Q~),   V^T = D J^T,   Ps~[t] = Pf~ + J V^T                      (kalmanfilter.py:461-474)
// run on 4x4 blocks: lane l = x0 + 4 b + 16 x2 holds element (x2, x0) of a block of model b; such a C/D block is the B
// operand as it stands and acts as its TRANSPOSE when passed as the A operand, so J's blocks (read once from LDS in
// transposed form) serve both products, V^T never leaves the registers and D~ is formed elementwise in the layout the
// previous step's result already has (its transposed blocks are its own mirror blocks: D is symmetric).  The smoothed
// mean rides as column n of the covariance: Pf~ = [Pf | F], D~'s column n = S[t+1] - Phi F[t] passes through the
// accumulator of product 1, column n of the result is S[t].  Padding (rows >= n, columns > n) is zero in the LDS images
// and stays zero (J's padding rows read a zero row).
// LDS per wavefront: a load image and a store image in tile layout (row r of model g at g IM + r TS: 16-byte row reads
// for the DPP part, conflict-free 8-byte block accesses for the MFMA part; mean of row r in column n) and J's rows.
template <int N, int K, bool SYM>
struct BlkLayout {
    static constexpr int n = N + K;
    using RIO = RecordIO<n, 16, SYM>;             // chunk geometry of the HBM records (RSO, RC, CW, PER) and locate()
    static constexpr int NBR = (n + 3) / 4;       // row (and k) blocks
    static constexpr int NBC = (n + 4) / 4;       // column blocks, the mean column included
    static constexpr int TR = 4 * NBR;            // tile rows kept in the image (rows >= n stay zero)
    static constexpr int TS = 20;                 // row stride, doubles (160 B: 16-byte aligned rows; block reads of the two
                                                  // x2 values of a half-wavefront land 40 banks apart)
    static constexpr int IM = ((TR * TS - 8 + 31) / 32) * 32 + 8; // doubles per model, = 8 mod 32: the four models' blocks
                                                                  // start 16 banks apart
    static constexpr int DUMP = 4 * IM, ZERO = 4 * IM + 2; // where loaded pads go / where stored pads come from
    static constexpr int IMG = 4 * IM + 4;        // doubles per image
    static constexpr int JR = 4 * NBR;            // J row stride: columns padded to the block boundary (zeros)
    static constexpr int JM = (n + 1) * JR;       // rows 0..n-1 and one zero row (every padding row reads it)
    static constexpr int WAVE = 2 * IMG + 4 * JM; // doubles per wavefront
    static_assert(n <= 15 && 4 * NBC <= TS, "the mean rides as column n of the tile");
    static_assert(IM % 2 == 0 && IMG % 2 == 0 && JM % 2 == 0 && JR % 2 == 0, "16-byte alignment of the row accesses");

    struct Map {           // loop-invariant per-lane addressing of its PER 16-byte chunks of the wavefront's 4 records
        long off[RIO::PER]; // element offset of the chunk inside the record ARRAY at t = 0
        int l0[RIO::PER], l1[RIO::PER];                     // load side: image positions of the chunk's two doubles
        int s0[RIO::PER], s1[RIO::PER];                     // store side (pads and the smoothed set's sigma/detf: zeros)
        int m0[SYM ? RIO::PER : 1], m1[SYM ? RIO::PER : 1]; // SYM loads: mirror positions of off-diagonal elements
    };
    // index p of a record -> position inside ONE model's tile image; kind: 0 payload, 1 sigma/detf, 2 zero pad
    static __device__ __forceinline__ void locate(int p, int &pos, int &mirror, int &kind)
    {
        if constexpr (SYM) {
            int rp, rm;
            RIO::locate(p, rp, rm, kind); // positions in the full-square record layout
            pos = rp < n ? rp * TS + n : ((rp - n) / n) * TS + (rp - n) % n;
            mirror = rm < n ? rm * TS + n : ((rm - n) / n) * TS + (rm - n) % n;
        } else {
            kind = p < RIO::NV ? 0 : (p < RIO::NV + 2 ? 1 : 2);
            pos = p < n ? p * TS + n : ((p - n) / n) * TS + (p - n) % n;
            mirror = pos;
        }
    }
    static __device__ __forceinline__ Map make_map(int lane64, long inst0, long B, long bs)
    {
        Map mp;
#pragma unroll
        for (int m = 0; m < RIO::PER; ++m) {
            int q = lane64 + 64 * m;
            if (q > RIO::CW - 1) q = RIO::CW - 1; // surplus lanes duplicate the last chunk (same bytes, same address)
            const int g = q / RIO::RC, w = q - g * RIO::RC;
            long ig = inst0 + g;
            if (ig > B - 1) ig = B - 1; // surplus groups replicate the last model
            mp.off[m] = ig * bs * RIO::RSO + 2 * w;
            int pos, mir, kind;
            locate(2 * w, pos, mir, kind);
            mp.l0[m] = kind == 0 ? g * IM + pos : DUMP;
            mp.s0[m] = kind == 0 ? g * IM + pos : ZERO;
            if constexpr (SYM) mp.m0[m] = kind == 0 ? g * IM + mir : DUMP;
            locate(2 * w + 1, pos, mir, kind);
            mp.l1[m] = kind == 0 ? g * IM + pos : DUMP + 1;
            mp.s1[m] = kind == 0 ? g * IM + pos : ZERO + 1;
            if constexpr (SYM) mp.m1[m] = kind == 0 ? g * IM + mir : DUMP + 1;
        }
        return mp;
    }
};

template <int N, int K, int EPI, bool SYM>
__global__ void __launch_bounds__(128) smoother_blk_kernel(SmootherArgs a)
{
    constexpr bool PROJ = (EPI == 1), VAR = (EPI == 2);
    constexpr int n = N + K, G = 16;
    using Gp = Group<G>;
    using BL = BlkLayout<N, K, SYM>;
    using RIO = typename BL::RIO;
    constexpr int PER = RIO::PER, NBR = BL::NBR, NBC = BL::NBC, TS = BL::TS, IM = BL::IM, JR = BL::JR, JM = BL::JM;
    __shared__ __attribute__((aligned(16))) double lds_blk[2 * BL::WAVE];

    const int wave = threadIdx.x >> 6;
    const int lane64 = threadIdx.x & 63;
    const long inst0 = (long)blockIdx.x * 8 + wave * 4; // first of the wavefront's four models
    const long T = a.T;
    double *imgL = lds_blk + wave * BL::WAVE, *imgS = imgL + BL::IMG, *jar = imgS + BL::IMG;
    const long rstep = a.ts * RIO::RSO;

    // ---- row mapping (DPP part, epilogues): lane r of group gw owns row r of model inst0 + gw
    const int lane = lane64 & 15, gw = lane64 >> 4;
    long inst = inst0 + gw;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1; // surplus groups replicate the last model (identical stores)
    const int r = lane < n ? lane : n - 1; // lanes >= n replicate lane n-1
    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    double phic[n], qd[n]; // diag(Phi) replicated; row r of Q = diag(q)
    sfor<0, n>(MK_LAMBDA(c) {
        phic[decltype(c)::value] = Gp::template bcast<decltype(c)::value>(phi_r);
        qd[decltype(c)::value] = (decltype(c)::value == r) ? q_r : 0.0;
    });

    // ---- block mapping (MFMA part): lane = x0 + 4 b + 16 x2 holds element (x2, x0) of a 4x4 block of model inst0 + b
    const int x0 = lane64 & 3, bm = (lane64 >> 2) & 3, x2 = lane64 >> 4;
    const int ibase = bm * IM + x2 * TS + x0; // block (ib, jb) of the image at ibase + 4 ib TS + 4 jb
    int jbase[NBC];                           // block (kb, jb) of J^T at jbase[jb] + 4 kb: J[4 jb + x0][4 kb + x2]
#pragma unroll
    for (int jb = 0; jb < NBC; ++jb) jbase[jb] = bm * JM + (4 * jb + x0 < n ? 4 * jb + x0 : n) * JR + x2;
    double npp[NBR][NBC], qb[NBR], msk[NBR]; // -(pp o .) factors, diagonal of Q, "column n" mask, in block layout
    {
        long ib_ = inst0 + bm;
        if (ib_ > a.B - 1) ib_ = a.B - 1;
        const double *ph = a.phi + ib_ * n, *qq = a.q + ib_ * n;
#pragma unroll
        for (int ib = 0; ib < NBR; ++ib) {
            const int row = 4 * ib + x2;
            const bool rv = row < n;
            const double ph_row = ph[rv ? row : 0];
#pragma unroll
            for (int jb = 0; jb < NBC; ++jb) {
                const int col = 4 * jb + x0;
                const double ph_col = col < n ? ph[col] : (col == n ? 1.0 : 0.0); // column n: Xp[t+1] = Phi F[t]
                npp[ib][jb] = rv ? -(ph_row * ph_col) : 0.0;
            }
            qb[ib] = (rv && x2 == x0) ? qq[row] : 0.0;
            msk[ib] = (rv && x0 == n % 4) ? 1.0 : 0.0;
        }
    }

    const typename BL::Map rmap = BL::make_map(lane64, inst0, a.B, a.bs);
    typename RIO::chunk_t prer[PER]; // chunks of the next filtered record
    for (int i = lane64; i < BL::WAVE; i += 64) imgL[i] = 0.0; // tile padding, J padding, the zero slots

    const double *recF = a.F + (T - 1) * rstep;            // record array positioned at the step being fetched
    double *recS = a.S ? a.S + (T - 1) * rstep : nullptr; // smoothed records are optional when projecting
    const long rec_id = inst % a.R;
    const int jr = lane < N ? lane : N - 1;
    double gam[K], pscale = 1.0, poffset = 0.0;
    double *pM = nullptr, *pV = nullptr;
    if constexpr (PROJ) {
#pragma unroll
        for (int k = 0; k < K; ++k) gam[k] = a.loadings[(rec_id * N + jr) * K + k];
        if (a.scale) pscale = a.scale[rec_id * N + jr];
        if (a.offset) poffset = a.offset[rec_id * N + jr];
        const long pidx = (inst * a.bs + (T - 1) * a.ts) * N + jr;
        pM = a.sim_means ? a.sim_means + pidx : nullptr;
        pV = a.sim_vars ? a.sim_vars + pidx : nullptr;
    }
    double *sM = nullptr, *sV = nullptr; // VAR: state means / variances [., n]
    if constexpr (VAR) {
        const long sidx = (inst * a.bs + (T - 1) * a.ts) * n + r;
        sM = a.state_means + sidx;
        sV = a.state_vars + sidx;
    }

    auto issue = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < PER; ++m) prer[m] = *reinterpret_cast<const typename RIO::chunk_t *>(recF + rmap.off[m]);
        recF -= rstep;
    };
    auto to_image = [&]() __attribute__((always_inline)) { // chunks -> load image (tile layout)
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            imgL[rmap.l0[m]] = prer[m].x;
            imgL[rmap.l1[m]] = prer[m].y;
            if constexpr (SYM) { // mirror of an off-diagonal element; means and diagonals rewrite themselves
                imgL[rmap.m0[m]] = prer[m].x;
                imgL[rmap.m1[m]] = prer[m].y;
            }
        }
        wave_lds_sync();
    };
    // `img` holds the smoothed moments of this step: emit the record, run the epilogues
    auto store = [&](const double *img) __attribute__((always_inline)) {
        wave_lds_sync();
        if (recS) {
            typename RIO::chunk_t tmp[PER];
#pragma unroll
            for (int m = 0; m < PER; ++m) tmp[m] = typename RIO::chunk_t{img[rmap.s0[m]], img[rmap.s1[m]]};
#pragma unroll
            for (int m = 0; m < PER; ++m) *reinterpret_cast<typename RIO::chunk_t *>(recS + rmap.off[m]) = tmp[m];
            recS -= rstep;
        }
        if constexpr (PROJ || VAR) {
            double row[n];
            const double xv = img[gw * IM + r * TS + n];
            load_row<n>(img + gw * IM + r * TS, row);
            if constexpr (PROJ) {
                double mean, var;
                project<N, K, G>(xv, row, gam, pscale, poffset, lane, mean, var);
                if (lane < N && live) { // factor rows hold other values: mask (one exec region per step)
                    if (pM) *pM = mean;
                    if (pV) *pV = var;
                }
                if (pM) pM -= a.ts * N;
                if (pV) pV -= a.ts * N;
            }
            if constexpr (VAR) {
                double diag = 0.0;
                sfor<0, n>(MK_LAMBDA(c) { diag = (decltype(c)::value == lane) ? row[decltype(c)::value] : diag; });
                if (lane < n && live) {
                    *sM = xv;
                    *sV = diag;
                }
                sM -= a.ts * n;
                sV -= a.ts * n;
            }
        }
    };

    // last step: smoothed = filtered (:450-451): the load image as it stands
    double Ps[NBR][NBC];
    issue();
    to_image();
#pragma unroll
    for (int ib = 0; ib < NBR; ++ib)
#pragma unroll
        for (int jb = 0; jb < NBC; ++jb) Ps[ib][jb] = imgL[ibase + 4 * ib * TS + 4 * jb];
    store(imgL);
    double pivmin = 1.0;

    // Software pipeline: at the top of iteration t the load image holds record t, row r of Pf[t] is in registers and
    // the chunks of record t-1 are in flight; they are consumed (image, row) at the END of the iteration, after the
    // image's last reader and BEFORE this iteration's stores are issued (loads and stores share vmcnt).
    double Pfc[n];
    if (T >= 2) {
        issue();
        to_image();
        load_row<n>(imgL + gw * IM + r * TS, Pfc);
        if (T >= 3) issue();
    }
    for (long t = T - 2; t >= 0; --t) {
        // ======== row mapping: J[t] = Pf Phi^T Pp[t+1]^-1 (:453-460) ========
        {
            double A[n], zz[n];
#pragma unroll
            for (int c = 0; c < n; ++c) {
                zz[c] = Pfc[c] * phic[c]; // W = Pf Phi, the right-hand side of the solve
                A[c] = fma(phi_r, zz[c], qd[c]); // Pp[t+1] = Phi Pf Phi + Q (row r)
            }
            dpp_guard(A); // A is compiler-produced (build-time hazard check)
            double dinv[n];
            ldlt_factor<n, G, false>(A, dinv, pivmin);
            if (__builtin_expect(__ballot(!(pivmin > 0.0)) != 0ull, 0)) { // a null direction (cold): pinv-like redo
#pragma unroll
                for (int c = 0; c < n; ++c) A[c] = fma(phi_r, Pfc[c] * phic[c], qd[c]);
                dpp_guard(A);
                ldlt_factor<n, G, true>(A, dinv, pivmin);
            }
            sfor<0, n>(MK_LAMBDA(kc) { // forward: L y = b;  z[c] -= L(c,k) y_k, L(c,k) lives in lane c
                constexpr int k = decltype(kc)::value;
                Gp::template axpy_col<k + 1, n, true, n>(zz, A[k], zz[k]);
            });
#pragma unroll
            for (int c = 0; c < n; ++c) zz[c] *= dinv[c]; // D^{-1}
            sfor_down<0, n>(MK_LAMBDA(kc) { // backward: L^T z = y;  z[c] -= L(k,c) z_k, L(k,c) lives in lane k
                constexpr int k = decltype(kc)::value;
                Gp::template axpy_lane<k, 0, k, true, n>(zz, A, zz[k]);
            });
            store_row<n>(jar + gw * JM + r * JR, zz); // row r of J[t]
        }
        wave_lds_sync();
        // ======== block mapping: the recursion (:461-474) on 4x4 blocks ========
        {
            double JT[NBR][NBC], D[NBR][NBC], acc[NBR][NBC], out[NBR][NBC];
#pragma unroll
            for (int ib = 0; ib < NBR; ++ib)
#pragma unroll
                for (int jb = 0; jb < NBC; ++jb) {
                    out[ib][jb] = imgL[ibase + 4 * ib * TS + 4 * jb]; // Pf~ block: the accumulator of product 2
                    JT[ib][jb] = jar[jbase[jb] + 4 * ib];
                }
            // D~ = Ps~[t+1] - Pp~[t+1]; column n: S[t+1] - Xp[t+1]
#pragma unroll
            for (int ib = 0; ib < NBR; ++ib)
#pragma unroll
                for (int jb = 0; jb < NBC; ++jb) {
                    const double d = fma(npp[ib][jb], out[ib][jb], Ps[ib][jb]);
                    D[ib][jb] = jb == ib ? d - qb[ib] : d;
                    acc[ib][jb] = jb == n / 4 ? D[ib][jb] * msk[ib] : 0.0; // D~'s column n passes through product 1
                }
            // V^T (ib, jb) (+)= sum_kb D(ib, kb) J^T(kb, jb): the A operand D(ib, kb) is the block D[kb][ib] (transposed by
            // the operand layout, D symmetric)
#pragma unroll
            for (int kb = 0; kb < NBR; ++kb)
#pragma unroll
                for (int ib = 0; ib < NBR; ++ib)
#pragma unroll
                    for (int jb = 0; jb < NBC; ++jb)
                        acc[ib][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(D[kb][ib], JT[kb][jb], acc[ib][jb], 0, 0, 0);
            // Ps~(ib, jb) = Pf~(ib, jb) + sum_kb J(ib, kb) V^T(kb, jb): the A operand J(ib, kb) is the block JT[kb][ib]
#pragma unroll
            for (int kb = 0; kb < NBR; ++kb)
#pragma unroll
                for (int ib = 0; ib < NBR; ++ib)
#pragma unroll
                    for (int jb = 0; jb < NBC; ++jb)
                        out[ib][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(JT[kb][ib], acc[kb][jb], out[ib][jb], 0, 0, 0);
#pragma unroll
            for (int ib = 0; ib < NBR; ++ib)
#pragma unroll
                for (int jb = 0; jb < NBC; ++jb) {
                    Ps[ib][jb] = out[ib][jb];
                    imgS[ibase + 4 * ib * TS + 4 * jb] = out[ib][jb];
                }
        }
        // consume the chunks of step t-1, request those of step t-2 -- before this iteration's stores are issued
        if (t >= 1) {
            to_image();
            load_row<n>(imgL + gw * IM + r * TS, Pfc);
            if (t >= 2) issue();
        }
        store(imgS);
    }
    if (a.status && live && lane == 0 && pivot_flags(pivmin)) atomicOr(a.status + inst, pivot_flags(pivmin));
}


// dense arrays (any strides, optional outputs): symmetric column runs, no LDS
template <int n, int G>
__global__ void __launch_bounds__(256) smoother_dense_kernel(SmootherArgs a)
{
    static_assert(n <= G, "state dimension must fit the lane group");
    using Gp = Group<G>;
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    long inst = (long)blockIdx.x * GPB + threadIdx.x / G;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1; // surplus groups replicate the last model (identical stores)
    const int r = lane < n ? lane : n - 1; // lanes >= n replicate lane n-1
    const long T = a.T;

    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    double phic[n], qd[n]; // diag(Phi) replicated; row r of Q = diag(q)
    sfor<0, n>(MK_LAMBDA(c) {
        phic[decltype(c)::value] = Gp::template bcast<decltype(c)::value>(phi_r);
        qd[decltype(c)::value] = (decltype(c)::value == r) ? q_r : 0.0;
    });

    // (b, t) at block index b*bs + t*ts; start at the last step
    const long blkT = inst * a.bs + (T - 1) * a.ts;
    MomentPtr iF = moment_ptr<n>(const_cast<double *>(a.F), const_cast<double *>(a.Pf), blkT, a.ts, a.rs, r);
    MomentPtr oS = moment_ptr<n>(a.S, a.Ps, blkT, a.ts, a.rs, r);
    // smoothed records: the pad doubles are written as zeros so that every cache line is written whole
    constexpr int NV = record_payload(n), RS = record_stride_c(n), PADN = RS - NV;
    double *padS = (a.rs > 0 && a.S) ? a.S + blkT * RS + NV + (lane < PADN ? lane : PADN - 1) : nullptr;
    auto store = [&](double xv, const double(&row)[n]) __attribute__((always_inline)) {
        if (oS.vec) *oS.vec = xv;
        if (oS.mat) store_cols<n>(oS.mat, row);
        oS.advance(-1);
        if (padS) {
            *padS = 0.0;
            padS -= a.ts * RS;
        }
    };

    // last step: smoothed = filtered (:450-451)
    double xs = *iF.vec, Psn[n];
    load_cols<n>(iF.mat, Psn);
    store(xs, Psn);
    double pivmin = 1.0;

    // Software pipeline: the rows of Pf[t] / F[t] are loaded one iteration ahead (column runs, no LDS).
    double Pfc[n], Pfn[n], xfc = 0.0, xfn = 0.0;
    if (T >= 2) {
        iF.advance_nn(-1);
        xfc = *iF.vec;
        load_cols<n>(iF.mat, Pfc);
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
        if constexpr (G == 16) dpp_guard(A); // A is compiler-produced (build-time hazard check)
        double dinv[n];
        ldlt_factor<n, G, false>(A, dinv, pivmin);
        if (__builtin_expect(__ballot(!(pivmin > 0.0)) != 0ull, 0)) { // a null direction (cold): pinv-like redo
#pragma unroll
            for (int c = 0; c < n; ++c) A[c] = fma(phi_r, z[c], qd[c]);
            if constexpr (G == 16) dpp_guard(A);
            ldlt_factor<n, G, true>(A, dinv, pivmin);
        }

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
        // Ps[r][c] = Pf[r][c] + sum_k V[r][k] J[c][k], J[c][k] broadcast from lane c
#pragma unroll
        for (int c = 0; c < n; ++c) Psn[c] = Pfc[c];
        // mid-iteration prefetch of step t-1 (consumed at the next iteration).  vmcnt retires in order: the
        // loads must not sit right behind this iteration's stores, or consuming them means waiting for
        // those stores' acknowledgements (~2 us under HBM write load)
        if (t >= 1) {
            iF.advance_nn(-1);
            xfn = *iF.vec;
            load_cols<n>(iF.mat, Pfn);
        }
        sfor<0, n>(MK_LAMBDA(kc) {
            constexpr int k = decltype(kc)::value;
            Gp::template axpy_col<0, n, false, n>(Psn, z[k], V[k]);
        });

        store(xs, Psn);
#pragma unroll
        for (int c = 0; c < n; ++c) Pfc[c] = Pfn[c];
        xfc = xfn;
    }
    if (a.status && live && lane == 0 && pivot_flags(pivmin)) atomicOr(a.status + inst, pivot_flags(pivmin));
}

// =====================================================================================
// Objective of ONE record for many parameter sets, walking only the OBSERVED steps
//   (the solver loop of Metran.solve on real data: every finite-difference instance shares the record, and
//   real records are sparse -- examples/data has observations on 343 of its 6255 daily steps).  Between two
//   observed steps t- < t the prediction is applied in closed form (Phi diagonal):
//       x <- phi^g x,   P[r][c] <- (phi_r phi_c)^g P[r][c] + [r == c] q_r (1 - phi_r^2g)/(1 - phi_r^2),   g = t - t-
//   (g = 1 is the ordinary predict), then the scalar updates of step t run as in filter_kernel.  No state
//   output exists in this mode, so nothing is lost by not visiting the empty steps; the compressed warm-up
//   index of get_mle (:563-564) is simply the position in the list of observed steps.
//   observed_steps_kernel builds that list (ascending) on the device: tlist[0] = count, tlist[1..] = t.
// =====================================================================================
__global__ void __launch_bounds__(256) observed_steps_kernel(long T, int N, long ostep, const double *obs, int *tlist)
{
    __shared__ int s_wcnt[4];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (long t0 = 0; t0 < T; t0 += 256) {
        const long t = t0 + tid;
        bool f = false;
        if (t < T)
            for (int j = 0; j < N; ++j) f = f || isfinite(obs[t * ostep + j]);
        const unsigned long long b = __ballot(f);
        if (lane == 0) s_wcnt[w] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int i = 0; i < w; ++i) off += s_wcnt[i];
        if (f) tlist[1 + off + __popcll(b & ((1ull << lane) - 1ull))] = (int)t;
        __syncthreads();
        if (tid == 0) s_base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    if (tid == 0) tlist[0] = s_base;
}

// REC (round 5): the same walk also WRITES the packed records of the observed steps -- predicted moments before the updates,
// filtered moments and the compressed (sigma, detf) entry after them -- and fill_gaps_kernel then writes the records of the
// empty steps in closed form, fully parallel.  This is the single-record engine route (mk_filter with one record, a
// handful of instances, all four state arrays: what Metran.solve() drives through seqkalmanfilter ~80 times): on
// examples/data 343 sequential steps instead of 6255.
template <int N, int K, int G, bool REC = false>
__global__ void __launch_bounds__(256) loglik_sparse_kernel(SparseArgs a)
{
    constexpr int n = N + K;
    static_assert(G == 16 && n <= 16, "sparse objective kernel: one model per 16-lane group");
    [[maybe_unused]] constexpr int NV = record_payload(n);
    using Gp = Group<G>;
    constexpr int GPB = 256 / G;
    constexpr bool HOIST = (N * K <= 32);
    const int lane = threadIdx.x % G;
    long inst = (long)blockIdx.x * GPB + threadIdx.x / G;
    if (inst > a.B - 1) inst = a.B - 1;
    const int r = lane < n ? lane : n - 1;
    const bool lead = lane == 0;

    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    const double lphi_r = log(phi_r); // -inf when phi underflowed to 0 at the lower bound: exp(-inf) = 0 below
    double pp[n], qd[n];
    sfor<0, n>(MK_LAMBDA(c) {
        constexpr int cc = decltype(c)::value;
        pp[cc] = phi_r * Gp::template bcast<cc>(phi_r);
        qd[cc] = (cc == r) ? q_r : 0.0;
    });
    const double inv_em = 1.0 / expm1(2.0 * lphi_r); // 1/(phi_r^2 - 1)
    const int jr = lane < N ? lane : N - 1;
    double gam[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gam[k] = a.loadings[jr * K + k];
    const double rvar = a.obsvar ? a.obsvar[jr] : 0.0;
    double Gh[HOIST ? N : 1][K];
    if constexpr (HOIST) {
        sfor<0, N>(MK_LAMBDA(j) {
            sfor<0, K>(MK_LAMBDA(k) {
                Gh[decltype(j)::value][decltype(k)::value] = Gp::template bcast<decltype(j)::value>(gam[decltype(k)::value]);
            });
        });
    }
    double x = a.x0 ? a.x0[inst * n + r] : 0.0;
    double P[n];
#pragma unroll
    for (int c = 0; c < n; ++c) P[c] = a.P0 ? a.P0[(inst * n + r) * n + c] : (c == r ? 1.0 : 0.0);

    const int cnt = a.tlist[0];
    double sum_sig = 0.0, run_mant = 1.0, fmin_seen = 1.0;
    long run_exp = 0, nobs = 0, tprev = -1;
    // The observed steps come through LDS, a tile of 256 at a time: every thread of the workgroup fetches ONE step (its time index
    // from the list, then its N observations -- every instance shares the record) and all sixteen groups walk the tile.  Fetched
    // per step inside the walk (rounds 3-5: the next step's index and observation one iteration ahead), the two dependent loads
    // were the walk's critical path -- 1.1 us per observed step for ~0.4 us of arithmetic -- and, in the record-writing form,
    // every consumed load waited for the older record stores (vmcnt retires in order).
    constexpr int TSP = 256;
    __shared__ int s_t[TSP];
    __shared__ double s_y[TSP * N];

    for (int i0 = 0; i0 < cnt; i0 += TSP) {
    __syncthreads(); // the previous tile has been walked by every group
    {
        const int idx = i0 + (int)threadIdx.x;
        if (idx < cnt) {
            const int tt = a.tlist[1 + idx];
            s_t[threadIdx.x] = tt;
            const double *src = a.obs + (long)tt * a.ostep;
#pragma unroll
            for (int j = 0; j < N; ++j) s_y[threadIdx.x * N + j] = src[j];
        }
    }
    __syncthreads();
    const int iend = cnt - i0 < TSP ? cnt - i0 : TSP;
    for (int s_ = 0; s_ < iend; ++s_) {
        const int i = i0 + s_;
        const long t = s_t[s_];
        const double y = s_y[s_ * N + jr];
        const long gap = t - tprev;
        tprev = t;
        if (gap == 1) { // ordinary predict (:318-331)
            x = phi_r * x;
#pragma unroll
            for (int c = 0; c < n; ++c) P[c] = fma(P[c], pp[c], qd[c]);
        } else { // gap - 1 empty steps and the predict of step t in one go
            // (phi_r phi_c)^gap as the product of the two lanes' phi^gap: ONE exponential per lane and n broadcast products
            // instead of n exponentials (an f64 exp is ~45 instructions; six of them were half of this walk's time per step)
            const double g = (double)gap;
            const double er = exp(g * lphi_r);
            x *= er;
            const double qg = q_r * expm1(2.0 * g * lphi_r) * inv_em;
            sfor<0, n>(MK_LAMBDA(c) {
                constexpr int cc = decltype(c)::value;
                P[cc] = fma(P[cc], er * Gp::template bcast<cc>(er), cc == r ? qg : 0.0);
            });
        }
        const unsigned long long ball = __ballot(lane < N && isfinite(y));
        const auto vm = Gp::group_bits(ball);
        if constexpr (REC) { // predicted moments of step t (:332-333): mean element r, covariance as symmetric column runs
            double *rp = a.Xp + (inst * a.bs + t * a.ts) * a.rs;
            rp[r] = x;
            store_cols<n>(rp + n + r, P);
        }
        double sigma = 0.0, fmant = 1.0;
        int fexp = 0;
        sfor<0, N>(MK_LAMBDA(jc) {
            constexpr int j = decltype(jc)::value;
            if ((vm >> j) & 1) {
                double vl = y - x;
                sfor<0, K>(MK_LAMBDA(k) { Gp::template fmac<N + decltype(k)::value, true>(vl, x, gam[decltype(k)::value]); });
                const double v = Gp::template bcast<j>(vl);
                double dr = P[j];
                sfor<0, K>(MK_LAMBDA(k) {
                    constexpr int kk = decltype(k)::value;
                    double g;
                    if constexpr (HOIST) g = Gh[j][kk];
                    else g = Gp::template bcast<j>(gam[kk]);
                    dr = fma(P[N + kk], g, dr);
                });
                double fl = rvar + dr;
                dpp_pin(dr);
                sfor<0, K>(MK_LAMBDA(k) { Gp::template fmac<N + decltype(k)::value, false>(fl, dr, gam[decltype(k)::value]); });
                const double f = Gp::template bcast<j>(fl);
                const double rf = rcp_nr(f);
                const double kr = dr * rf;
                Gp::template axpy_col<0, n, true, n>(P, dr, kr);
                x = fma(kr, v, x);
                sigma = fma(v * v, rf, sigma);
                fmant *= f;
                fmin_seen = min_f64(fmin_seen, f);
            }
            if constexpr ((j & 3) == 3 || j == N - 1) {
                fexp += __builtin_amdgcn_frexp_exp(fmant);
                fmant = __builtin_amdgcn_frexp_mant(fmant);
            }
        });
        if constexpr (REC) { // filtered moments of step t (:384-390); the compressed (sigma, detf) entry i in record i's pad (:380-382)
            double *rf_ = a.F + (inst * a.bs + t * a.ts) * a.rs;
            rf_[r] = x;
            store_cols<n>(rf_ + n + r, P);
            if (lead) {
                double *pad = a.F + (inst * a.bs + (long)i * a.ts) * a.rs + NV;
                pad[0] = sigma;
                pad[1] = fma((double)fexp, kLn2, log(fmant));
            }
        }
        if (i >= a.warmup) { // compressed index of the observed step (:563-564)
            sum_sig += sigma;
            run_mant *= fmant;
            run_exp += fexp + __builtin_amdgcn_frexp_exp(run_mant);
            run_mant = __builtin_amdgcn_frexp_mant(run_mant);
        }
        if (t >= a.warmup) nobs += __popcll((unsigned long long)vm); // TIME index (:565)
    }
    }
    if (lead) {
        const double sum_det = fma((double)run_exp, kLn2, log(run_mant));
        if (a.mle) a.mle[inst] = ((double)nobs * kLog2Pi + sum_det) + sum_sig;
        if (a.status) a.status[inst] = (fmin_seen > 0.0) ? 0u : MK_FLAG_NONPOSITIVE_F;
        if (REC && a.sigmacount) a.sigmacount[inst] = cnt;
    }
}

// The records of the steps WITHOUT an observation, after loglik_sparse_kernel<.., REC> has written those of the observed ones:
// over a run of empty steps the recursion is closed-form under a diagonal transition (kalmanfilter.py:318-331 applied g times,
// :384-390 with no update): with (x_a, P_a) the filtered moments of the last observed step t_a < t (or the initial moments,
// t_a = -1) and g = t - t_a,
//     x_t = phi^g o x_a,   P_t[r][c] = (phi_r phi_c)^g P_a[r][c] + [r = c] q_r (1 - phi_r^(2g)) / (1 - phi_r^2)
// are BOTH the predicted and the filtered moments of step t.  One workgroup per (instance, step), one thread per record
// element; the previous observed step by bisection of the list.  Also zeroes the (sigma, detf) pads beyond the compressed
// entries (np.zeros, :307-308).
__global__ void __launch_bounds__(256) fill_gaps_kernel(SparseArgs a, int n)
{
    const long t = blockIdx.x, inst = blockIdx.y;
    const int cnt = a.tlist[0], NV = n + n * n;
    int lo = 0, hi = cnt; // number of observed steps < t
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a.tlist[1 + mid] < t) lo = mid + 1;
        else hi = mid;
    }
    const bool observed = lo < cnt && a.tlist[1 + lo] == t;
    double *rF = a.F + (inst * a.bs + t * a.ts) * a.rs, *rP = a.Xp + (inst * a.bs + t * a.ts) * a.rs;
    const long ta = lo > 0 ? a.tlist[lo] : -1; // last observed step before t (entry lo - 1 of the list)
    const double g = (double)(t - ta);
    const double *src = ta >= 0 ? a.F + (inst * a.bs + ta * a.ts) * a.rs : nullptr;
    for (int e = threadIdx.x; e < (int)a.rs; e += 256) {
        if (e >= NV) { // the record pads: compressed entries live in records 0 .. cnt-1 of the filtered array
            if (t >= cnt) rF[e] = 0.0;
            rP[e] = 0.0;
            continue;
        }
        if (observed) continue;
        double v;
        if (e < n) {
            const double x0 = src ? src[e] : (a.x0 ? a.x0[inst * n + e] : 0.0);
            v = x0 * exp(g * log(a.phi[inst * n + e]));
        } else {
            const int rr = (e - n) / n, cc = (e - n) % n;
            const double p0 = src ? src[e] : (a.P0 ? a.P0[(inst * n + rr) * n + cc] : (rr == cc ? 1.0 : 0.0));
            const double lr = log(a.phi[inst * n + rr]), lc = log(a.phi[inst * n + cc]);
            v = p0 * exp(g * (lr + lc));
            if (rr == cc) v = fma(a.q[inst * n + rr], lr == 0.0 ? g : expm1(2.0 * g * lr) / expm1(2.0 * lr), v); // sum_{i<g} phi^(2i)
        }
        rF[e] = v;
        rP[e] = v;
    }
}

// =====================================================================================
// Reverse-mode (adjoint) gradient of the objective -2 log L with respect to diag(Phi) and diag(Q)
//                                                     ("analytic/adjoint gradient")
//   The reference gives scipy no gradient (metran/solver.py:248-255): every gradient is P+1 filter runs.
//   This kernel walks the filter BACKWARDS once.  Per step it re-reads the filtered moments of step t-1
//   (the packed records written by filter_kernel, OUT = 3), recomputes the prediction and the scalar
//   updates of step t keeping (d, 1/f, v) of every update, and pulls the adjoints (xb, Pb) of the filtered
//   state back through them:
//       update  x' = x + d v/f, P' = P - d d^T/f, l += w (log f + v^2/f), d = P z, f = z^T d + R, v = y - z^T x
//         a = xb.d   b = Pb d   c = d.b
//         vb = (2 w v + a)/f      fb = (w (1 - v^2/f) - a v/f + c/f)/f
//         db = xb v/f - 2 b/f + fb z      xb -= vb z      Pb += (db z^T + z db^T)/2
//       predict x = phi x-, P = (phi phi^T) o P- + diag(q)
//         gq += diag(Pb)   gphi_r += xb_r x-_r + 2 sum_c Pb[r][c] P-[r][c] phi_c
//         xb = phi o xb    Pb = (phi phi^T) o Pb
//   Lane r owns row r of P and Pb and accumulates the gradient of ITS parameters; the three cross-lane
//   sums per update and the two matrix-vector products are fused DPP broadcast-FMAs.  One model per
//   16-lane group (n <= 16); wider models keep the finite-difference path.
//   Checked against central differences of the oracle and a numpy restatement (tests/adjoint_ref.py).
// =====================================================================================
template <int N, int K, int G>
__global__ void __launch_bounds__(256) adjoint_kernel(AdjointArgs a)
{
    constexpr int n = N + K;
    static_assert(G == 16 && n <= 16, "adjoint kernel: one model per 16-lane group");
    using Gp = Group<G>;
    constexpr int GPB = 256 / G;
    // Registers decide this kernel's speed (round 5): with more than 256 (VGPRs + AGPRs) a SIMD holds ONE wavefront of it, and
    // a flight of 8192 models -- 2048 wavefronts on 1024 SIMDs -- runs as two rounds of a latency-bound kernel (round 4: 294
    // registers, 4.85 ms against 1.72 ms for the recording forward pass).  What was hoisted for speed is therefore kept small:
    // the loadings of series j reach the multiply-adds as DPP broadcasts of the lanes' own `gam` (no N x K table), element r
    // of every observation row Z_j is one array zt[N], and the innovation / reciprocal variance of update j wait in lane j
    // alone (two registers instead of 2 N) and are broadcast on the way back.
    constexpr bool FUSED = Sweeps<n>::fused; // whole sums as ONE asm statement (mk_sweeps.h): no padding nops in dependent chains
    const int lane = threadIdx.x % G;
    long inst = (long)blockIdx.x * GPB + threadIdx.x / G;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1;
    const long rec = inst % a.R;
    const int r = lane < n ? lane : n - 1;
    const long T = a.T;

    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    double pp[n]; // row r of Phi (x) Phi
    sfor<0, n>(MK_LAMBDA(c) { pp[decltype(c)::value] = phi_r * Gp::template bcast<decltype(c)::value>(phi_r); });
    const int jr = lane < N ? lane : N - 1;
    double gam[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gam[k] = a.loadings[(rec * N + jr) * K + k];
    const double rvar = a.obsvar ? a.obsvar[rec * N + jr] : 0.0;
    double zt[N]; // element r of Z_j = e_j + sum_k loadings[j,k] e_{N+k}, j = 0 .. N-1
    sfor<0, N>(MK_LAMBDA(jc) {
        constexpr int j = decltype(jc)::value;
        double z = (r == j) ? 1.0 : 0.0;
        sfor<0, K>(MK_LAMBDA(k) {
            constexpr int kk = decltype(k)::value;
            const double g = Gp::template bcast<j>(gam[kk]);
            z = (r == N + kk) ? g : z;
        });
        zt[j] = z;
    });
    double ones[n];
#pragma unroll
    for (int c = 0; c < n; ++c) ones[c] = 1.0;
    const long sctot = a.sigmacount[inst]; // observed steps in total (written by the forward filter)
    long rem = 0;                          // observed steps already walked (from the end)
    const double one = 1.0;

    const long RS = a.rs;
    const double *recbase = a.F + inst * a.bs * RS;
    const long rstep = a.ts * RS;
    const double *obase = a.obs + rec * a.obs_bs * N + jr;
    const long ostep = a.obs_ts * N;

    auto load_prev = [&](long t, double &xv, double(&row)[n]) __attribute__((always_inline)) {
        if (t > 0) { // filtered moments of step t-1: vector at [r], matrix as column runs
            const double *p = recbase + (t - 1) * rstep;
            xv = p[r];
            load_cols<n>(p + n + r, row);
        } else { // run_filter defaults (kalmanfilter.py:747-750) or the caller's initial state
            xv = a.x0 ? a.x0[inst * n + r] : 0.0;
#pragma unroll
            for (int c = 0; c < n; ++c) row[c] = a.P0 ? a.P0[(inst * n + r) * n + c] : (c == r ? 1.0 : 0.0);
        }
    };

    double xb = 0.0, Pb[n], gphi = 0.0, gq = 0.0;
#pragma unroll
    for (int c = 0; c < n; ++c) Pb[c] = 0.0;
    double xnext, Pnext[n], ynext;
    load_prev(T - 1, xnext, Pnext);
    ynext = obase[(T - 1) * ostep];

    for (long t = T - 1; t >= 0; --t) {
        const double xprev = xnext, y = ynext;
        double Pprev[n];
#pragma unroll
        for (int c = 0; c < n; ++c) Pprev[c] = Pnext[c];
        if (t > 0) { // one step ahead; this loop stores nothing, so the loads never queue behind stores
            load_prev(t - 1, xnext, Pnext);
            ynext = obase[(t - 1) * ostep];
        }
        const unsigned long long ball = __ballot(lane < N && isfinite(y));
        const auto vm = Gp::group_bits(ball);

        if (vm != 0) { // uniform within the lane group
            const double w = (sctot - rem - 1 >= a.warmup) ? 1.0 : 0.0; // compressed index of this step (:563-564)
            ++rem;
            // ---- forward: prediction and scalar updates of step t, as filter_kernel ----
            double x = phi_r * xprev, P[n];
            int rv = r; // opaque copies: the n selects stay inside the loop instead of 2n hoisted VGPRs
            double qv = q_r;
            asm volatile("" : "+v"(rv), "+v"(qv));
#pragma unroll
            for (int c = 0; c < n; ++c) P[c] = fma(Pprev[c], pp[c], c == rv ? qv : 0.0);
            double dS[N];
            double vown = 0.0, rfown = 0.0; // lane j: innovation and reciprocal variance of ITS update
            sfor<0, N>(MK_LAMBDA(jc) {
                constexpr int j = decltype(jc)::value;
                dS[j] = 0.0;
                if ((vm >> j) & 1) {
                    double vl = y - x;
                    sfor<0, K>(MK_LAMBDA(k) { Gp::template fmac<N + decltype(k)::value, true>(vl, x, gam[decltype(k)::value]); });
                    const double v = Gp::template bcast<j>(vl);
                    double dr = P[j];
                    sfor<0, K>(MK_LAMBDA(k) { // d_r = P[r][j] + sum_k P[r][N+k] loadings[j,k], the loadings broadcast from lane j
                        constexpr int kk = decltype(k)::value;
                        Gp::template fmac<j, false>(dr, gam[kk], P[N + kk]);
                    });
                    double fl = rvar + dr;
                    dpp_pin(dr);
                    sfor<0, K>(MK_LAMBDA(k) { Gp::template fmac<N + decltype(k)::value, false>(fl, dr, gam[decltype(k)::value]); });
                    const double f = Gp::template bcast<j>(fl);
                    const double rf = rcp_nr(f);
                    const double kr = dr * rf;
                    Gp::template axpy_col<0, n, true, n>(P, dr, kr);
                    x = fma(kr, v, x);
                    dS[j] = dr;
                    vown = (r == j) ? v : vown;
                    rfown = (r == j) ? rf : rfown;
                }
            });
            // ---- reverse: adjoints back through the updates, last observation first ----
            sfor_down<0, N>(MK_LAMBDA(jc) {
                constexpr int j = decltype(jc)::value;
                if ((vm >> j) & 1) {
                    double dr = dS[j];
                    const double rf = Gp::template bcast<j>(rfown), v = Gp::template bcast<j>(vown);
                    double pa = xb * dr;
                    dpp_guard1(pa, dr);
                    double asum, b;
                    if constexpr (FUSED) { // a = sum_r xb_r d_r and b_r = sum_c Pb[r][c] d_c, one asm statement each (two chains inside)
                        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
                        Sweeps<n>::mean(a0, a1, pa, ones);
                        Sweeps<n>::mean(b0, b1, dr, Pb);
                        asum = a0 + a1;
                        b = b0 + b1;
                    } else {
                        double a0 = 0.0, b0 = 0.0, b1 = 0.0;
                        sfor<0, n>(MK_LAMBDA(cc) {
                            constexpr int c = decltype(cc)::value;
                            Gp::template fmac<c, false>(a0, pa, one);
                            if constexpr (c % 2 == 0) Gp::template fmac<c, false>(b0, dr, Pb[c]);
                            else Gp::template fmac<c, false>(b1, dr, Pb[c]);
                        });
                        asum = a0;
                        b = b0 + b1;
                    }
                    double ps = dr * b;
                    dpp_guard1(ps, ps);
                    double csum;
                    if constexpr (FUSED) {
                        double c0 = 0.0, c1 = 0.0;
                        Sweeps<n>::mean(c0, c1, ps, ones);
                        csum = c0 + c1;
                    } else {
                        csum = 0.0;
                        sfor<0, n>(MK_LAMBDA(cc) { Gp::template fmac<decltype(cc)::value, false>(csum, ps, one); });
                    }
                    const double vrf = v * rf;
                    const double vbar = fma(2.0 * w, v, asum) * rf;
                    const double fbar = (fma(-w * v, vrf, w) - asum * vrf + csum * rf) * rf;
                    const double zr = zt[j]; // element r of Z_j
                    double dbar = fma(xb, vrf, fma(-2.0 * rf, b, fbar * zr));
                    xb = fma(-vbar, zr, xb);
                    double hd = 0.5 * dbar;
                    const double hz = 0.5 * zr;
                    Pb[j] += hd;
                    dpp_pin(hd);
                    sfor<0, K>(MK_LAMBDA(k) { // Pb[r][N+k] += db_r loadings[j,k] / 2, the loadings broadcast from lane j
                        constexpr int kk = decltype(k)::value;
                        Gp::template fmac<j, false>(Pb[N + kk], gam[kk], hd);
                    });
                    dpp_guard1(dbar, dbar);
                    Gp::template axpy_col<0, n, false, n>(Pb, dbar, hz); // Pb[r][c] += z_r db_c / 2
                }
            });
        }
        // ---- prediction adjoint ----
        double diag = 0.0, ts0 = 0.0, ts1 = 0.0;
        double phv = phi_r;
        dpp_pin(phv);
        sfor<0, n>(MK_LAMBDA(cc) {
            constexpr int c = decltype(cc)::value;
            diag = (c == r) ? Pb[c] : diag;
            const double pbp = Pb[c] * Pprev[c];
            if constexpr (c % 2 == 0) Gp::template fmac<c, false>(ts0, phv, pbp); // += phi_c Pb[r][c] P-[r][c]
            else Gp::template fmac<c, false>(ts1, phv, pbp);
            Pb[c] *= pp[c];
        });
        gq += diag;
        gphi = fma(xb, xprev, fma(2.0, ts0 + ts1, gphi));
        xb *= phi_r;
    }
    if (live && lane < n) {
        if (a.gphi) a.gphi[inst * n + lane] = gphi;
        if (a.gq) a.gq[inst * n + lane] = gq;
    }
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
        sim_vars[i] = v < 0.0 ? 0.0 : v; // np.maximum(., 0) keeps a NaN (:601-602)
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
    if constexpr (n > 16 && N <= 32) { // wide models: several per wavefront in the split layout (mk_split.hip)
        // ... when the batch is large enough for it to pay: a split wavefront serves two (four) models in little more than the
        // time a lane-per-state wavefront serves one, but with B <= 2 x #SIMDs the lane-per-state kernel still has a SIMD per
        // one or two wavefronts and the split one leaves half of them idle (measured, (32,4), T = 400, records / objective:
        // B = 1024: 4.5 / 4.2 ms against 6.9 / 6.2; B = 2048: 6.8 / 5.7 against 7.7 / 6.3; B = 3072: 10.8 / 9.1 against 10.0 / 8.0).
        // variant bit 0: lane per state always; bit 1: split always; the tape (MK_OUT_TAPE) exists in the split layout only.
        static const long simds = [] {
            int dev = 0, cu = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
            return 4L * cu;
        }();
        const bool split = a.tape || (a.variant & 2) || (!(a.variant & 1) && a.B > 2 * simds);
        if (split) {
            const hipError_t e = launch_filter_split(N, K, a, s);
            if (e != hipErrorNotSupported) return e;
        }
    }
    if constexpr (n > 16 && N > 32) { // the backward tape of the shapes beyond the split layout: the lane-per-state loop filter writes it
        if (a.tape) {
            if (book) hipLaunchKernelGGL((filter_kernel<N, K, G, 4, true, false>), dim3(grid), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((filter_kernel<N, K, G, 4, false, false>), dim3(grid), dim3(256), 0, s, a);
            return hipGetLastError();
        }
    } else if (a.tape) {
        return hipErrorNotSupported;
    }
    if (!any && !book)
        hipLaunchKernelGGL((filter_kernel<N, K, G, 0, false, false>), dim3(grid), dim3(256), 0, s, a);
    else if (!any)
        hipLaunchKernelGGL((filter_kernel<N, K, G, 0, true, false>), dim3(grid), dim3(256), 0, s, a);
    else if (a.rs > 0 && a.Xp && a.sym) // packed-symmetric records
        hipLaunchKernelGGL((filter_kernel<N, K, G, 1, true, true>), dim3(grid), dim3(256), 0, s, a);
    else if (a.rs > 0 && a.Xp) // packed records (validated by the C ABI): whole-cache-line stores
        hipLaunchKernelGGL((filter_kernel<N, K, G, 1, true, false>), dim3(grid), dim3(256), 0, s, a);
    else if (a.rs > 0 && a.sym) // filtered record only
        hipLaunchKernelGGL((filter_kernel<N, K, G, 3, true, true>), dim3(grid), dim3(256), 0, s, a);
    else if (a.rs > 0)
        hipLaunchKernelGGL((filter_kernel<N, K, G, 3, true, false>), dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((filter_kernel<N, K, G, 2, true, false>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// n <= 15, packed records: smoother_record_kernel (all DPP; the default, measured faster at every batch size, DESIGN.md
// section 4) or, with mk_set_kernel_variant(ctx, MK_VARIANT_SMOOTHER16, 1), smoother_blk_kernel (products as 4x4x4 f64
// MFMA blocks); both tested against the oracle (tests/test_smoother_variants.py)
template <int N, int K>
static hipError_t launch_smoother_nk(const SmootherArgs &a, hipStream_t s)
{
    constexpr int n = N + K;
    constexpr int G = n <= 16 ? 16 : 64;
    if constexpr (G == 64) { // one model per wavefront: mk_wide.hip
        return launch_smoother_wide(N, K, a, s);
    } else {
        constexpr int GPB = 256 / G;
        const unsigned grid = (unsigned)((a.B + GPB - 1) / GPB);
        const int epi = (a.sim_means || a.sim_vars) ? 1 : (a.state_means ? 2 : 0);
        if constexpr (n <= 15) {
            if (a.rs > 0 && (a.variant & 1)) {
                const unsigned bgrid = (unsigned)((a.B + 7) / 8);
#define MK_LAUNCH_BLK(E, S) hipLaunchKernelGGL((smoother_blk_kernel<N, K, E, S>), dim3(bgrid), dim3(128), 0, s, a)
                if (a.sym) {
                    if (epi == 1) MK_LAUNCH_BLK(1, true);
                    else if (epi == 2) MK_LAUNCH_BLK(2, true);
                    else MK_LAUNCH_BLK(0, true);
                } else {
                    if (epi == 1) MK_LAUNCH_BLK(1, false);
                    else if (epi == 2) MK_LAUNCH_BLK(2, false);
                    else MK_LAUNCH_BLK(0, false);
                }
#undef MK_LAUNCH_BLK
                return hipGetLastError();
            }
        }
        if (a.rs > 0) {
#define MK_LAUNCH_REC(E, S) hipLaunchKernelGGL((smoother_record_kernel<N, K, G, E, S>), dim3(grid), dim3(256), 0, s, a)
            if (a.sym) {
                if (epi == 1) MK_LAUNCH_REC(1, true);
                else if (epi == 2) MK_LAUNCH_REC(2, true);
                else MK_LAUNCH_REC(0, true);
            } else {
                if (epi == 1) MK_LAUNCH_REC(1, false);
                else if (epi == 2) MK_LAUNCH_REC(2, false);
                else MK_LAUNCH_REC(0, false);
            }
#undef MK_LAUNCH_REC
        } else
            hipLaunchKernelGGL((smoother_dense_kernel<n, G>), dim3(grid), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

#define MK_CASE_FILTER(NN, KK) \
    if (N == NN && K == KK) return launch_filter_nk<NN, KK>(a, s);
#define MK_CASE_SMOOTH(NN, KK) \
    if (N == NN && K == KK) return launch_smoother_nk<NN, KK>(a, s);
// without projection the smoother depends on n only: any compiled shape of the same state dimension serves
#define MK_CASE_SMOOTH_N(NN, KK) \
    if (N + K == NN + KK && !(a.sim_means || a.sim_vars)) return launch_smoother_nk<NN, KK>(a, s);
#define MK_CASE_LIST(NN, KK) {NN, KK},

hipError_t launch_filter(int N, int K, const FilterArgs &a, hipStream_t s)
{
    MK_SHAPES(MK_CASE_FILTER)
    return hipErrorInvalidValue;
}

hipError_t launch_smoother(int N, int K, const SmootherArgs &a, hipStream_t s)
{
    MK_SHAPES(MK_CASE_SMOOTH)
    MK_SHAPES(MK_CASE_SMOOTH_N)
    return hipErrorInvalidValue;
}

int num_shapes();
void get_shape(int i, int *N, int *K);

#ifdef MK_SHAPE_MODULE
#ifndef MK_SHAPE_MODULE_TUS // one translation unit per run-time shape module (scripts/compile_shape.sh); metran_amd/jit.py compiles
} // namespace mk           // the four files -- mk_wide.hip in its slices -- as separate units in parallel and links them
#include "mk_wide.hip"
#include "mk_split.hip"
#include "mk_dk.hip"
namespace mk {
#endif
// ---------------------------------------------------------------------------------------------
// Shape module: this same translation unit compiled at run time for ONE (N, K) that is not in the
// ahead-of-time list (metran_amd/jit.py drives hipcc, runs the DPP hazard check on the assembly and
// registers the module with mk_register_shape_module).  The kernels are fully unrolled over n, so
// specialising per shape is what makes them fast; this is how an arbitrary Metran model gets one.
// ---------------------------------------------------------------------------------------------
extern "C" {
MK_API int mkmod_abi(void)
{
    return (int)(sizeof(FilterArgs) * 1000 + sizeof(SmootherArgs) + sizeof(AdjointArgs) + sizeof(SparseArgs));
}
MK_API int mkmod_launch_sparse(const SparseArgs *a, void *stream)
{
    int N, K;
    get_shape(0, &N, &K);
    return (int)launch_sparse(N, K, *a, (hipStream_t)stream);
}
MK_API int mkmod_launch_adjoint(const AdjointArgs *a, void *stream)
{
    int N, K;
    get_shape(0, &N, &K);
    return (int)launch_adjoint(N, K, *a, (hipStream_t)stream);
}
MK_API int mkmod_shape(int *N, int *K)
{
    get_shape(0, N, K);
    return num_shapes();
}
MK_API int mkmod_launch_filter(const FilterArgs *a, void *stream)
{
    int N, K;
    get_shape(0, &N, &K);
    return (int)launch_filter(N, K, *a, (hipStream_t)stream);
}
MK_API int mkmod_launch_smoother(const SmootherArgs *a, void *stream)
{
    int N, K;
    get_shape(0, &N, &K);
    return (int)launch_smoother(N, K, *a, (hipStream_t)stream);
}
}
#endif

static const int kShapes[][2] = {MK_SHAPES(MK_CASE_LIST)};

int record_stride(int n) { return record_stride_c(n); }
int record_stride_sym(int n) { return record_stride_sym_c(n); }

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

template <int N, int K>
static hipError_t launch_sparse_nk(const SparseArgs &a, hipStream_t s)
{
    constexpr int n = N + K;
    if constexpr (n <= 16) {
        constexpr int GPB = 256 / 16;
        if (a.rebuild) hipLaunchKernelGGL(observed_steps_kernel, dim3(1), dim3(256), 0, s, a.T, N, a.ostep, a.obs, a.tlist);
        if (a.F && a.Xp) { // record outputs: the observed steps one after the other, then every empty step in parallel
            if (a.rs != record_stride_c(n)) return hipErrorInvalidValue;
            hipLaunchKernelGGL((loglik_sparse_kernel<N, K, 16, true>), dim3((unsigned)((a.B + GPB - 1) / GPB)), dim3(256), 0, s, a);
            hipLaunchKernelGGL(fill_gaps_kernel, dim3((unsigned)a.T, (unsigned)a.B), dim3(256), 0, s, a, n);
            return hipGetLastError();
        }
        hipLaunchKernelGGL((loglik_sparse_kernel<N, K, 16>), dim3((unsigned)((a.B + GPB - 1) / GPB)), dim3(256), 0, s, a);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}
#define MK_CASE_SPARSE(NN, KK) \
    if (N == NN && K == KK) return launch_sparse_nk<NN, KK>(a, s);
hipError_t launch_sparse(int N, int K, const SparseArgs &a, hipStream_t s)
{
    MK_SHAPES(MK_CASE_SPARSE)
    return hipErrorInvalidValue;
}

template <int N, int K>
static hipError_t launch_adjoint_nk(const AdjointArgs &a, hipStream_t s)
{
    constexpr int n = N + K;
    if constexpr (n <= 16) {
        constexpr int GPB = 256 / 16;
        hipLaunchKernelGGL((adjoint_kernel<N, K, 16>), dim3((unsigned)((a.B + GPB - 1) / GPB)), dim3(256), 0, s, a);
        return hipGetLastError();
    } else {
        return launch_adjoint_wide(N, K, a, s); // one model per wavefront: mk_split.hip
    }
}
#define MK_CASE_ADJOINT(NN, KK) \
    if (N == NN && K == KK) return launch_adjoint_nk<NN, KK>(a, s);
hipError_t launch_adjoint(int N, int K, const AdjointArgs &a, hipStream_t s)
{
    MK_SHAPES(MK_CASE_ADJOINT)
    return hipErrorInvalidValue;
}

// chain rule of Metran._phi / get_transition_covariance (metran.py:246-322):
//   phi = exp(-dt/alpha), q = (1 - phi^2) c  =>  d/dalpha = (gphi - 2 phi c gq) phi dt / alpha^2
__global__ void alpha_grad_kernel(long B, long R, int N, int K, const double *alpha, const double *loadings,
                                  double dt, const double *gphi, const double *gq, double *galpha)
{
    const int n = N + K;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n) return;
    const long b = i / n;
    const int s = (int)(i % n);
    const double al = alpha[i], ph = exp(-dt / al);
    double c = 1.0;
    if (s < N) {
        const double *g = loadings + ((b % R) * N + s) * K;
        for (int k = 0; k < K; ++k) c -= g[k] * g[k];
    }
    galpha[i] = (gphi[i] - 2.0 * ph * c * gq[i]) * ph * dt / (al * al);
}
hipError_t launch_alpha_grad(long B, long R, int N, int K, const double *alpha, const double *loadings, double dt,
                             const double *gphi, const double *gq, double *galpha, hipStream_t s)
{
    const long tot = B * (N + K);
    hipLaunchKernelGGL(alpha_grad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, B, R, N, K, alpha,
                       loadings, dt, gphi, gq, galpha);
    return hipGetLastError();
}

hipError_t launch_sum(long count, const double *v, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, s, count, v, out);
    return hipGetLastError();
}

} // namespace mk
