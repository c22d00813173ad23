// mk_split.hip -- sequential Kalman filter for wide models (n = N + K > 16 states) in the SPLIT layout:
// the N series states on the lanes, the K common-factor states replicated -- several models per wavefront.
// Reference semantics: seqkalmanfilter, /root/reference/metran/kalmanfilter.py:236-400 (predict :318-333, scalar
// updates :341-378, compressed bookkeeping :380-382, filtered moments :384-390) and get_mle (:550-567).
//
// Why.  filter_kernel<N,K,64> gives every state a lane: at configs[3] (32 series + 4 factors) 36 of the 64 lanes carry
// data and a wavefront serves ONE model; the kernel is bound by VALU issue (round-2 counters: 2 467 VALU instructions
// per model-step, the f64 rank-one update alone is n of them per observation), so the 28 idle lanes are 44 % of the
// machine.  Metran's state vector is [series states | factor states] (metran.py:283-370) and every covariance is
// SYMMETRIC, so the factor ROWS of P are the factor COLUMNS of the series rows: lane l of a group of H = 16 or 32 lanes
// holds row l of P for the series l < N -- all n columns, i.e. including P[l][N+k] -- and the only part no series row
// contains, the K x K factor block, is replicated in every lane of the group (K(K+1)/2 doubles), as are the K factor
// means.  A wavefront then serves 64 / H models (two at configs[3], four for N <= 16 such as the (14,3) fixture) with
// the same instruction stream per update:
//     innovation      v_l = y_l - x_l - sum_k g_lk xf_k                     (own loadings; lane j's value is gathered)
//     d = P Z_j^T     d_l = P[l][j] + sum_k P[l][N+k] g_jk                  (series part; P[l][j] by a uniform switch)
//                     d_{N+k} = P[j][N+k] + sum_k' PF[k][k'] g_jk'          (valid at lane j, which parks it in LDS)
//     f = R_j + Z_j d = R_j + d_j + sum_k g_jk d_{N+k}                      (all own-lane quantities at lane j)
//     P[l][c] -= d_c d_l / f  for all n columns c                           (d_c: group-uniform LDS reads, as before)
//     PF[k][k'] -= d_{N+k} d_{N+k'} / f,  xf_k += d_{N+k} v / f             (replicated, K(K+1)/2 + K multiply-adds)
// The models of a wavefront observe different series: the update loop runs over the union of their set bits, one set
// bit of EVERY model per iteration (max instead of mean count: +7 % iterations at configs[3]'s 30 % missing), a model
// whose bits are exhausted runs the body as a no-op (1/f := 0, v := 0).
// The records written are the same full-square [ mean(n) | covariance(n x n) | sigma, detf | pad ] records the smoothers
// read (mk_prims.h): rows 0 .. n-1, columns < N through the symmetric column runs of the series rows; columns >= N of
// the series rows as K contiguous doubles per lane; the factor block from the replicated copy.
// Modes: OUT 0 (objective only), 1 (predicted + filtered records), 3 (filtered record), full-square or (SYM, round 4)
// packed-symmetric records; dense outputs keep filter_kernel<N,K,64>.
// OUT 4 (round 4): the BACKWARD TAPE of the inverse-free smoother (mk_dk.hip) instead of a covariance record -- per
// (step, series) one entry of n + 4 doubles in the OBSERVABLE basis xt = T x, T = [[I, G], [0, I]] (series states
// replaced by the observables y_j = x_j + sum_k g_jk x_{N+k}, in which the observation rows are unit vectors):
//     observed series j     [ kt = T k (n) | v/f | 1/f | y_j | 0 ]   k = P z_j^T / f, the gain of that scalar update
//     unobserved series u   [ pt = T Pf z_u^T (n) | z_u x_f | z_u Pf z_u^T | NaN | 0 ]   (end of the step)
// Block of one (model, step) in HBM: mk_tape_stride = N (n + 4) doubles, the N entries one after the other (mk_internal.h);
// the vector's series part is one 8 N-byte run per store instruction and model, the side row (factor part and scalars) is
// written by lane j alone.
// T k costs K multiply-adds a lane (kt_l = k_l + sum_k g_lk k_{N+k}: own loadings, replicated factor gains); the
// unobserved entries are one extra pass per unobserved series and model (column pick, d = Pf z_u^T as in an update,
// no rank-one update).  tests/dk_ref.py::filter_tape is the numpy restatement.
#include "mk_prims.h"
#include "mk_jump.h"

namespace mk {

// LaneRow: a lane's row of the covariance in the split-layout kernels -- the N <= 32 series columns as two 16-double VECTOR values
// pinned to v[0:31] and v[32:63] at the pick (the register allocator then keeps them there for the whole kernel: no copies in the
// generated code), the K factor columns as scalars.  The run-time column j of a scalar update (d = P z_j', kalmanfilter.py:349-357)
// is then ONE move under the VGPR index mode -- s_set_gpr_idx_on / v_mov_b64 v[0:1] + 2 j / s_set_gpr_idx_off, the models of the
// wavefront one after the other under their lanes' exec masks: 14 scalar / vector instructions a pass for two models and no
// branch.  Until the end of round 6 the column came through the jump tables of mk_jump.h (pick_column: 14 + 3 scalar
// instructions and three taken branches per model and pick -- 12 % of filter_obs_kernel's time by a timing build without the
// picks; hiding them under the LDS round trip bought nothing, the kernel pays per ISSUED instruction:
// profiles/r06/ab_filter_pick_*.log).  hipcc's own lowering of a uniform index into a vector (two idx_on / idx_off sections,
// four moves and four selects per model) got a third of the gain; an ARRAY of vectors, or the indexed read inside a lambda,
// sends the row to scratch memory.
typedef double v16d __attribute__((ext_vector_type(16)));
template <int N, int K>
struct LaneRow {
    static constexpr int n = N + K;
    static_assert(N <= 32, "two 16-double tuples");
    v16d s0, s1; // columns 0 .. 15, 16 .. 31 (two MEMBERS: an array of vectors is an alloca hipcc leaves in scratch memory)
    double f[K];
    __device__ __forceinline__ double get(int c) const { return c < N ? (c < 16 ? s0[c & 15] : s1[c & 15]) : f[c - N]; }
    __device__ __forceinline__ void set(int c, double v)
    {
        if (c < N) {
            if (c < 16) s0[c & 15] = v;
            else s1[c & 15] = v;
        } else {
            f[c - N] = v;
        }
    }
    // lo / hi: d_0 .. d_15 / d_16 .. d_31 of the lane's model in EVERY 16-lane row of the model, from d_l in lane l.  H = 32: the two
    // rows of a model trade places through v_permlane16_swap (odd rows of the first operand <-> even rows of the second: with both
    // operands copies of d the first comes back as [row 0, row 0, row 2, row 2], the second as [row 1, row 1, row 3, row 3]).
    template <int H>
    static __device__ __forceinline__ void split_rows(double d, double &lo, double &hi)
    {
        if constexpr (H == 32) {
            const unsigned a = (unsigned)__double2loint(d), b = (unsigned)__double2hiint(d);
            const auto r0 = __builtin_amdgcn_permlane16_swap(a, a, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(b, b, false, false);
            lo = __hiloint2double((int)r1[0], (int)r0[0]);
            hi = __hiloint2double((int)r1[1], (int)r0[1]);
        } else {
            lo = hi = d;
        }
    }
    // series columns of the rank-one update  P[l][c] += d_c m  (m = -k_l: kalmanfilter.py:368-372), d_c broadcast inside the rows
    // by v_fmac_f64_dpp row_newbcast -- one instruction per column on the pinned tuples, no LDS read of d
    template <int BASE, int C>
    static __device__ __forceinline__ void fmac_col(v16d &t, double s, double m) // element C of the tuple at v[BASE : BASE + 31]
    {
        if constexpr (BASE == 0)
            asm volatile("v_fmac_f64_dpp v[%[r0]:%[r1]], %[s], %[m] row_newbcast:%[j]" MK_DPPMASK
                         : [t] "+{v[0:31]}"(t) : [r0] "n"(2 * C), [r1] "n"(2 * C + 1), [s] "v"(s), [m] "v"(m), [j] "n"(C));
        else
            asm volatile("v_fmac_f64_dpp v[%[r0]:%[r1]], %[s], %[m] row_newbcast:%[j]" MK_DPPMASK
                         : [t] "+{v[32:63]}"(t) : [r0] "n"(32 + 2 * C), [r1] "n"(32 + 2 * C + 1), [s] "v"(s), [m] "v"(m), [j] "n"(C));
    }
    __device__ __forceinline__ void rank_one(double lo, double hi, double m)
    {
        sfor<0, (N < 16 ? N : 16)>(MK_LAMBDA(cc) { fmac_col<0, decltype(cc)::value>(s0, lo, m); });
        if constexpr (N > 16) sfor<16, N>(MK_LAMBDA(cc) { fmac_col<32, decltype(cc)::value - 16>(s1, hi, m); });
    }
    // dst (lanes of model g) = column j[g] of the own row; j[g] wavefront-uniform, < N.  H lanes per model, 64 / H models.
    template <int H>
    __device__ __forceinline__ void pick(double &dst, const int (&j)[64 / H]) const
    {
        int t;
        unsigned long long sv;
        if constexpr (H == 32) {
            asm volatile("s_mov_b64 %[sv], exec\n\t"
                         "s_lshl_b32 %[t], %[j0], 1\n\t"
                         "s_mov_b32 exec_lo, -1\n\t"
                         "s_mov_b32 exec_hi, 0\n\t"
                         "s_set_gpr_idx_on %[t], gpr_idx(SRC0)\n\t"
                         "v_mov_b64 %[d], v[0:1]\n\t"
                         "s_lshl_b32 %[t], %[j1], 1\n\t"
                         "s_set_gpr_idx_idx %[t]\n\t"
                         "s_mov_b32 exec_lo, 0\n\t"
                         "s_mov_b32 exec_hi, -1\n\t"
                         "v_mov_b64 %[d], v[0:1]\n\t"
                         "s_set_gpr_idx_off\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [d] "+v"(dst), [t] "=&s"(t), [sv] "=&s"(sv)
                         : [j0] "s"(j[0]), [j1] "s"(j[1]), "{v[0:31]}"(s0), "{v[32:63]}"(s1)
                         : "scc");
        } else {
            static_assert(H == 16 && N <= 16, "four models of at most 16 series");
            asm volatile("s_mov_b64 %[sv], exec\n\t"
                         "s_lshl_b32 %[t], %[j0], 1\n\t"
                         "s_mov_b32 exec_lo, 0xffff\n\t"
                         "s_mov_b32 exec_hi, 0\n\t"
                         "s_set_gpr_idx_on %[t], gpr_idx(SRC0)\n\t"
                         "v_mov_b64 %[d], v[0:1]\n\t"
                         "s_lshl_b32 %[t], %[j1], 1\n\t"
                         "s_set_gpr_idx_idx %[t]\n\t"
                         "s_mov_b32 exec_lo, 0xffff0000\n\t"
                         "v_mov_b64 %[d], v[0:1]\n\t"
                         "s_lshl_b32 %[t], %[j2], 1\n\t"
                         "s_set_gpr_idx_idx %[t]\n\t"
                         "s_mov_b32 exec_lo, 0\n\t"
                         "s_mov_b32 exec_hi, 0xffff\n\t"
                         "v_mov_b64 %[d], v[0:1]\n\t"
                         "s_lshl_b32 %[t], %[j3], 1\n\t"
                         "s_set_gpr_idx_idx %[t]\n\t"
                         "s_mov_b32 exec_hi, 0xffff0000\n\t"
                         "v_mov_b64 %[d], v[0:1]\n\t"
                         "s_set_gpr_idx_off\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [d] "+v"(dst), [t] "=&s"(t), [sv] "=&s"(sv)
                         : [j0] "s"(j[0]), [j1] "s"(j[1]), [j2] "s"(j[2]), [j3] "s"(j[3]), "{v[0:31]}"(s0)
                         : "scc");
        }
    }
};


template <int N, int K, int H, int OUT, bool BOOK, bool SYM = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) filter_split_kernel(FilterArgs a)
{
    constexpr int n = N + K, M = 64 / H;
    static_assert(N <= H && (H == 16 || H == 32) && n > 16, "split layout: N series on H lanes, wide models");
    constexpr int NP = n + (n & 1);              // even: rows of 16-byte pieces
    constexpr int GT = (N * K + 1) & ~1;         // loadings table of one model, even
    constexpr int KF = K * (K + 1) / 2;          // upper triangle of the factor block
    constexpr int KK2 = (K * K + 1) & ~1;
    static_assert(!SYM || OUT == 1 || OUT == 3, "packed-symmetric layout applies to record outputs");
    constexpr int NV = SYM ? record_payload_sym(n) : record_payload(n), RS = SYM ? record_stride_sym_c(n) : record_stride_c(n), PADN = RS - NV;
    constexpr bool RECF = (OUT == 1 || OUT == 3);
    constexpr bool TAPE = (OUT == 4);
    constexpr int SW = tape_side_c(K);           // side row of a tape entry: [ factor part (K) | s0 | s1 | s2 | 0 ]
    constexpr int XS = tape_xs_c(N, K), SS = tape_ss_c(N, K), SO = tape_so_c(N, K); // tape block addressing (mk_internal.h)
    constexpr bool PAIRS = (K % 2 == 0 && N % 2 == 0); // 16-byte accesses to the factor part / the entry's scalars
    constexpr unsigned NM = N >= 32 ? 0xffffffffu : ((1u << (N & 31)) - 1u);
    constexpr int TS = 16;                       // observation tile: time steps per LDS refill
    constexpr unsigned HM = H == 32 ? 0xffffffffu : 0xffffu;
    auto pf = [](int k, int k2) constexpr { return k * K - k * (k - 1) / 2 + (k2 - k); }; // k <= k2

    const int lane = threadIdx.x, h = lane / H, l = lane % H;
    const int jr = l < N ? l : N - 1;            // lanes >= N of a group replicate lane N-1 (identical stores)
    long inst = (long)blockIdx.x * M + h;
    if (inst > a.B - 1) inst = a.B - 1;          // surplus groups replicate the last model
    const long rec = inst % a.R;
    const long T = a.T;

    // wave-private LDS, one slice per model of the wavefront
    constexpr int KP = (K + 1) & ~1;
    constexpr int DVS = NP + 2;                  // a d buffer: d [n] (+ pad), then the innovation v and its variance f of lane j
    __shared__ __attribute__((aligned(16))) double lds[M * (NP + 2 * DVS + GT + KK2 + KP + TS * N + (TAPE ? N * KP : 0))];
    double *phim = lds + h * NP;                                   // diag(Phi) [n]
    double *dbuf = lds + M * NP + h * 2 * DVS;                     // d = P Z_j^T, two buffers
    double *gtab = lds + M * (NP + 2 * DVS) + h * GT;              // loadings [N][K]
    double *pfs = lds + M * (NP + 2 * DVS + GT) + h * KK2;         // factor block staging for the record stores
    double *qtab = lds + M * (NP + 2 * DVS + GT + KK2) + h * KP;   // diag(Q) of the factor states
    double *otile = lds + M * (NP + 2 * DVS + GT + KK2 + KP) + h * TS * N; // observations of TS steps
    [[maybe_unused]] double *fct = lds + M * (NP + 2 * DVS + GT + KK2 + KP + TS * N) + h * N * KP; // TAPE: factor columns Pf[l][N+k]

    const double phi_l = a.phi[inst * n + jr];
    const double q_l = a.q[inst * n + jr];
    double gam[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gam[k] = a.loadings[(rec * N + jr) * K + k];
    const double rvar = a.obsvar ? a.obsvar[rec * N + jr] : 0.0;
    phim[jr] = phi_l;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        phim[N + k] = a.phi[inst * n + N + k];   // every lane of the group writes the same value
        gtab[jr * K + k] = gam[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) qtab[k] = a.q[inst * n + N + k];
    wave_lds_sync();

    // initial state (run_filter defaults, kalmanfilter.py:747-750)
    double x = a.x0 ? a.x0[inst * n + jr] : 0.0;
    double xk[K], PF[KF];
    LaneRow<N, K> P;
#pragma unroll
    for (int k = 0; k < K; ++k) xk[k] = a.x0 ? a.x0[inst * n + N + k] : 0.0;
#pragma unroll
    for (int c = 0; c < n; ++c) P.set(c, a.P0 ? a.P0[(inst * n + jr) * n + c] : (c == jr ? 1.0 : 0.0));
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int k2 = k; k2 < K; ++k2) PF[pf(k, k2)] = a.P0 ? a.P0[(inst * n + N + k) * n + N + k2] : (k == k2 ? 1.0 : 0.0);

    const double *obase = a.obs + rec * a.obs_bs * N + jr;
    const long ostep = a.obs_ts * N;
    double *recP = (OUT == 1) ? a.Xp + inst * a.bs * RS : nullptr;
    double *recF = RECF ? a.F + inst * a.bs * RS : nullptr;
    const long rstep = a.ts * RS;
    double *trec = TAPE ? a.F + inst * a.bs * a.rs : nullptr; // tape block of (model, step): N entries of XS doubles
    const long tstep = a.ts * a.rs;
    const int kl = l < K ? l : K - 1;            // factor state whose mean this lane writes
    const int fl16 = l < K * K ? l : K * K - 1;  // factor-block element this lane writes

    // one (model, step) record: [ mean | covariance, row-major | p0, p1, zeros ]
    auto emit = [&](double *r, double xv, const double(&xf)[K], const LaneRow<N, K> &Pr, const double(&PFv)[KF], double p0, double p1)
                    __attribute__((always_inline)) {
        r[jr] = xv;
        double xm = xf[0];
#pragma unroll
        for (int k = 1; k < K; ++k) xm = (kl == k) ? xf[k] : xm;
        r[N + kl] = xm;
        double *cov = r + n;
        if constexpr (SYM) {
            // packed upper triangle by rows (mk_prims.h: element (c, r), c <= r, at sym_row_offset(n, c) + r - c): lane l writes
            // its P[l][c] to (c, l) for c <= l -- for a fixed c one contiguous run of the lanes l >= c --, its factor columns
            // P[l][N+k] as the tail of row l; the factor block comes from the replicated copy
#pragma unroll
            for (int c = 0; c < N; ++c)
                if (jr >= c) cov[sym_row_offset(n, c) + (jr - c)] = Pr.get(c);
            const int rowl = jr * n - (jr * (jr - 1)) / 2 - jr;         // sym_row_offset(n, l) - l
#pragma unroll
            for (int k = 0; k < K; ++k) cov[rowl + N + k] = Pr.get(N + k);
            wave_lds_sync();                                            // the previous record's staging reads are done
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int k2 = k; k2 < K; ++k2) pfs[pf(k, k2)] = PFv[pf(k, k2)];
            wave_lds_sync();
            // rows N+k of the triangle are contiguous and in the order of pf(): element e of the replicated block, one
            // element per lane and pass (KF > H for K >= 6 at H = 16, K >= 8 at H = 32: run-time-specialised shapes)
#pragma unroll
            for (int p = 0; p < (KF + H - 1) / H; ++p) {
                const int e = l + p * H < KF ? l + p * H : KF - 1;
                cov[sym_row_offset(n, N) + e] = pfs[e];
            }
        } else {
#pragma unroll
            for (int c = 0; c < n; ++c) cov[c * n + jr] = Pr.get(c);            // (c, l) <- P[l][c]: a contiguous run per model
            if constexpr (K % 2 == 0 && N % 2 == 0) {                       // (l, N+k) <- P[l][N+k]: K contiguous doubles
#pragma unroll
                for (int k = 0; k < K; k += 2) *reinterpret_cast<v2d *>(cov + jr * n + N + k) = v2d{Pr.get(N + k), Pr.get(N + k + 1)};
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) cov[jr * n + N + k] = Pr.get(N + k);
            }
            wave_lds_sync();                                                // the previous record's staging reads are done
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int k2 = 0; k2 < K; ++k2) pfs[k * K + k2] = PFv[k <= k2 ? pf(k, k2) : pf(k2, k)]; // same value from every lane
            wave_lds_sync();
#pragma unroll
            for (int p = 0; p < (K * K + H - 1) / H; ++p) { // one element per lane and pass (K * K > H for K >= 5 at H = 16)
                const int e = fl16 + p * H < K * K ? fl16 + p * H : K * K - 1;
                cov[(N + e / K) * n + N + e % K] = pfs[e];
            }
        }
#pragma unroll
        for (int p = 0; p < (PADN + H - 1) / H; ++p) { // the record's pad: sigma, detf, zeros (whole cache lines)
            const int slot = l + p * H < PADN ? l + p * H : PADN - 1;
            r[NV + slot] = slot == 0 ? p0 : (slot == 1 ? p1 : 0.0);
        }
    };

    double sum_sig = 0.0, sum_det = 0.0;
    double run_mant = 1.0; // !BOOK: product of f over the counted steps, normalised
    long run_exp = 0;
    long nobs = 0, sc = 0;
    double fmin_seen = 1.0;

    for (long t0 = 0; t0 < T; t0 += TS) {
        wave_lds_sync(); // the previous tile's reads are complete
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            long tr = t0 + s;
            if (tr > T - 1) tr = T - 1;
            otile[s * N + jr] = obase[tr * ostep];
        }
        wave_lds_sync();
        const long tend = t0 + TS < T ? t0 + TS : T;
        for (long t = t0; t < tend; ++t) {
            const double y = otile[(int)(t - t0) * N + jr];
            // which series are observed at this step (NaN / inf = missing, kalmanfilter.py:657)
            const unsigned long long ball = __ballot(l < N && isfinite(y));
            unsigned mrem[M];
#pragma unroll
            for (int g = 0; g < M; ++g) mrem[g] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(ball >> (g * H)) & HM));
            unsigned maskl = mrem[0];
#pragma unroll
            for (int g = 1; g < M; ++g) maskl = (h == g) ? mrem[g] : maskl;
            const int cnt = __popc(maskl);

            // ---- predict (:318-331; Phi diagonal) ----
            {
                int jv = jr; // opaque copies: keeps the n selects inside the loop (hoisted, they are 2n VGPRs)
                double qv = q_l;
                asm volatile("" : "+v"(jv), "+v"(qv));
                double phc[n];
                load_row<n>(phim, phc);
                x = phi_l * x;
#pragma unroll
                for (int k = 0; k < K; ++k) xk[k] = phc[N + k] * xk[k];
#pragma unroll
                for (int c = 0; c < n; ++c) P.set(c, fma(P.get(c) * phi_l, phc[c], c == jv ? qv : 0.0));
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int k2 = k; k2 < K; ++k2)
                        PF[pf(k, k2)] = fma(PF[pf(k, k2)] * phc[N + k], phc[N + k2], k == k2 ? qtab[k] : 0.0);
            }
            if constexpr (OUT == 1) {
                emit(recP, x, xk, P, PF, 0.0, 0.0); // :332-333
                recP += rstep;
            }

            // ---- sequential scalar updates (:341-378), ascending series order, one observation of every model per pass ----
            double sigma = 0.0, fmant = 1.0;
            int fexp = 0, nupd = 0;
            unsigned many = 0;
            [[maybe_unused]] unsigned urem[M]; // TAPE: the series NOT observed at this step, per model
#pragma unroll
            for (int g = 0; g < M; ++g) urem[g] = ~mrem[g] & NM;
#pragma unroll
            for (int g = 0; g < M; ++g) many |= mrem[g];
            while (many) {
                // the pass's bookkeeping on the scalar unit and exec = the lanes of the models that still have an observation
                // (the same block as filter_obs_kernel's, where it is explained and was measured: -4 % of the launch)
                int jsel[M];
                unsigned long long okm = 0, jm = 0;
#pragma unroll
                for (int g = 0; g < M; ++g) {
                    unsigned vs, tt;
                    asm volatile("s_ff1_i32_b32 %[j], %[m]\n\t"
                                 "s_cmp_lg_u32 %[m], 0\n\t"
                                 "s_cselect_b32 %[j], %[j], 0\n\t"
                                 "s_cselect_b32 %[v], -1, 0\n\t"
                                 "s_add_i32 %[t], %[m], -1\n\t"
                                 "s_and_b32 %[m], %[m], %[t]"
                                 : [j] "=&s"(jsel[g]), [v] "=&s"(vs), [t] "=&s"(tt), [m] "+s"(mrem[g])
                                 :
                                 : "scc");
                    okm |= (unsigned long long)(vs & HM) << (g * H);
                    jm |= (unsigned long long)(vs & (1u << jsel[g])) << (g * H);
                }
                many = 0;
#pragma unroll
                for (int g = 0; g < M; ++g) many |= mrem[g];
                double *dv = dbuf + (nupd & 1) * DVS; // two buffers: one fence per update
                ++nupd;
                if (__builtin_amdgcn_inverse_ballot_w64(okm)) {
                const bool isj = __builtin_amdgcn_inverse_ballot_w64(jm); // l == j of the lane's model
                int jl = jsel[0];
#pragma unroll
                for (int g = 1; g < M; ++g) jl = (h == g) ? jsel[g] : jl;
                // innovation (:344-347): every lane forms v_l with ITS loadings; lane j's value is the model's
                double vl = y - x, vl2 = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (k % 2 == 0) vl = fma(-gam[k], xk[k], vl);
                    else vl2 = fma(-gam[k], xk[k], vl2);
                }
                vl += vl2;
                // d = P Z_j^T (:349-357): column j of the own row through a wavefront-uniform switch per model
                double dr = 0.0;
                P.template pick<H>(dr, jsel); // lanes of group g: dr = P[l][j_g] (VGPR index mode, exec = the group's lanes: LaneRow)
                double gj[K];
                {
                    const double *gp = gtab + jl * K; // loadings of series j of the lane's model
                    if constexpr (K % 2 == 0) {
#pragma unroll
                        for (int k = 0; k < K; k += 2) {
                            const v2d t2 = *reinterpret_cast<const v2d *>(gp + k);
                            gj[k] = t2.x;
                            gj[k + 1] = t2.y;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < K; ++k) gj[k] = gp[k];
                    }
                }
                double df[K]; // d_{N+k}: right at lane l == j (its P[l][N+k] is P[N+k][j])
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    dr = fma(P.get(N + k), gj[k], dr);
                    double s = P.get(N + k);
#pragma unroll
                    for (int k2 = 0; k2 < K; ++k2) s = fma(PF[k <= k2 ? pf(k, k2) : pf(k2, k)], gj[k2], s);
                    df[k] = s;
                }
                double dlo, dhi;               // d in every row of the model: the operands of the rank-one update
                LaneRow<N, K>::template split_rows<H>(dr, dlo, dhi);
                // innovation variance f = R_j + Z_j d (:359-362), all from lane j's own values
                double fl = rvar + dr, fl2 = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (k % 2 == 0) fl = fma(gam[k], df[k], fl);
                    else fl2 = fma(gam[k], df[k], fl2);
                }
                fl += fl2;
                if (isj) {
#pragma unroll
                    for (int k = 0; k < K; ++k) dv[N + k] = df[k];
                    // lane j's innovation and its variance travel with d through LDS (round 4: gathering them with two readlane
                    // pairs + selects per model cost the wide filter 6 %: 52.9 -> 49.8 ms at configs[3])
                    *reinterpret_cast<v2d *>(dv + NP) = v2d{vl, fl};
                }
                wave_lds_sync();
                const v2d vf = *reinterpret_cast<const v2d *>(dv + NP);
                const double v = vf.x, f = vf.y;
                const double rf = rcp_nr(f);
                const double kr = dr * rf;     // Kalman gain element l (:364-366)
                {
                    // P -= k k^T f (:368-372): P[l][c] -= d_c k_l.  The series columns by broadcast multiply-adds inside the rows of
                    // the model (LaneRow::rank_one; until the end of round 6 d was read back from LDS in pieces), the factor columns
                    // from lane j's entries
                    double kf[K], dfk[K];
                    {   // the factor entries first: they feed the replicated block and the factor means
                        double dfa[K];
                        if constexpr (K % 2 == 0 && N % 2 == 0) {
#pragma unroll
                            for (int k = 0; k < K; k += 2) {
                                const v2d t2 = *reinterpret_cast<const v2d *>(dv + N + k);
                                dfa[k] = t2.x;
                                dfa[k + 1] = t2.y;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < K; ++k) dfa[k] = dv[N + k];
                        }
#pragma unroll
                        for (int k = 0; k < K; ++k) kf[k] = dfa[k] * rf;
#pragma unroll
                        for (int k = 0; k < K; ++k) dfk[k] = dfa[k];
#pragma unroll
                        for (int k = 0; k < K; ++k) {
#pragma unroll
                            for (int k2 = k; k2 < K; ++k2) PF[pf(k, k2)] = fma(-dfa[k], kf[k2], PF[pf(k, k2)]);
                            xk[k] = fma(kf[k], v, xk[k]);
                        }
                    }
                    if constexpr (TAPE) { // tape entry of the observed series jl: the gain in the observable basis
                        double kt = kr;
#pragma unroll
                        for (int k = 0; k < K; ++k) kt = fma(gam[k], kf[k], kt);
                        if (!MK_TUNE_SKIP(a, 1)) {
                            trec[jl * XS + jr] = kt;
                            if (isj) {
                                double *sd = trec + SO + l * SS;
                                if constexpr (PAIRS) {
#pragma unroll
                                    for (int k = 0; k < K; k += 2) *reinterpret_cast<v2d *>(sd + k) = v2d{kf[k], kf[k + 1]};
                                    *reinterpret_cast<v2d *>(sd + SW - 4) = v2d{v * rf, rf};
                                    *reinterpret_cast<v2d *>(sd + SW - 2) = v2d{y, 0.0};
                                } else {
#pragma unroll
                                    for (int k = 0; k < K; ++k) sd[k] = kf[k];
                                    sd[SW - 4] = v * rf;
                                    sd[SW - 3] = rf;
                                    sd[SW - 2] = y;
                                    sd[SW - 1] = 0.0;
                                }
                            }
                        }
                    }
                    P.rank_one(dlo, dhi, -kr);
#pragma unroll
                    for (int k = 0; k < K; ++k) P.set(N + k, fma(-dfk[k], kr, P.get(N + k)));
                }
                x = fma(kr, v, x);             // :374-375
                sigma = fma(v * v, rf, sigma); // :377
                fmant *= f;                    // detf += log f (:378) as mantissa * 2^exp
                fexp += __builtin_amdgcn_frexp_exp(fmant);
                fmant = __builtin_amdgcn_frexp_mant(fmant);
                fmin_seen = min_f64(fmin_seen, f);
                } // lanes of the valid models
            }
            fexp += __builtin_amdgcn_frexp_exp(fmant);
            fmant = __builtin_amdgcn_frexp_mant(fmant);

            double pad0 = 0.0, pad1 = 0.0;
            if (cnt > 0) { // :380-382 compressed bookkeeping (per model: lanes of a group agree)
                if constexpr (BOOK) {
                    int le;
                    const double lm = log_mant(fmant, le);
                    const double detf = fma((double)(fexp + le), kLn2, lm);
                    if constexpr (RECF) {
                        // compressed entry sc lives in the pad of filtered record sc; sc == t unless an earlier step
                        // of this model was empty (then: one scattered 16-byte store, rare)
                        if (sc == t) {
                            pad0 = sigma;
                            pad1 = detf;
                        } else if (l == 0 && a.sigmas)
                            *reinterpret_cast<v2d *>(a.F + (inst * a.bs + sc * a.ts) * RS + NV) = v2d{sigma, detf};
                    } else {
                        if (a.sigmas && l == 0) a.sigmas[(inst * a.bs + sc * a.ts) * a.sig_stride] = sigma;
                        if (a.detfs && l == 0) a.detfs[(inst * a.bs + sc * a.ts) * a.sig_stride] = detf;
                    }
                    if (sc >= a.warmup) { // get_mle: COMPRESSED indices (:563-564)
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

            if constexpr (RECF) {
                emit(recF, x, xk, P, PF, pad0, pad1); // :384-390
                recF += rstep;
            }
            if constexpr (TAPE) {
                // entries of the series NOT observed at this step, from the filtered moments: pt = T Pf z_u^T (what the
                // backward pass multiplies r and N with), the filtered observable z_u x_f and its variance z_u Pf z_u^T.
                //   pt_l = Pf[l][u] + sum_k g_uk Q[l][k] + sum_k g_lk Pf[u][N+k],   Q[l][k] = Pf[l][N+k] + sum_k' g_lk' PF[k'][k]
                // Q (= the factor columns of T Pf) is formed once per step; Pf[u][N+k] comes from a table every lane fills
                // with its own factor columns; at lane u, Q[u][.] IS the factor part of pt and pt_u IS z_u Pf z_u^T.
                unsigned uany = 0;
#pragma unroll
                for (int g = 0; g < M; ++g) uany |= urem[g];
                const bool stape = a.tape == 2; // STATE tape: K more entries per step, the factor columns of T Pf T'
                if ((uany || stape) && !MK_TUNE_SKIP(a, 2)) {
                    double Q[K], yh = x;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        yh = fma(gam[k], xk[k], yh);
                        double sq = P.get(N + k);
#pragma unroll
                        for (int k2 = 0; k2 < K; ++k2) sq = fma(gam[k2], PF[k <= k2 ? pf(k, k2) : pf(k2, k)], sq);
                        Q[k] = sq;
                    }
                    if (stape) {
                        // entry N + k = [ Q[.][k] (series part) | PF[.][k] | x_f[N+k] | PF[k][k] | NaN | 0 ]: column N + k of T Pf T'
                        // (T' e_{N+k} = e_{N+k}), the filtered factor mean and its variance.  One 8 N-byte run per model for
                        // the series part; the side row is replicated in every lane, lane 0 of the group writes it.
                        const double qnan = __builtin_nan("");
                        sfor<0, K>(MK_LAMBDA(kk) {
                            constexpr int k = decltype(kk)::value;
                            trec[(N + k) * XS + jr] = Q[k];
                            if (l == 0) {
                                double *sd = trec + SO + (N + k) * SS;
                                if constexpr (PAIRS) { // 16-byte stores: the side rows are 16-byte aligned
#pragma unroll
                                    for (int k2 = 0; k2 < K; k2 += 2)
                                        *reinterpret_cast<v2d *>(sd + k2) = v2d{PF[k2 <= k ? pf(k2, k) : pf(k, k2)], PF[k2 + 1 <= k ? pf(k2 + 1, k) : pf(k, k2 + 1)]};
                                    *reinterpret_cast<v2d *>(sd + SW - 4) = v2d{xk[k], PF[pf(k, k)]};
                                    *reinterpret_cast<v2d *>(sd + SW - 2) = v2d{qnan, 0.0};
                                } else {
#pragma unroll
                                    for (int k2 = 0; k2 < K; ++k2) sd[k2] = PF[k2 <= k ? pf(k2, k) : pf(k, k2)];
                                    sd[SW - 4] = xk[k];
                                    sd[SW - 3] = PF[pf(k, k)];
                                    sd[SW - 2] = qnan;
                                    sd[SW - 1] = 0.0;
                                }
                            }
                        });
                    }
                    if (uany) {
                    wave_lds_sync(); // the previous step's table reads are done
                    if constexpr (K % 2 == 0) {
#pragma unroll
                        for (int k = 0; k < K; k += 2) *reinterpret_cast<v2d *>(fct + jr * KP + k) = v2d{P.get(N + k), P.get(N + k + 1)};
                    } else {
#pragma unroll
                        for (int k = 0; k < K; ++k) fct[jr * KP + k] = P.get(N + k);
                    }
                    wave_lds_sync();
                    // static loop over the series (the column P[.][c] is a named register, the loadings / factor-column
                    // rows of series c sit at immediate LDS offsets): a wavefront-uniform branch skips the series every model
                    // of the wavefront observed; a model that observed c computes along and does not store
                    unsigned uml = urem[0];
#pragma unroll
                    for (int g = 1; g < M; ++g) uml = (h == g) ? urem[g] : uml;
                    sfor<0, N>(MK_LAMBDA(cc) {
                        constexpr int c = decltype(cc)::value;
                        if (uany & (1u << c)) {
                            const double *gp = gtab + c * K, *fp = fct + c * KP;
                            double pt = P.get(c), pt2 = 0.0;
                            if constexpr (K % 2 == 0) {
#pragma unroll
                                for (int k = 0; k < K; k += 2) {
                                    const v2d g2 = *reinterpret_cast<const v2d *>(gp + k), f2 = *reinterpret_cast<const v2d *>(fp + k);
                                    pt = fma(g2.x, Q[k], pt);
                                    pt2 = fma(g2.y, Q[k + 1], pt2);
                                    pt = fma(gam[k], f2.x, pt);
                                    pt2 = fma(gam[k + 1], f2.y, pt2);
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < K; ++k) {
                                    pt = fma(gp[k], Q[k], pt);
                                    pt2 = fma(gam[k], fp[k], pt2);
                                }
                            }
                            pt += pt2;
                            if (((uml >> c) & 1u) && !MK_TUNE_SKIP(a, 4)) {
                                trec[c * XS + jr] = pt;
                                if (l == c) {
                                    const double qnan = __builtin_nan("");
                                    double *sd = trec + SO + c * SS;
                                    if constexpr (PAIRS) {
#pragma unroll
                                        for (int k = 0; k < K; k += 2) *reinterpret_cast<v2d *>(sd + k) = v2d{Q[k], Q[k + 1]};
                                        *reinterpret_cast<v2d *>(sd + SW - 4) = v2d{yh, pt};
                                        *reinterpret_cast<v2d *>(sd + SW - 2) = v2d{qnan, 0.0};
                                    } else {
#pragma unroll
                                        for (int k = 0; k < K; ++k) sd[k] = Q[k];
                                        sd[SW - 4] = yh;
                                        sd[SW - 3] = pt;
                                        sd[SW - 2] = qnan;
                                        sd[SW - 1] = 0.0;
                                    }
                                }
                            }
                        }
                    });
                    } // uany
                }
                trec += tstep;
            }
        }
    }

    // zero tail of the compressed arrays (np.zeros init, :307-308); record pads were written as zeros
    if (BOOK && !RECF) {
        for (long i = sc + l; i < T; i += H) {
            if (a.sigmas) a.sigmas[(inst * a.bs + i * a.ts) * a.sig_stride] = 0.0;
            if (a.detfs) a.detfs[(inst * a.bs + i * a.ts) * a.sig_stride] = 0.0;
        }
    }
    if (l == 0) {
        if (!BOOK) sum_det = fma((double)run_exp, kLn2, log(run_mant));
        if (a.mle) a.mle[inst] = ((double)nobs * kLog2Pi + sum_det) + sum_sig; // :566
        if (a.sigmacount) a.sigmacount[inst] = sc;
        if (a.status) a.status[inst] = (fmin_seen > 0.0) ? 0u : MK_FLAG_NONPOSITIVE_F; // NaN f also flags
    }
}

// =====================================================================================
// The TAPE filter in the OBSERVABLE basis (round 6).  filter_split_kernel<.., OUT = 4> above filters in Metran's own state basis
// (series states | factor states) and converts what it writes -- per scalar update the gain T k, per unobserved series the
// column T Pf z_u' -- into the basis of the backward pass, xt = T x, T = [[I, G], [0, I]] (the series STATES replaced by the
// OBSERVABLES y_l = x_l + sum_k g_lk x_{N+k}).  This kernel filters in that basis from the start: the same recursion
// (kalmanfilter.py:236-400; the tape it leaves is the same, tests/dk_ref.py) on  Pt = T P T',  where the observation row of
// series j is the unit vector e_j:
//     innovation      v = y_j - xt_j                        (no loadings)
//     d = Pt e_j      d_l = Pt[l][j],  d_{N+k} = Pt[j][N+k]  (a column pick and lane j's own factor columns: no arithmetic)
//     f = R_j + d_j
//     gain            kt = d / f IS the tape's vector;  an unobserved series' entry is COLUMN u of Pt, its mean xt_u, its
//                     variance Pt[u][u]: registers, not products
// -- 32 of the ~100 f64 instructions of a pass and the whole unobserved-entry pass (16 multiply-adds per entry) go.  The price
// is the prediction, where the transition is no longer diagonal:  Pht = T Phi T^-1 = [[Phi_s, C], [0, Phi_f]],
// C[l][k] = g_lk (phi_{N+k} - phi_l),  Qt = T Q T' = [[Q_s + G Q_f G', G Q_f], [Q_f G', Q_f]]:
//     xt_l   <- phi_l xt_l + sum_k C[l][k] xf_k
//     W[k][c] = phi_c Pt[c][N+k] + sum_k' PF[k][k'] C[c][k']        (lane c, published through LDS)
//     Pt[l][N+k] <- phi_{N+k} W[k][l] + g_lk q_{N+k}
//     Pt[l][c]   <- phi_c (phi_l Pt[l][c] - sum_k g_ck A[k]) + sum_k g_ck E[k] + sum_k C[l][k] W[k][c] + [l = c] q_l,
//                   A[k] = phi_l Pt[l][N+k],  E[k] = A[k] phi_{N+k} + g_lk q_{N+k}
// 2 + 3K instructions per covariance element and step where the diagonal transition has one -- ~480 per step against the
// ~800 + ~280 it saves at configs[3] (24 passes of two models per wavefront-step; 16 unobserved-entry bodies).  Same layout as
// filter_split_kernel: series l of a model on lane l of its H-lane group, the K x K factor block and the factor means
// replicated; same bookkeeping (get_mle's compressed indices, :380-382, :550-567); objective and tape only (the records of the
// other modes are in the state basis: they keep the kernel above).
// =====================================================================================
template <int N, int K, int H, bool BOOK>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) filter_obs_kernel(FilterArgs a)
{
    constexpr int n = N + K, M = 64 / H;
    static_assert(N <= H && (H == 16 || H == 32) && n > 16, "split layout: N series on H lanes, wide models");
    constexpr int NP = n + (n & 1);
    constexpr int KF = K * (K + 1) / 2;
    constexpr int KP = (K + 1) & ~1;             // a row of K doubles, padded to 16-byte pieces
    constexpr int GP = (K + 2) & ~1;             // a row [ g_c0 .. g_c,K-1 | phi_c ] of the loadings table
    constexpr int SW = tape_side_c(K);
    constexpr int XS = tape_xs_c(N, K), SS = tape_ss_c(N, K), SO = tape_so_c(N, K);
    constexpr bool PAIRS = (K % 2 == 0 && N % 2 == 0);
    constexpr unsigned NM = N >= 32 ? 0xffffffffu : ((1u << (N & 31)) - 1u);
    constexpr int TS = 16;
    constexpr unsigned HM = H == 32 ? 0xffffffffu : 0xffffu;
    auto pf = [](int k, int k2) constexpr { return k * K - k * (k - 1) / 2 + (k2 - k); }; // k <= k2

    const int lane = threadIdx.x, h = lane / H, l = lane % H;
    const int jr = l < N ? l : N - 1;
    long inst = (long)blockIdx.x * M + h;
    if (inst > a.B - 1) inst = a.B - 1;
    const long rec = inst % a.R;
    const long T = a.T;

    constexpr int DVS = NP + 2;                  // a d buffer: d [n] (+ pad), then the innovation v and its variance f of lane j
    // DIST: the K x K factor block and the factor means, replicated in every lane by the other kernels, live here ONE ELEMENT PER
    // LANE between the predictions (lane e < KF: PF element e; lane KF + k: factor mean k): a scalar update is then two
    // instructions on that lane's element (with its two operands read from d's buffer at the lane's own addresses) instead of
    // KF + 2K replicated ones, and the step's prediction reads the elements back through FS doubles of LDS per model.
    constexpr bool DIST = (KF + K <= H) && (K % 2 == 0) && !MK_TUNE_SKIP(a, 512);
    constexpr int FS = DIST ? H : 0;
    __shared__ __attribute__((aligned(16))) double lds[M * (2 * DVS + N * GP + N * KP + 2 * KP + TS * N + N * 2 * KP + FS)];
    double *dbuf = lds + h * 2 * DVS;                                        // d = Pt e_j, two buffers
    double *gtab = lds + M * 2 * DVS + h * N * GP;                           // [N][GP]: loadings of series c, then phi_c
    double *wtab = lds + M * (2 * DVS + N * GP) + h * N * KP;                // [N][KP]: W[.][c] of the step
    double *ftab = lds + M * (2 * DVS + N * GP + N * KP) + h * 2 * KP;       // phi and q of the factor states
    double *otile = lds + M * (2 * DVS + N * GP + N * KP + 2 * KP) + h * TS * N;
    // the lane's own prediction constants C[l][.] and (G Q_f)[l][.] wait in LDS between steps (registers are what this kernel is
    // short of: the row of Pt, the factor block and the update's pieces of d fill a two-wavefront budget)
    double *ltab = lds + M * (2 * DVS + N * GP + N * KP + 2 * KP + TS * N) + h * N * 2 * KP + jr * 2 * KP;
    [[maybe_unused]] double *fbuf = lds + M * (2 * DVS + N * GP + N * KP + 2 * KP + TS * N + N * 2 * KP) + h * FS;

    auto load_k = [](const double *src, double(&dst)[K]) __attribute__((always_inline)) { // K doubles from a 16-byte aligned row
        if constexpr (K % 2 == 0) {
#pragma unroll
            for (int k = 0; k < K; k += 2) {
                const v2d t2 = *reinterpret_cast<const v2d *>(src + k);
                dst[k] = t2.x;
                dst[k + 1] = t2.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k + 1 < K; k += 2) {
                const v2d t2 = *reinterpret_cast<const v2d *>(src + k);
                dst[k] = t2.x;
                dst[k + 1] = t2.y;
            }
            dst[K - 1] = src[K - 1];
        }
    };

    const double phi_l = a.phi[inst * n + jr];
    const double q_l = a.q[inst * n + jr];
    double gam[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        gam[k] = a.loadings[(rec * N + jr) * K + k];
        const double pfk = a.phi[inst * n + N + k];
        const double qf = a.q[inst * n + N + k];
        ltab[k] = gam[k] * (pfk - phi_l);        // C[l][k]
        ltab[KP + k] = gam[k] * qf;              // (G Q_f)[l][k]
        gtab[jr * GP + k] = gam[k];
        ftab[k] = pfk;                           // every lane of the group writes the same values
        ftab[KP + k] = qf;
    }
    gtab[jr * GP + K] = phi_l;
    const double rvar = a.obsvar ? a.obsvar[rec * N + jr] : 0.0;
    wave_lds_sync();

    // initial moments (run_filter defaults, kalmanfilter.py:747-750, or the caller's) into the observable basis:
    // xt = T x0, Pt = T P0 T':  Pt[l][N+k] = P0[l][N+k] + sum_k' g_lk' P0[N+k'][N+k],
    // Pt[l][c] = P0[l][c] + sum_k g_ck P0[l][N+k] + sum_k g_lk Pt[c][N+k]  (the last sum uses lane c's factor columns: through wtab)
    double xo, xk[K], PF[KF];
    LaneRow<N, K> P;
#pragma unroll
    for (int k = 0; k < K; ++k) xk[k] = a.x0 ? a.x0[inst * n + N + k] : 0.0;
    xo = a.x0 ? a.x0[inst * n + jr] : 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) xo = fma(gam[k], xk[k], xo);
    if (a.P0) { // a caller's initial covariance (two separate paths: one loop with a test per element kept 160 loads in flight and spilled)
        const double *p0 = a.P0 + (inst * n + jr) * n;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int k2 = k; k2 < K; ++k2) PF[pf(k, k2)] = a.P0[(inst * n + N + k) * n + N + k2];
        double p0f[K];
#pragma unroll
        for (int k = 0; k < K; ++k) p0f[k] = p0[N + k];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double v = p0f[k];
#pragma unroll
            for (int k2 = 0; k2 < K; ++k2) v = fma(gam[k2], PF[k <= k2 ? pf(k, k2) : pf(k2, k)], v);
            P.set(N + k, v);
            wtab[jr * KP + k] = v;
        }
        wave_lds_sync();
        sfor<0, N>(MK_LAMBDA(cc) {
            constexpr int c = decltype(cc)::value;
            double gc[K], wc[K];
            load_k(gtab + c * GP, gc);
            load_k(wtab + c * KP, wc);
            double v = p0[c];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                v = fma(gc[k], p0f[k], v);
                v = fma(gam[k], wc[k], v);
            }
            P.set(c, v);
            __builtin_amdgcn_sched_barrier(0);
        });
    } else { // P0 = I:  Pt = [[I + G G', G], [G', I]]
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int k2 = k; k2 < K; ++k2) PF[pf(k, k2)] = k == k2 ? 1.0 : 0.0;
        int jv0 = jr; // opaque: the identity is built here, not hoisted as n selects
        asm volatile("" : "+v"(jv0));
        sfor<0, N>(MK_LAMBDA(cc) {
            constexpr int c = decltype(cc)::value;
            double gc[K];
            load_k(gtab + c * GP, gc);
            double v = c == jv0 ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < K; ++k) v = fma(gam[k], gc[k], v);
            P.set(c, v);
            if constexpr (c % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int k = 0; k < K; ++k) P.set(N + k, gam[k]);
    }
    wave_lds_sync(); // wtab is rewritten by the first prediction

    // DIST: this lane's element (facc), where its two operands sit in a d buffer (a: dfa[k] / -v, b: dfa[k2] / dfa[k]), and the
    // constants of its transition  facc <- (facc pa) pb + qd  (the replicated form's operations, element by element)
    [[maybe_unused]] double facc = 0.0, fpa = 0.0, fpb = 0.0, fqd = 0.0;
    [[maybe_unused]] int fao = 0, fbo = 0;
    if constexpr (DIST) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double pk = a.phi[inst * n + N + k], qk = a.q[inst * n + N + k];
#pragma unroll
            for (int k2 = k; k2 < K; ++k2) {
                const double pk2 = a.phi[inst * n + N + k2];
                if (l == pf(k, k2)) {
                    facc = PF[pf(k, k2)];
                    fao = N + k;
                    fbo = N + k2;
                    fpa = pk;
                    fpb = pk2;
                    fqd = k == k2 ? qk : 0.0;
                }
            }
            if (l == KF + k) {
                facc = xk[k];
                fao = 0;                         // -v rides in d's slot 0
                fbo = N + k;
                fpa = pk;
                fpb = 1.0;
                fqd = -0.0;                      // (x pa) 1 + (-0) = x pa, signed zeros included
            }
        }
        fbuf[l < FS ? l : 0] = facc;
        wave_lds_sync();
    }

    const double *obase = a.obs + rec * a.obs_bs * N + jr;
    const long ostep = a.obs_ts * N;
    double *trec = a.F ? a.F + inst * a.bs * a.rs : nullptr; // tape block of (model, step); nullptr: objective only
    const long tstep = a.ts * a.rs;

    double sum_sig = 0.0, sum_det = 0.0;
    double run_mant = 1.0;
    long run_exp = 0;
    long nobs = 0, sc = 0;
    double fmin_seen = 1.0;

    for (long t0 = 0; t0 < T; t0 += TS) {
        wave_lds_sync();
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            long tr = t0 + s;
            if (tr > T - 1) tr = T - 1;
            otile[s * N + jr] = obase[tr * ostep];
        }
        wave_lds_sync();
        const long tend = t0 + TS < T ? t0 + TS : T;
        for (long t = t0; t < tend; ++t) {
            const double y = otile[(int)(t - t0) * N + jr];
            const unsigned long long ball = __ballot(l < N && isfinite(y));
            unsigned mrem[M];
#pragma unroll
            for (int g = 0; g < M; ++g) mrem[g] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(ball >> (g * H)) & HM));
            unsigned maskl = mrem[0];
#pragma unroll
            for (int g = 1; g < M; ++g) maskl = (h == g) ? mrem[g] : maskl;
            const int cnt = __popc(maskl);

            // ---- predict in the observable basis (header) ----
            {
                double A[K], E[K], W[K], Cl[K], Bq[K], phif[K];
                if constexpr (DIST) { // the filtered factor block and means of the previous step, every lane a copy for this block only
                    double fv[KF + K];
#pragma unroll
                    for (int e = 0; e < KF + K; e += 2) {
                        const v2d t2 = *reinterpret_cast<const v2d *>(fbuf + e);
                        fv[e] = t2.x;
                        fv[e + 1] = t2.y;
                    }
#pragma unroll
                    for (int e = 0; e < KF; ++e) PF[e] = fv[e];
#pragma unroll
                    for (int k = 0; k < K; ++k) xk[k] = fv[KF + k];
                    facc = fma(facc * fpa, fpb, fqd);
                }
                load_k(ltab, Cl);
                load_k(ltab + KP, Bq);
                load_k(ftab, phif);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    A[k] = phi_l * P.get(N + k);
                    double w = A[k];
#pragma unroll
                    for (int k2 = 0; k2 < K; ++k2) w = fma(PF[k <= k2 ? pf(k, k2) : pf(k2, k)], Cl[k2], w);
                    W[k] = w;
                    E[k] = fma(A[k], phif[k], Bq[k]);
                }
                if constexpr (K % 2 == 0) {
#pragma unroll
                    for (int k = 0; k < K; k += 2) *reinterpret_cast<v2d *>(wtab + jr * KP + k) = v2d{W[k], W[k + 1]};
                } else {
#pragma unroll
                    for (int k = 0; k < K; ++k) wtab[jr * KP + k] = W[k];
                }
                // means: xt_l <- phi_l xt_l + C[l] . xf (old factor means)
                xo = phi_l * xo;
#pragma unroll
                for (int k = 0; k < K; ++k) xo = fma(Cl[k], xk[k], xo);
                if constexpr (!DIST) {
#pragma unroll
                    for (int k = 0; k < K; ++k) xk[k] = phif[k] * xk[k];
                }
                wave_lds_sync();
                int jv = jr; // opaque copies: keeps the selects inside the loop
                double qv = q_l;
                asm volatile("" : "+v"(jv), "+v"(qv));
                sfor<0, N>(MK_LAMBDA(cc) {
                    constexpr int c = decltype(cc)::value;
                    double gc[K], wc[K];
                    load_k(gtab + c * GP, gc);
                    load_k(wtab + c * KP, wc);
                    const double phc = gtab[c * GP + K];
                    double u = P.get(c) * phi_l;
#pragma unroll
                    for (int k = 0; k < K; ++k) u = fma(-gc[k], A[k], u);
                    double sacc = fma(u, phc, c == jv ? qv : 0.0);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        sacc = fma(gc[k], E[k], sacc);
                        sacc = fma(Cl[k], wc[k], sacc);
                    }
                    P.set(c, sacc);
                    if constexpr (c % 2 == 1) __builtin_amdgcn_sched_barrier(0); // bounds the table rows in flight (2 x (2K + 1) doubles)
                });
#pragma unroll
                for (int k = 0; k < K; ++k) P.set(N + k, fma(phif[k], W[k], Bq[k]));
                if constexpr (!DIST) {
                    double qf[K];
                    load_k(ftab + KP, qf);
#pragma unroll
                    for (int k = 0; k < K; ++k)
#pragma unroll
                        for (int k2 = k; k2 < K; ++k2) PF[pf(k, k2)] = fma(PF[pf(k, k2)] * phif[k], phif[k2], k == k2 ? qf[k] : 0.0);
                }
            }

            // ---- sequential scalar updates (:341-378), ascending series order, one observation of every model per pass ----
            double sigma = 0.0, fmant = 1.0;
            int fexp = 0, nupd = 0;
            unsigned many = 0;
            double skf[K], svr = 0.0, srf = 0.0;   // this lane's side row of the step's tape block (see the end of the pass)
#pragma unroll
            for (int k = 0; k < K; ++k) skf[k] = 0.0;
            unsigned urem[M];
#pragma unroll
            for (int g = 0; g < M; ++g) urem[g] = ~mrem[g] & NM;
#pragma unroll
            for (int g = 0; g < M; ++g) many |= mrem[g];
            while (many) {
                // The pass's bookkeeping on the SCALAR unit, spelled out (left to the compiler the loop-carried masks move to vector
                // registers: bit scans, pops, selects and lane compares were a quarter of the pass's 32-bit vector instructions, and
                // a wave64 instruction of any width holds the SIMD for four cycles): next set bit of every model's mask (0 and
                // "invalid" for a model that has none left), the lanes of the valid models, the models' lanes j.
                int jsel[M];
                unsigned long long okm = 0, jm = 0;
#pragma unroll
                for (int g = 0; g < M; ++g) {
                    unsigned vs, tt;
                    asm volatile("s_ff1_i32_b32 %[j], %[m]\n\t"
                                 "s_cmp_lg_u32 %[m], 0\n\t"
                                 "s_cselect_b32 %[j], %[j], 0\n\t"
                                 "s_cselect_b32 %[v], -1, 0\n\t"
                                 "s_add_i32 %[t], %[m], -1\n\t"
                                 "s_and_b32 %[m], %[m], %[t]"
                                 : [j] "=&s"(jsel[g]), [v] "=&s"(vs), [t] "=&s"(tt), [m] "+s"(mrem[g])
                                 :
                                 : "scc");
                    okm |= (unsigned long long)(vs & HM) << (g * H);
                    jm |= (unsigned long long)(vs & (1u << jsel[g])) << (g * H);
                }
                many = 0;
#pragma unroll
                for (int g = 0; g < M; ++g) many |= mrem[g];
                double *dv = dbuf + (nupd & 1) * DVS; // two buffers: one fence per update
                ++nupd;
                // a model with no observation left sits the pass out: exec = the lanes of the valid models, straight from the
                // scalar mask (until the end of round 6 its lanes ran the body as a no-op through six selects on v, f and 1/f)
                if (__builtin_amdgcn_inverse_ballot_w64(okm)) {
                const bool isj = __builtin_amdgcn_inverse_ballot_w64(jm); // l == j of the lane's model
                int jl = jsel[0];                  // per lane: only the tape's address needs it
#pragma unroll
                for (int g = 1; g < M; ++g) jl = (h == g) ? jsel[g] : jl;
                const double vl = y - xo;          // innovation of THIS lane's series; lane j's is the model's (:344-347)
                double dr = 0.0;                   // d_l = Pt[l][j]: column j of the own row (:349-357 with Z_j = e_j)
                P.template pick<H>(dr, jsel);      // (writes the lanes of every model; those sitting out do not read it)
                double dlo, dhi;                   // d in every row of the model: the operands of the rank-one update
                LaneRow<N, K>::template split_rows<H>(dr, dlo, dhi);
                if (isj) { // lane j: the factor part of d is its own factor columns; v and f = R_j + d_j (:359-362) ride along
                    if constexpr (PAIRS) {
#pragma unroll
                        for (int k = 0; k < K; k += 2) *reinterpret_cast<v2d *>(dv + N + k) = v2d{P.get(N + k), P.get(N + k + 1)};
                    } else {
#pragma unroll
                        for (int k = 0; k < K; ++k) dv[N + k] = P.get(N + k);
                    }
                    *reinterpret_cast<v2d *>(dv + NP) = v2d{vl, rvar + dr};
                    if constexpr (DIST) *reinterpret_cast<v2d *>(dv) = v2d{-vl, 0.0};
                }
                wave_lds_sync();
                [[maybe_unused]] double fa = 0.0, fb = 0.0;
                if constexpr (DIST) {
                    fa = dv[fao];
                    fb = dv[fbo];
                }
                const v2d vf = *reinterpret_cast<const v2d *>(dv + NP);
                const double v = MK_TUNE_SKIP(a, 2048) ? vl : vf.x, f = MK_TUNE_SKIP(a, 2048) ? rvar + dr : vf.y; // (timing builds)
                const double rf = MK_TUNE_SKIP(a, 1024) ? __builtin_amdgcn_rcp(f) : rcp_nr(f);
                const double kr = dr * rf;         // gain element l (:364-366) = the tape's vector
                {
                    double kf[K];
                    double dfk[K];
                    {
                        double dfa[K];
                        if constexpr (PAIRS) {
#pragma unroll
                            for (int k = 0; k < K; k += 2) {
                                const v2d t2 = *reinterpret_cast<const v2d *>(dv + N + k);
                                dfa[k] = t2.x;
                                dfa[k + 1] = t2.y;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < K; ++k) dfa[k] = dv[N + k];
                        }
#pragma unroll
                        for (int k = 0; k < K; ++k) kf[k] = dfa[k] * rf; // (DIST: only the tape's writer, lane j, still needs them)
#pragma unroll
                        for (int k = 0; k < K; ++k) dfk[k] = dfa[k];
                        if constexpr (DIST) {
                            facc = fma(-fa, fb * rf, facc); // PF[k][k2] -= dfa[k] kf[k2]  |  xf[k] += v kf[k]: the same two roundings
                        } else {
#pragma unroll
                            for (int k = 0; k < K; ++k) {
#pragma unroll
                                for (int k2 = k; k2 < K; ++k2) PF[pf(k, k2)] = fma(-dfa[k], kf[k2], PF[pf(k, k2)]);
                                xk[k] = fma(kf[k], v, xk[k]);
                            }
                        }
                    }
                    if (trec && !MK_TUNE_SKIP(a, 1)) { // tape entry of the observed series jl
                        trec[jl * XS + jr] = kr;
                        // the entry's side row [ kf | v/f | 1/f | y | 0 ] is lane j's own: it KEEPS it (a series is observed once a
                        // step) and every lane stores its series' side row once, at the end of the step -- stored here, under
                        // exec = lane j, the K/2 + 2 sixteen-byte stores of one or two lanes per pass were 5.6 of the launch's 39.7 ms
                        // at configs[3] (timing builds, profiles/r06: a store instruction costs its trip through the address unit
                        // whatever its exec mask)
                        if (isj) {
#pragma unroll
                            for (int k = 0; k < K; ++k) skf[k] = kf[k];
                            svr = v * rf;
                            srf = rf;
                        }
                    }
                    // Pt -= k k' f (:368-372): the series columns by broadcast multiply-adds inside the rows (LaneRow::rank_one -- until the
                    // end of round 6 every lane read d back from LDS, 18 sixteen-byte reads a pass), the factor columns from lane j's entries
                    P.rank_one(dlo, dhi, -kr);
#pragma unroll
                    for (int k = 0; k < K; ++k) P.set(N + k, fma(-dfk[k], kr, P.get(N + k)));
                }
                xo = fma(kr, v, xo);               // :374-375
                sigma = fma(v * v, rf, sigma);     // :377
                fmant *= f;                        // detf += log f (:378) as mantissa * 2^exp, renormalised every pass (three
                fexp += __builtin_amdgcn_frexp_exp(fmant);   // instructions; every fourth pass under selects was six)
                fmant = __builtin_amdgcn_frexp_mant(fmant);
                fmin_seen = min_f64(fmin_seen, f);
                } // lanes of the valid models
            }
            fexp += __builtin_amdgcn_frexp_exp(fmant);
            fmant = __builtin_amdgcn_frexp_mant(fmant);
            if constexpr (DIST) { // the filtered factor block and means: where the state tape's writer and the next prediction read them
                fbuf[l < FS ? l : 0] = facc;
                wave_lds_sync();
            }

            if (cnt > 0) { // :380-382 compressed bookkeeping (per model: lanes of a group agree)
                if constexpr (BOOK) {
                    int le;
                    const double lm = log_mant(fmant, le);
                    const double detf = fma((double)(fexp + le), kLn2, lm);
                    if (a.sigmas && l == 0) a.sigmas[(inst * a.bs + sc * a.ts) * a.sig_stride] = sigma;
                    if (a.detfs && l == 0) a.detfs[(inst * a.bs + sc * a.ts) * a.sig_stride] = detf;
                    if (sc >= a.warmup) { // get_mle: COMPRESSED indices (:563-564)
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

            if (trec && MK_TUNE_SKIP(a, 2)) trec += tstep;
            if (trec && !MK_TUNE_SKIP(a, 2)) {
                // entries of the series NOT observed at this step: column u of the filtered Pt, the filtered observable and its
                // variance -- all of them registers (static loop: the column is a named register; a wavefront-uniform branch skips
                // the series every model of the wavefront observed)
                unsigned uany = 0;
#pragma unroll
                for (int g = 0; g < M; ++g) uany |= urem[g];
                const double qnan = __builtin_nan("");
                int lv = l; // opaque copy: the 32 comparisons l == c below stay inside the loop (hoisted, they are 32 SGPR pairs and spills)
                asm volatile("" : "+v"(lv));
                if (a.tape == 2) { // STATE tape: entry N + k = [ Pt[.][N+k] | PF[.][k] | xf_k | PF[k][k] | NaN | 0 ]
                    if constexpr (DIST) { // the filtered factor block and means: from the exchange buffer
#pragma unroll
                        for (int e = 0; e < KF + K; e += 2) {
                            const v2d t2 = *reinterpret_cast<const v2d *>(fbuf + e);
                            if (e < KF) PF[e] = t2.x; else xk[e - KF] = t2.x;
                            if (e + 1 < KF) PF[e + 1] = t2.y; else xk[e + 1 - KF] = t2.y;
                        }
                    }
                    sfor<0, K>(MK_LAMBDA(kk) {
                        constexpr int k = decltype(kk)::value;
                        trec[(N + k) * XS + jr] = P.get(N + k);
                        if (l == 0) {
                            double *sd = trec + SO + (N + k) * SS;
                            if constexpr (PAIRS) {
#pragma unroll
                                for (int k2 = 0; k2 < K; k2 += 2)
                                    *reinterpret_cast<v2d *>(sd + k2) = v2d{PF[k2 <= k ? pf(k2, k) : pf(k, k2)], PF[k2 + 1 <= k ? pf(k2 + 1, k) : pf(k, k2 + 1)]};
                                *reinterpret_cast<v2d *>(sd + SW - 4) = v2d{xk[k], PF[pf(k, k)]};
                                *reinterpret_cast<v2d *>(sd + SW - 2) = v2d{qnan, 0.0};
                            } else {
#pragma unroll
                                for (int k2 = 0; k2 < K; ++k2) sd[k2] = PF[k2 <= k ? pf(k2, k) : pf(k, k2)];
                                sd[SW - 4] = xk[k];
                                sd[SW - 3] = PF[pf(k, k)];
                                sd[SW - 2] = qnan;
                                sd[SW - 1] = 0.0;
                            }
                        }
                    });
                }
                if (uany) {
                    unsigned uml = urem[0];
#pragma unroll
                    for (int g = 1; g < M; ++g) uml = (h == g) ? urem[g] : uml;
                    sfor<0, N>(MK_LAMBDA(cc) {
                        constexpr int c = decltype(cc)::value;
                        if (uany & (1u << c)) {
                            if ((uml >> c) & 1u) {
                                trec[c * XS + jr] = P.get(c);
                                if (lv == c) srf = P.get(c);   // the filtered variance of the unobserved series c, for its side row below
                            }
                        }
                    });
                }
                // side rows, every series' by its own lane: [ kf | v/f | 1/f | y | 0 ] kept from its pass if it was observed,
                // [ Pt[l][N..] | xo | Pt[l][l] | NaN | 0 ] otherwise
                if (!MK_TUNE_SKIP(a, 4096)) {
                    double sy = y;
                    if (!((maskl >> l) & 1u)) {
#pragma unroll
                        for (int k = 0; k < K; ++k) skf[k] = P.get(N + k);
                        svr = xo;
                        sy = qnan;
                    }
                    if (l < N) {
                        double *sd = trec + SO + l * SS;
                        if constexpr (PAIRS) {
#pragma unroll
                            for (int k = 0; k < K; k += 2) *reinterpret_cast<v2d *>(sd + k) = v2d{skf[k], skf[k + 1]};
                            *reinterpret_cast<v2d *>(sd + SW - 4) = v2d{svr, srf};
                            *reinterpret_cast<v2d *>(sd + SW - 2) = v2d{sy, 0.0};
                        } else {
#pragma unroll
                            for (int k = 0; k < K; ++k) sd[k] = skf[k];
                            sd[SW - 4] = svr;
                            sd[SW - 3] = srf;
                            sd[SW - 2] = sy;
                            sd[SW - 1] = 0.0;
                        }
                    }
                }
                trec += tstep;
            }
        }
    }

    if (BOOK) { // zero tail of the compressed arrays (np.zeros init, :307-308)
        for (long i = sc + l; i < T; i += H) {
            if (a.sigmas) a.sigmas[(inst * a.bs + i * a.ts) * a.sig_stride] = 0.0;
            if (a.detfs) a.detfs[(inst * a.bs + i * a.ts) * a.sig_stride] = 0.0;
        }
    }
    if (l == 0) {
        if (!BOOK) sum_det = fma((double)run_exp, kLn2, log(run_mant));
        if (a.mle) a.mle[inst] = ((double)nobs * kLog2Pi + sum_det) + sum_sig; // :566
        if (a.sigmacount) a.sigmacount[inst] = sc;
        if (a.status) a.status[inst] = (fmin_seen > 0.0) ? 0u : MK_FLAG_NONPOSITIVE_F; // NaN f also flags
    }
}

// Which calls the split kernel serves: wide models (n > 16) with N <= 32, objective-only or full-square record outputs.
template <int N, int K>
static hipError_t launch_split_nk(const FilterArgs &a, hipStream_t s)
{
    if constexpr (N + K > 16 && N <= 32) {
        constexpr int H = N <= 16 ? 16 : 32, M = 64 / H;
        const unsigned grid = (unsigned)((a.B + M - 1) / M);
        const bool book = a.sigmas || a.detfs;
        const bool any = a.F || a.Pf || a.Xp || a.Pp;
        if (a.sym) { // packed-symmetric records (round 4; before: filter_kernel<N,K,64>)
            if (a.rs > 0 && a.Xp) hipLaunchKernelGGL((filter_split_kernel<N, K, H, 1, true, true>), dim3(grid), dim3(64), 0, s, a);
            else if (a.rs > 0 && any) hipLaunchKernelGGL((filter_split_kernel<N, K, H, 3, true, true>), dim3(grid), dim3(64), 0, s, a);
            else return hipErrorNotSupported;
            return hipGetLastError();
        }
        if (a.tape) { // the backward tape of mk_dk.hip in a.F (a.rs = N (n + 4) doubles per model-step)
            if (a.tape_basis == 1) { // the round-4 writer (state basis), kept as the tested A/B partner: MK_VARIANT_TAPE_FILTER
                if (book) hipLaunchKernelGGL((filter_split_kernel<N, K, H, 4, true>), dim3(grid), dim3(64), 0, s, a);
                else hipLaunchKernelGGL((filter_split_kernel<N, K, H, 4, false>), dim3(grid), dim3(64), 0, s, a);
            } else { // round 6: the filter in the observable basis
                if (book) hipLaunchKernelGGL((filter_obs_kernel<N, K, H, true>), dim3(grid), dim3(64), 0, s, a);
                else hipLaunchKernelGGL((filter_obs_kernel<N, K, H, false>), dim3(grid), dim3(64), 0, s, a);
            }
            return hipGetLastError();
        }
        if (!any && !book) hipLaunchKernelGGL((filter_split_kernel<N, K, H, 0, false>), dim3(grid), dim3(64), 0, s, a);
        else if (!any) hipLaunchKernelGGL((filter_split_kernel<N, K, H, 0, true>), dim3(grid), dim3(64), 0, s, a);
        else if (a.rs > 0 && a.Xp) hipLaunchKernelGGL((filter_split_kernel<N, K, H, 1, true>), dim3(grid), dim3(64), 0, s, a);
        else if (a.rs > 0) hipLaunchKernelGGL((filter_split_kernel<N, K, H, 3, true>), dim3(grid), dim3(64), 0, s, a);
        else return hipErrorNotSupported; // dense outputs: filter_kernel<N,K,64>
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}
#define MK_CASE_SPLIT(NN, KK) \
    if (N == NN && K == KK) return launch_split_nk<NN, KK>(a, s);
hipError_t launch_filter_split(int N, int K, const FilterArgs &a, hipStream_t s)
{
    MK_SHAPES(MK_CASE_SPLIT)
    return hipErrorNotSupported;
}

} // namespace mk

namespace mk {

// =====================================================================================
// Reverse-mode (adjoint) gradient of -2 log L for WIDE models (16 < n <= 64), one model per wavefront, lane r = state r.
//   The same backward walk as adjoint_kernel (mk_kernels.hip; formulas there and in tests/adjoint_ref.py): per step the
//   filtered record of step t-1 is re-read, the prediction and the scalar updates of step t are recomputed keeping
//   (d, 1/f, v) of every update -- here in a wave-private LDS table, whose rows also serve as the broadcast source of the
//   rank-one update -- and the adjoints (xb, Pb) are pulled back through them, last observation first.  What the 16-lane
//   kernel does with fused DPP broadcasts is done with group-uniform LDS reads (b = Pb d, Pb += z db^T / 2), two DPP tree
//   sums per update (a = xb.d, c = d.b) and jump tables for the run-time column j (P[r][j], Pb[r][j] += db_r / 2).
//   Registers: the rows of Pb and of P (which doubles as the prefetch buffer of the next record); the filtered row of
//   step t-1, needed again by the prediction adjoint, waits in LDS.  The reference has no counterpart: scipy differences
//   n + 1 filter runs (metran/solver.py:248-255) -- 37 of them at configs[3]'s shape.
// =====================================================================================
// UPD (round 6): the (d, 1/f, v) of every update are not recomputed but READ from the update tape the recording forward pass wrote
// (FilterArgs.upd: filter_kernel<N,K,64,OUT=3>, one block of N slots per (model, step)) -- the forward half of a step was 38 % of this
// kernel's instructions, and the walk is one wavefront's dependent chain.  The step's block is copied HBM -> LDS by the wavefront
// itself (global_load_lds, asynchronous, no registers), issued when the previous block's last slot has been consumed.
typedef __attribute__((address_space(3))) void adj_lds_void_t;
typedef __attribute__((address_space(3))) char adj_lds_char_t;
typedef __attribute__((address_space(1))) const void adj_global_cvoid_t;

template <int N, int K, bool UPD>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) adjoint_wide_kernel(AdjointArgs a)
{
    constexpr int n = N + K;
    static_assert(n > 16 && n <= 64, "one model per wavefront, 16 < n <= 64");
    using Gp = Group<64>;
    constexpr int NP = n + (n & 1);
    constexpr int DS = NP + 2;                  // one update: d [n] (+ pad), 1/f, v
    static_assert(DS == adjoint_update_slot_c(N, K), "slot layout of the update tape");
    constexpr int GT = (N * K + 1) & ~1;
    constexpr int DB = 12, NBT = (n + DB - 1) / DB; // uniform LDS vectors are read in pieces of DB doubles
    constexpr int UB = N * DS * 8, CHUNK = 64 * 16, NCH = (UB + CHUNK - 1) / CHUNK; // the block copy: 16 bytes a lane and instruction
    constexpr int DSB = UPD ? NCH * CHUNK / 8 : N * DS;                             // (+ what its last chunk over-reads)
    const int lane = threadIdx.x;
    long inst = (long)blockIdx.x;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1;
    const long rec = inst % a.R;
    const int r = lane < n ? lane : n - 1;      // lanes >= n replicate lane n-1 (and contribute nothing to the sums)
    const double lmask = lane < n ? 1.0 : 0.0;
    const long T = a.T;

    __shared__ __attribute__((aligned(16))) double lds[DSB + NP + GT + n * NP + NP];
    double *dS = lds;                           // per-update table of the current step (first: its LDS address is the array's own)
    [[maybe_unused]] adj_lds_char_t *const dS3 = (adj_lds_char_t *)lds;
    double *phim = dS + DSB;                    // diag(Phi)
    double *gtab = phim + NP;                   // loadings [N][K]
    double *PL = gtab + GT;                     // filtered covariance of step t-1, row r at PL + r NP
    double *dbv = PL + n * NP;                  // db of the current update

    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    const int jr = lane < N ? lane : N - 1;
    double gam[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gam[k] = a.loadings[(rec * N + jr) * K + k];
    const double rvar = a.obsvar ? a.obsvar[rec * N + jr] : 0.0;
    phim[r] = phi_r;
#pragma unroll
    for (int k = 0; k < K; ++k) gtab[jr * K + k] = gam[k];
    wave_lds_sync();
    const long sctot = a.sigmacount[inst];      // observed steps in total (written by the forward filter)
    long rem = 0;                               // observed steps already walked (from the end)

    const long RS = a.rs;
    const double *recbase = a.F + inst * a.bs * RS;
    const long rstep = a.ts * RS;
    const double *obase = a.obs + rec * a.obs_bs * N + jr;
    const long ostep = a.obs_ts * N;

    // uniform LDS vector `src` in pieces, one piece ahead of its use: body(c, value)
    auto sweep = [&](const double *src, auto body) __attribute__((always_inline)) {
        double cur[DB], nxt[DB];
        auto fetch = [&](auto bb, double(&dst)[DB]) __attribute__((always_inline)) {
            constexpr int c0 = DB * decltype(bb)::value;
#pragma unroll
            for (int i = 0; i < DB; i += 2)
                if (c0 + i < n) {
                    const v2d t2 = *reinterpret_cast<const v2d *>(src + c0 + i);
                    dst[i] = t2.x;
                    dst[i + 1] = t2.y;
                }
        };
        fetch(std::integral_constant<int, 0>{}, cur);
        sfor<0, NBT>(MK_LAMBDA(bb) {
            constexpr int b = decltype(bb)::value, c0 = DB * b;
            if constexpr (b + 1 < NBT) fetch(std::integral_constant<int, b + 1>{}, nxt);
            __builtin_amdgcn_sched_barrier(0);
            sfor<0, DB>(MK_LAMBDA(ii) {
                constexpr int c = c0 + decltype(ii)::value;
                if constexpr (c < n) body(std::integral_constant<int, c>{}, cur[decltype(ii)::value]);
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (b + 1 < NBT) {
#pragma unroll
                for (int i = 0; i < DB; ++i) cur[i] = nxt[i];
            }
        });
    };

    double xb = 0.0, Pb[n], gphi = 0.0, gq = 0.0;
#pragma unroll
    for (int c = 0; c < n; ++c) Pb[c] = 0.0;
    double xn, P[n]; // filtered moments of step t-1 (prefetched), then the running covariance row of step t
    auto load_prev = [&](long t) __attribute__((always_inline)) {
        if (t > 0) {
            const double *p = recbase + (t - 1) * rstep;
            xn = p[r];
            load_cols<n>(p + n + r, P);
        } else { // run_filter defaults (kalmanfilter.py:747-750) or the caller's initial state
            xn = a.x0 ? a.x0[inst * n + r] : 0.0;
            int rv0 = r; // opaque: the identity row is built HERE (hoisted out of the time loop it is n spilled doubles)
            asm volatile("" : "+v"(rv0));
#pragma unroll
            for (int c = 0; c < n; ++c) P[c] = a.P0 ? a.P0[(inst * n + r) * n + c] : (c == rv0 ? 1.0 : 0.0);
        }
    };
    [[maybe_unused]] auto fetch_updates = [&](long t) __attribute__((always_inline)) { // HBM -> LDS, completion = vmcnt
        const char *g = reinterpret_cast<const char *>(a.upd + (inst * a.bs + t * a.ts) * a.us);
        sfor<0, NCH>(MK_LAMBDA(cc) {
            constexpr int c = decltype(cc)::value;
            long off = (long)c * CHUNK + (long)lane * 16;
            if constexpr ((c + 1) * CHUNK > UB) off = off < (long)UB - 16 ? off : (long)UB - 16; // stay inside the block
            __builtin_amdgcn_global_load_lds((adj_global_cvoid_t *)(g + off), (adj_lds_void_t *)(dS3 + c * CHUNK), 16, 0, 0);
        });
    };
    if constexpr (UPD) fetch_updates(T - 1);
    load_prev(T - 1);

    for (long t = T - 1; t >= 0; --t) {
        const double xprev = xn;
        const double y = obase[t * ostep];
        if constexpr (UPD) __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): this step's block of the update tape has landed in LDS
        wave_lds_sync(); // the previous step's reads of PL are complete
        store_row<n>(PL + r * NP, P);
        const unsigned long long vm = __ballot(lane < N && isfinite(y));
        if (vm != 0) {
            const double w = (sctot - rem - 1 >= a.warmup) ? 1.0 : 0.0; // compressed index of this step (:563-564)
            ++rem;
            int cnt = 0;
            if constexpr (UPD) {
                cnt = __popcll(vm);             // the slots of this step are in dS already
            } else {
            // ---- forward: prediction and scalar updates of step t, as the filter ----
            double x = phi_r * xprev;
            {
                int rv = r;
                double qv = q_r;
                asm volatile("" : "+v"(rv), "+v"(qv));
                sweep(phim, [&](auto cc, double ph) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    P[c] = fma(P[c] * phi_r, ph, c == rv ? qv : 0.0);
                });
            }
            for (unsigned long long m = vm; m; m &= m - 1) {
                const int j = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(m));
                double vl = y - x;
                sfor<0, K>(MK_LAMBDA(k) { vl = fma(-gam[decltype(k)::value], Gp::template bcast<N + decltype(k)::value>(x), vl); });
                const double v = readlane_f64(vl, j);
                double dr = 0.0;
                pick_column_all<N, n>(dr, j, P);
                const double *gp = gtab + j * K;
                sfor<0, K>(MK_LAMBDA(k) { dr = fma(P[N + decltype(k)::value], gp[decltype(k)::value], dr); });
                double *drow = dS + cnt * DS;
                drow[r] = dr;
                double fl = rvar + dr;
                sfor<0, K>(MK_LAMBDA(k) { fl = fma(Gp::template bcast<N + decltype(k)::value>(dr), gam[decltype(k)::value], fl); });
                const double f = readlane_f64(fl, j);
                const double rf = rcp_nr(f);
                const double kr = dr * rf;
                *reinterpret_cast<v2d *>(drow + NP) = v2d{rf, v}; // the same value from every lane
                wave_lds_sync();
                sweep(drow, [&](auto cc, double dcv) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    P[c] = fma(-dcv, kr, P[c]);
                });
                x = fma(kr, v, x);
                ++cnt;
            }
            } // !UPD
            // ---- reverse: adjoints back through the updates, last observation first ----
            unsigned long long mm = vm;
            for (int u = cnt - 1; u >= 0; --u) {
                const int j = __builtin_amdgcn_readfirstlane(63 - (int)__builtin_clzll(mm));
                mm &= ~(1ull << j);
                const double *drow = dS + u * DS;
                const double dr = drow[r];
                const v2d rv2 = *reinterpret_cast<const v2d *>(drow + NP);
                const double rf = rv2.x, v = rv2.y;
                double b0 = 0.0, b1 = 0.0;
                sweep(drow, [&](auto cc, double dcv) __attribute__((always_inline)) { // b_r = sum_c Pb[r][c] d_c
                    constexpr int c = decltype(cc)::value;
                    if constexpr (c % 2 == 0) b0 = fma(Pb[c], dcv, b0);
                    else b1 = fma(Pb[c], dcv, b1);
                });
                const double b = b0 + b1;
                // (round 6: the two wavefront sums on the matrix pipe -- 2 MFMA + 2 rotate-adds each, as in mk_dk.hip, instead of 8
                // permutes, 8 lane reads and 11 additions: this walk is ONE wavefront's dependent chain, 26 us per step at configs[3]'s shape)
                const double asum = wave_sum_mfma(lmask * (xb * dr)); // a = xb . d
                const double csum = wave_sum_mfma(lmask * (dr * b));  // c = d . b
                const double vrf = v * rf;
                const double vbar = fma(2.0 * w, v, asum) * rf;
                const double fbar = (fma(-w * v, vrf, w) - asum * vrf + csum * rf) * rf;
                const double *gp = gtab + j * K;
                double g[K];
                sfor<0, K>(MK_LAMBDA(k) { g[decltype(k)::value] = gp[decltype(k)::value]; });
                double zr = (r == j) ? 1.0 : 0.0; // element r of Z_j = e_j + sum_k loadings[j,k] e_{N+k}
                sfor<0, K>(MK_LAMBDA(k) { zr = (r == N + decltype(k)::value) ? g[decltype(k)::value] : zr; });
                const double dbar = fma(xb, vrf, fma(-2.0 * rf, b, fbar * zr));
                xb = fma(-vbar, zr, xb);
                const double hd = 0.5 * dbar, hz = 0.5 * zr;
                dbv[r] = dbar;
                add_column<N, n>(j, hd, Pb);                                  // Pb[r][j] += db_r / 2
                sfor<0, K>(MK_LAMBDA(k) { Pb[N + decltype(k)::value] = fma(hd, g[decltype(k)::value], Pb[N + decltype(k)::value]); });
                wave_lds_sync();
                sweep(dbv, [&](auto cc, double dbc) __attribute__((always_inline)) { // Pb[r][c] += z_r db_c / 2
                    constexpr int c = decltype(cc)::value;
                    Pb[c] = fma(hz, dbc, Pb[c]);
                });
                wave_lds_sync(); // dbv is rewritten by the next update
            }
        }
        if constexpr (UPD) {
            if (t > 0) {
                __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): no LDS read of this step's slots is still in flight
                fetch_updates(t - 1);               // the next block on its way under the prediction adjoint
            }
        }
        // next record into the (now dead) P row: in flight during the prediction adjoint; this loop stores nothing
        if (t > 0) load_prev(t - 1);
        // ---- prediction adjoint (Pprev from LDS) ----
        {
            double diag = 0.0, ts0 = 0.0, ts1 = 0.0;
            const double *prow = PL + r * NP;
            int rv = r;
            asm volatile("" : "+v"(rv));
            // two vectors in step: diag(Phi) (uniform) and the lane's own row of Pf[t-1], both in pieces
            double curA[DB], curB[DB], nxtA[DB], nxtB[DB];
            auto fetch2 = [&](auto bb, double(&da)[DB], double(&db)[DB]) __attribute__((always_inline)) {
                constexpr int c0 = DB * decltype(bb)::value;
#pragma unroll
                for (int i = 0; i < DB; i += 2)
                    if (c0 + i < n) {
                        const v2d ta = *reinterpret_cast<const v2d *>(phim + c0 + i);
                        const v2d tb = *reinterpret_cast<const v2d *>(prow + c0 + i);
                        da[i] = ta.x;
                        da[i + 1] = ta.y;
                        db[i] = tb.x;
                        db[i + 1] = tb.y;
                    }
            };
            fetch2(std::integral_constant<int, 0>{}, curA, curB);
            sfor<0, NBT>(MK_LAMBDA(bb) {
                constexpr int b = decltype(bb)::value, c0 = DB * b;
                if constexpr (b + 1 < NBT) fetch2(std::integral_constant<int, b + 1>{}, nxtA, nxtB);
                __builtin_amdgcn_sched_barrier(0);
                sfor<0, DB>(MK_LAMBDA(ii) {
                    constexpr int i = decltype(ii)::value, c = c0 + i;
                    if constexpr (c < n) {
                        diag = (c == rv) ? Pb[c] : diag;
                        const double t1 = Pb[c] * curA[i];      // Pb[r][c] phi_c
                        if constexpr (c % 2 == 0) ts0 = fma(t1, curB[i], ts0);
                        else ts1 = fma(t1, curB[i], ts1);
                        Pb[c] = t1 * phi_r;
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (b + 1 < NBT) {
#pragma unroll
                    for (int i = 0; i < DB; ++i) {
                        curA[i] = nxtA[i];
                        curB[i] = nxtB[i];
                    }
                }
            });
            gq += diag;
            gphi = fma(xb, xprev, fma(2.0, ts0 + ts1, gphi));
            xb *= phi_r;
        }
    }
    if (live && lane < n) {
        if (a.gphi) a.gphi[inst * n + lane] = gphi;
        if (a.gq) a.gq[inst * n + lane] = gq;
    }
}

template <int N, int K>
static hipError_t launch_adjoint_wide_nk(const AdjointArgs &a, hipStream_t s)
{
    if constexpr (N + K > 16) {
        if (a.upd) {
            if (a.us != adjoint_update_stride_c(N, K)) return hipErrorInvalidValue;
            hipLaunchKernelGGL((adjoint_wide_kernel<N, K, true>), dim3((unsigned)a.B), dim3(64), 0, s, a);
        } else {
            hipLaunchKernelGGL((adjoint_wide_kernel<N, K, false>), dim3((unsigned)a.B), dim3(64), 0, s, a);
        }
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}
#define MK_CASE_ADJW(NN, KK) \
    if (N == NN && K == KK) return launch_adjoint_wide_nk<NN, KK>(a, s);
hipError_t launch_adjoint_wide(int N, int K, const AdjointArgs &a, hipStream_t s)
{
    MK_SHAPES(MK_CASE_ADJW)
    return hipErrorNotSupported;
}

} // namespace mk
