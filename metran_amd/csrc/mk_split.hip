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
// Modes: OUT 0 (objective only), 1 (predicted + filtered records), 3 (filtered record); full-square records only --
// dense outputs and packed-symmetric records keep filter_kernel<N,K,64>.
#include "mk_prims.h"

namespace mk {

#define MK_SPLIT_CASES(X)                                                                                              \
    X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) \
    X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)


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

template <int N, int K, int H, int OUT, bool BOOK>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) filter_split_kernel(FilterArgs a)
{
    constexpr int n = N + K, M = 64 / H;
    static_assert(N <= H && (H == 16 || H == 32) && n > 16, "split layout: N series on H lanes, wide models");
    constexpr int NP = n + (n & 1);              // even: rows of 16-byte pieces
    constexpr int GT = (N * K + 1) & ~1;         // loadings table of one model, even
    constexpr int KF = K * (K + 1) / 2;          // upper triangle of the factor block
    constexpr int KK2 = (K * K + 1) & ~1;
    constexpr int NV = record_payload(n), RS = record_stride_c(n), PADN = RS - NV;
    constexpr bool RECF = (OUT == 1 || OUT == 3);
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
    __shared__ __attribute__((aligned(16))) double lds[M * (3 * NP + GT + KK2 + KP + TS * N)];
    double *phim = lds + h * NP;                                   // diag(Phi) [n]
    double *dbuf = lds + M * NP + h * 2 * NP;                      // d = P Z_j^T, two buffers
    double *gtab = lds + M * 3 * NP + h * GT;                      // loadings [N][K]
    double *pfs = lds + M * (3 * NP + GT) + h * KK2;               // factor block staging for the record stores
    double *qtab = lds + M * (3 * NP + GT + KK2) + h * KP;         // diag(Q) of the factor states
    double *otile = lds + M * (3 * NP + GT + KK2 + KP) + h * TS * N; // observations of TS steps

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
    double xk[K], P[n], PF[KF];
#pragma unroll
    for (int k = 0; k < K; ++k) xk[k] = a.x0 ? a.x0[inst * n + N + k] : 0.0;
#pragma unroll
    for (int c = 0; c < n; ++c) P[c] = a.P0 ? a.P0[(inst * n + jr) * n + c] : (c == jr ? 1.0 : 0.0);
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int k2 = k; k2 < K; ++k2) PF[pf(k, k2)] = a.P0 ? a.P0[(inst * n + N + k) * n + N + k2] : (k == k2 ? 1.0 : 0.0);

    const double *obase = a.obs + rec * a.obs_bs * N + jr;
    const long ostep = a.obs_ts * N;
    double *recP = (OUT == 1) ? a.Xp + inst * a.bs * RS : nullptr;
    double *recF = RECF ? a.F + inst * a.bs * RS : nullptr;
    const long rstep = a.ts * RS;
    const int kl = l < K ? l : K - 1;            // factor state whose mean this lane writes
    const int fl16 = l < K * K ? l : K * K - 1;  // factor-block element this lane writes

    // one (model, step) record: [ mean | covariance, row-major | p0, p1, zeros ]
    auto emit = [&](double *r, double xv, const double(&xf)[K], const double(&Pr)[n], const double(&PFv)[KF], double p0, double p1)
                    __attribute__((always_inline)) {
        r[jr] = xv;
        double xm = xf[0];
#pragma unroll
        for (int k = 1; k < K; ++k) xm = (kl == k) ? xf[k] : xm;
        r[N + kl] = xm;
        double *cov = r + n;
#pragma unroll
        for (int c = 0; c < n; ++c) cov[c * n + jr] = Pr[c];            // (c, l) <- P[l][c]: a contiguous run per model
        if constexpr (K % 2 == 0 && N % 2 == 0) {                       // (l, N+k) <- P[l][N+k]: K contiguous doubles
#pragma unroll
            for (int k = 0; k < K; k += 2) *reinterpret_cast<v2d *>(cov + jr * n + N + k) = v2d{Pr[N + k], Pr[N + k + 1]};
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) cov[jr * n + N + k] = Pr[N + k];
        }
        wave_lds_sync();                                                // the previous record's staging reads are done
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int k2 = 0; k2 < K; ++k2) pfs[k * K + k2] = PFv[k <= k2 ? pf(k, k2) : pf(k2, k)]; // same value from every lane
        wave_lds_sync();
        cov[(N + fl16 / K) * n + N + fl16 % K] = pfs[fl16];
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
                for (int c = 0; c < n; ++c) P[c] = fma(P[c] * phi_l, phc[c], c == jv ? qv : 0.0);
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
#pragma unroll
            for (int g = 0; g < M; ++g) many |= mrem[g];
            while (many) {
                int jsel[M];
                bool val[M];
#pragma unroll
                for (int g = 0; g < M; ++g) {
                    val[g] = mrem[g] != 0u;
                    jsel[g] = val[g] ? (int)__builtin_ctz(mrem[g]) : 0;
                    mrem[g] &= mrem[g] - 1u;
                }
                many = 0;
#pragma unroll
                for (int g = 0; g < M; ++g) many |= mrem[g];
                int jl = jsel[0];
                bool okl = val[0];
#pragma unroll
                for (int g = 1; g < M; ++g) {
                    jl = (h == g) ? jsel[g] : jl;
                    okl = (h == g) ? val[g] : okl;
                }
                // innovation (:344-347): every lane forms v_l with ITS loadings; lane j's value is the model's
                double vl = y - x, vl2 = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (k % 2 == 0) vl = fma(-gam[k], xk[k], vl);
                    else vl2 = fma(-gam[k], xk[k], vl2);
                }
                vl += vl2;
                double v = readlane_f64(vl, jsel[0]);
#pragma unroll
                for (int g = 1; g < M; ++g) {
                    const double vg = readlane_f64(vl, g * H + jsel[g]);
                    v = (h == g) ? vg : v;
                }
                // d = P Z_j^T (:349-357): column j of the own row through a wavefront-uniform switch per model
                double dr = 0.0;
                sfor<0, M>(MK_LAMBDA(gg) { // lanes of group g: dr = P[l][j_g] (jump table, exec = the group's lanes)
                    constexpr int g = decltype(gg)::value;
                    constexpr unsigned long long GM = (H == 32 ? 0xffffffffull : 0xffffull) << (g * H);
                    pick_column<N, n, GM>(dr, __builtin_amdgcn_readfirstlane(jsel[g]), P);
                });
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
                    dr = fma(P[N + k], gj[k], dr);
                    double s = P[N + k];
#pragma unroll
                    for (int k2 = 0; k2 < K; ++k2) s = fma(PF[k <= k2 ? pf(k, k2) : pf(k2, k)], gj[k2], s);
                    df[k] = s;
                }
                double *dv = dbuf + (nupd & 1) * NP; // two buffers: one fence per update
                dv[jr] = dr;
                if (l == jl) {
#pragma unroll
                    for (int k = 0; k < K; ++k) dv[N + k] = df[k];
                }
                // innovation variance f = R_j + Z_j d (:359-362), all from lane j's own values
                double fl = rvar + dr, fl2 = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (k % 2 == 0) fl = fma(gam[k], df[k], fl);
                    else fl2 = fma(gam[k], df[k], fl2);
                }
                fl += fl2;
                double f = readlane_f64(fl, jsel[0]);
#pragma unroll
                for (int g = 1; g < M; ++g) {
                    const double fg = readlane_f64(fl, g * H + jsel[g]);
                    f = (h == g) ? fg : f;
                }
                f = okl ? f : 1.0;             // a model with no observation left: the body is a no-op
                v = okl ? v : 0.0;
                double rf = rcp_nr(f);
                rf = okl ? rf : 0.0;
                const double kr = dr * rf;     // Kalman gain element l (:364-366)
                wave_lds_sync();
                {
                    // P -= k k^T f (:368-372): P[l][c] -= d_c k_l.  d is read back from LDS (group-uniform addresses) in
                    // pieces of DB doubles, one piece ahead of its multiply-adds: the whole vector at once is 2n VGPRs
                    // next to the 2n of the row, and two resident wavefronts per SIMD leave 256
                    constexpr int DB = 12, NBT = (n + DB - 1) / DB;
                    double kf[K];
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
                        for (int k = 0; k < K; ++k) {
#pragma unroll
                            for (int k2 = k; k2 < K; ++k2) PF[pf(k, k2)] = fma(-dfa[k], kf[k2], PF[pf(k, k2)]);
                            xk[k] = fma(kf[k], v, xk[k]);
                        }
                    }
                    double cur[DB], nxt[DB];
                    auto fetch = [&](auto bb, double(&dst)[DB]) __attribute__((always_inline)) {
                        constexpr int c0 = DB * decltype(bb)::value;
#pragma unroll
                        for (int i = 0; i < DB; i += 2) {
                            if (c0 + i < n) { // (dv has NP = n rounded up to even doubles: the pair read stays inside)
                                const v2d t2 = *reinterpret_cast<const v2d *>(dv + c0 + i);
                                dst[i] = t2.x;
                                dst[i + 1] = t2.y;
                            }
                        }
                    };
                    fetch(std::integral_constant<int, 0>{}, cur);
                    sfor<0, NBT>(MK_LAMBDA(bb) {
                        constexpr int b = decltype(bb)::value, c0 = DB * b;
                        if constexpr (b + 1 < NBT) fetch(std::integral_constant<int, b + 1>{}, nxt);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < DB; ++i)
                            if (c0 + i < n) P[c0 + i] = fma(-cur[i], kr, P[c0 + i]);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (b + 1 < NBT) {
#pragma unroll
                            for (int i = 0; i < DB; ++i) cur[i] = nxt[i];
                        }
                    });
                }
                x = fma(kr, v, x);             // :374-375
                sigma = fma(v * v, rf, sigma); // :377
                fmant *= f;                    // detf += log f (:378) as mantissa * 2^exp
                if ((++nupd & 3) == 0) {
                    fexp += __builtin_amdgcn_frexp_exp(fmant);
                    fmant = __builtin_amdgcn_frexp_mant(fmant);
                }
                fmin_seen = min_f64(fmin_seen, f);
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

// Which calls the split kernel serves: wide models (n > 16) with N <= 32, objective-only or full-square record outputs.
template <int N, int K>
static hipError_t launch_split_nk(const FilterArgs &a, hipStream_t s)
{
    if constexpr (N + K > 16 && N <= 32) {
        constexpr int H = N <= 16 ? 16 : 32, M = 64 / H;
        const unsigned grid = (unsigned)((a.B + M - 1) / M);
        const bool book = a.sigmas || a.detfs;
        const bool any = a.F || a.Pf || a.Xp || a.Pp;
        if (a.sym) return hipErrorNotSupported;
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
