// mk_dk.hip -- inverse-free backward pass of the wide models (n = N + K > 16): smoothed projection from the filter's TAPE.
// Reference semantics: kalmansmoother + simulate, /root/reference/metran/kalmanfilter.py:403-476, 569-603 with the scaled
// observation matrix of /root/reference/metran/metran.py:944-961 -- the same smoothed means / variances of the observables,
// computed WITHOUT the pseudo-inverse of the predicted covariance (:455) and without its five n x n products (:458-474).
//
// Formulation (round 4; Durbin & Koopman's r / N recursion for the sequential filter of :341-378).  Walk the scalar updates
// backwards with the gains the filter formed:  r <- z v/f + L'r,  N <- z z'/f + L'NL,  L = I - k z';  at any point of the walk
// the smoothed moments are  x_s = x + P r,  V = P - P N P  with the filter's (x, P) of that point;  across a step
// r <- Phi'r, N <- Phi'N Phi.  No factorisation, no pivot chain, a singular predicted covariance is a non-event.
// In Metran's own state basis z_j = e_j + sum_k g_jk e_{N+k} has 1 + K non-zeros, so every update would rewrite 1 + K rows and
// columns of N.  The kernel therefore works in the OBSERVABLE basis xt = T x, T = [[I, G], [0, I]] (the N series states
// replaced by the observables y_j): there z_j = e_j and an update rewrites ONE row / column,
//     w = N kt,  beta = kt.r,  alpha = kt.w;   r_j += v/f - beta;   N[:,j] = N[j,:] = N[:,j] - w;   N[j][j] += alpha + 1/f - w_j
// (kt = T k), at the price of a transition Pht = T Phi T^-1 = [[Phi_s, C], [0, Phi_f]], C[a][k] = g_ak (phi_{N+k} - phi_a),
// that mixes the factor columns once per STEP instead of once per observation.  What is asked for -- the projected means and
// variances of the series -- are components of xt itself: for a series u not observed at the step
//     mean_u = z_u x_f + pt.r,   var_u = z_u Pf z_u' - pt' N pt,    pt = T Pf z_u'
// with (r, N) as they stand BEFORE the step's updates; for an observed series (R = 0) the smoothed observable is the
// observation and its variance is zero.  Both vectors (kt, pt) and their scalars come from the filter: filter_split_kernel
// OUT = 4 writes one tape entry of n + 4 doubles per (step, series) (mk_split.hip; tests/dk_ref.py restates both sides).
//
// Mapping: one model per wavefront.  Lane a < n holds row a of N, lane n holds r as a ROW (the matrix-vector product then
// yields beta = r.kt in lane n for free), lanes > n replicate lane n.  The step's tape block (10 KB at configs[3]) is copied
// HBM -> LDS by the wavefront itself (global_load_lds_dwordx4: no registers, asynchronous), issued when the previous block's last
// entry has been consumed, so the copy runs under the transition.  An entry's vector is read from LDS in the DPP-replicated
// layout (lane 16q + i holds x[16m + i], m = 0 .. ceil(N/16)-1, and the K factor entries), so the product is n fused
// broadcast-multiply-adds (v_fmac_f64_dpp row_newbcast), sixteen per asm statement.  alpha is one wavefront sum on the matrix
// pipe (wave_sum_mfma, mk_jump.h).  The observed pass is UNROLLED over the series (static j: the column is a named register,
// LDS addresses are immediates).  The new column j goes to lane j's ROW through LDS (36 lanes write, lane j reads back every
// element but the diagonal, whose register is written from alpha: its reads complete under the wavefront sum and the next
// entry's product).  The unobserved entries change nothing: their products x_a w_a go back into the entry's own consumed LDS
// slots and lane u sums its entry once per step.  Transition: factor columns by n K broadcast-multiply-adds against replicated
// columns of C, series columns by two scalings, the factor rows by transposition through LDS, the K x K factor block by one
// transposed LDS reduction.  Measurements, and the variants that were measured and dropped: DESIGN.md section 4.
#include "mk_prims.h"
#include "mk_jump.h"

namespace mk {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

// HASR: the observation variances R_j (mk_problem.d_obsvar) are not all zero.  Right after its update the filter's moments of
// an observed series are  z x = y - v R/f  and  P z' = k R,  so its smoothed observable is  y - R (v/f - beta)  with variance
// R (1 - R/f) - R^2 alpha  -- the update's own beta and alpha; R = 0 (Metran: metran.py:382-384) gives (y, 0) and the
// instantiation without the four extra operations per entry.
// STATE (round 5; MK_OUT_TAPE | MK_OUT_VAR_ONLY, R = 0): the block carries K more entries, the factor columns of T Pf T'
// (state_tape_stride_c), and the kernel also writes the smoothed STATE means and variances [., n] of Metran's own basis --
// what kalmansmoother's S and diag(Ps) hold (kalmanfilter.py:461-474; consumers: metran.py:655-756) -- without the RTS chain:
//     xt_s = xt_f + Pt r,  Vt = Pt - Pt N Pt  at the END of the step (Pt = T Pf T' has zero rows / columns at the observed series),
//     x_a = xt_a - g_a . xt_F,   V_aa = Vt_aa - 2 g_a . Vt[a, F] + g_a' Vt_FF g_a,   x_{N+k} = xt_{N+k},   V_{N+k} = Vt_FF[k][k].
// Beyond the projection's work: K products w_k = N pt_{N+k} (the r row yields r . pt_{N+k}), and one product of the step's
// entries against them -- lane a reads ITS entry pt_a from the LDS block, w_k arrive DPP-replicated: c_a[k] = pt_a . w_k is
// Vt[a][N+k] = Pt[a][N+k] - c_a[k] for an unobserved series a and Vt_FF[k'][k] at lane N + k'.  tests/dk_ref.py::dk_smooth_state.
// More than 32 series (round 5; the tape then comes from the lane-per-state filter, mk_kernels.hip OUT = 4): the same kernel with
// 64-bit series masks; the row of N alone is 2 n registers, so beyond n = 40 one wavefront per SIMD owns the register file.
template <int N, int K, bool HASR, bool STATE = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(N + K <= 40 ? 2 : 1, N + K <= 40 ? 2 : 1))) smoother_dk_kernel(SmootherArgs a)
{
    static_assert(!(STATE && HASR), "the state outputs are served for R = 0 (Metran's observation variance, metran.py:382-384)");
    constexpr int n = N + K, SW = tape_side_c(K), RS = STATE ? state_tape_stride_c(N, K) : tape_stride_c(N, K);
    constexpr int XS = tape_xs_c(N, K), SS = tape_ss_c(N, K), SO = tape_so_c(N, K); // tape block addressing (mk_internal.h)
    static_assert(n > 16 && n + 1 <= 64 && K <= 16 && N >= 2, "one model per wavefront: rows 0..n-1 of N and the r row");
    constexpr int NB = (N + 15) / 16;          // DPP-replicated registers holding the series entries of a vector
    constexpr int NP = (n + 2) & ~1;           // >= n + 1, even: LDS rows of 16-byte pieces
    constexpr bool PAIRS = (N % 2 == 0 && K % 2 == 0); // side rows and their scalars are 16-byte aligned
    using mask_t = typename std::conditional<(N > 32), unsigned long long, unsigned>::type; // one bit per series
    constexpr mask_t NM = N >= 8 * (int)sizeof(mask_t) ? ~(mask_t)0 : (((mask_t)1 << (N % (8 * (int)sizeof(mask_t)))) - (mask_t)1);
    auto lowbit = [](mask_t m) __attribute__((always_inline)) { return N > 32 ? (int)__builtin_ctzll(m) : (int)__builtin_ctz((unsigned)m); };
    constexpr int QS = (N + 2) & ~1;           // series lanes 0..N-1 (+ one dummy slot)
    // the step's tape block is copied HBM -> LDS by the wavefront itself (global_load_lds: no registers, asynchronous):
    // 16 bytes a lane and instruction when the blocks are 16-byte aligned, 4 bytes otherwise
    constexpr int CB_BYTES = (RS % 2 == 0) ? 16 : 4, CHUNK = 64 * CB_BYTES;
    constexpr int NCH = (RS * 8 + CHUNK - 1) / CHUNK;
    constexpr int TBD = NCH * CHUNK / 8;       // doubles of the LDS image (block + what the last chunk over-reads)
    using G16 = Group<16>;
    // beyond two wavefronts' worth of registers the compiler parks values in AGPRs and reloads them right before their use -- a VALU
    // write it cannot know a DPP read follows: those instantiations take the guarded statements (mk_prims.h)
    constexpr bool GD = (n > 40);

    const int lane = threadIdx.x, i16 = lane & 15;
    long inst = (long)blockIdx.x;
    if (inst > a.B - 1) inst = a.B - 1;
    const long rec = inst % a.R;
    const int ra = lane < n ? lane : n;        // row held by this lane (n = the r row)
    const int js = lane < N ? lane : N - 1;    // series whose projection this lane writes (lanes >= N replicate N-1)
    const bool frow = lane >= N && lane < n;   // factor rows
    const long T = a.T;

    // wave-private LDS
    constexpr int GLT = STATE ? ((N * K + 1) & ~1) : 0; // STATE: loadings table [N][K]
    __shared__ __attribute__((aligned(16))) double lds[TBD + NP + NP + K * NP + K * K * QS + ((K * K + 1) & ~1) + GLT];
    double *tapeb = lds;                       // the step's tape block: N entries [ series part (N) | side row (SW) ]
    // ... and its LDS address for the copy, taken from the array itself: a run-time generic -> LDS cast of `tapeb + offset` is
    // folded by the compiler in most instantiations and mis-selected in one (hipcc 7.2, (48,3) inside a shape module:
    // "V_CMP_NE_U32 0, src_shared_base: operand has incorrect register class")
    lds_char_t *const tapeb3 = (lds_char_t *)lds;
    double *sideb = tapeb + SO;
    double *phim = tapeb + TBD;                // diag(Phi) [n]
    double *tbuf = phim + NP;                  // the new column j on its way to lane j's row
    double *fbuf = tbuf + NP;                  // [K][NP] factor columns on their way to the factor rows
    double *qbuf = fbuf + K * NP;              // [K*K][QS] products of the factor block
    double *ffb = qbuf + K * K * QS;           // [K*K] factor block sums
    [[maybe_unused]] double *gtab = ffb + ((K * K + 1) & ~1); // STATE: loadings [N][K]

    const double phi_own = lane < n ? a.phi[inst * n + lane] : 1.0; // row scaling of the transition (none for the r row)
    if (lane < n) phim[lane] = phi_own;
    wave_lds_sync();
    // C[a][k] = g_ak (phi_{N+k} - phi_a): own row (series lanes, for the factor block) and DPP-replicated columns
    double Cown[K], CB[K][NB];
    sfor<0, K>(MK_LAMBDA(kk) {
        constexpr int k = decltype(kk)::value;
        const double pfk = phim[N + k];
        Cown[k] = lane < N ? a.loadings[(rec * N + lane) * K + k] * (pfk - phi_own) : 0.0;
        sfor<0, NB>(MK_LAMBDA(mm) {
            constexpr int m = decltype(mm)::value;
            const int c = 16 * m + i16;
            const int cc = c < N ? c : N - 1;
            const double v = a.loadings[(rec * N + cc) * K + k] * (pfk - phim[cc]);
            CB[k][m] = c < N ? v : 0.0;
        });
    });
    [[maybe_unused]] const double rvar = HASR ? a.obsvar[rec * N + js] : 0.0;
    const double scale = a.scale ? a.scale[rec * N + js] : 1.0;
    const double offset = a.offset ? a.offset[rec * N + js] : 0.0;
    if constexpr (STATE) { // the loadings of this lane's series wait in LDS (registers are what this kernel is short of)
#pragma unroll
        for (int k = 0; k < K; ++k) gtab[js * K + k] = a.loadings[(rec * N + js) * K + k];
        wave_lds_sync();
    }

    double Nr[n];
#pragma unroll
    for (int c = 0; c < n; ++c) Nr[c] = 0.0;
    const double rrow = lane >= n ? 1.0 : 0.0; // indicator of the r row (and its replicas)

    // HBM -> LDS copy of one tape block (asynchronous; completion = vmcnt)
    const long blk_bytes = (long)RS * 8;
    auto fetch_block = [&](const double *blk) __attribute__((always_inline)) {
        const char *g = reinterpret_cast<const char *>(blk);
        sfor<0, NCH>(MK_LAMBDA(cc) {
            constexpr int c = decltype(cc)::value;
            long off = (long)c * CHUNK + (long)lane * CB_BYTES;
            if constexpr ((c + 1) * CHUNK > RS * 8) off = off < blk_bytes - CB_BYTES ? off : blk_bytes - CB_BYTES; // stay inside
            if constexpr (RS % 2 == 0)
                __builtin_amdgcn_global_load_lds((global_cvoid_t *)(g + off), (lds_void_t *)(tapeb3 + c * CHUNK), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((global_cvoid_t *)(g + off), (lds_void_t *)(tapeb3 + c * CHUNK), 4, 0, 0);
        });
    };

    // one tape entry in registers
    struct Ent {
        double XB[NB], XF, xa, s0, s1;
    };
    int xoff_b[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) xoff_b[m] = 16 * m + i16 < N ? 16 * m + i16 : N - 1;
    const int xoff_f = SO + (i16 < K ? i16 : K - 1);
    // element `lane` of the vector in its natural layout: series part, factor part, and for lanes >= n the entry's constant 0
    // (the same slot later receives this lane's product x_a w_a, resp. beta: see the unobserved entries below)
    const int xa_off = lane < N ? lane : SO + (lane < n ? lane - N : SW - 1);
    const int xa_str = lane < N ? XS : SS;
    auto load_ent = [&](int j, Ent &E) __attribute__((always_inline)) {
        const double *e = tapeb + j * XS, *sd = tapeb + j * SS;
        sfor<0, NB>(MK_LAMBDA(mm) { E.XB[decltype(mm)::value] = e[xoff_b[decltype(mm)::value]]; });
        E.XF = sd[xoff_f];
        E.xa = tapeb[xa_off + j * xa_str];
        if constexpr (PAIRS) {
            const v2d sc = *reinterpret_cast<const v2d *>(sideb + j * SS + SW - 4);
            E.s0 = sc.x;
            E.s1 = sc.y;
        } else {
            E.s0 = sideb[j * SS + SW - 4];
            E.s1 = sideb[j * SS + SW - 3];
        }
    };
    // w_a = sum_c N[a][c] x_c: the entry's vector broadcast inside the multiply-add (four per asm statement: hipcc pads
    // every asm statement that follows another with an s_nop)
    auto matvec = [&](const Ent &E) __attribute__((always_inline)) {
        double acc0 = 0.0, acc1 = 0.0;
        sfor<0, N / 16>(MK_LAMBDA(bb) { // sixteen columns per asm statement
            constexpr int c = 16 * decltype(bb)::value;
            G16::fmac16<GD>(acc0, acc1, E.XB[c / 16], Nr[c], Nr[c + 1], Nr[c + 2], Nr[c + 3], Nr[c + 4], Nr[c + 5], Nr[c + 6], Nr[c + 7], Nr[c + 8],
                        Nr[c + 9], Nr[c + 10], Nr[c + 11], Nr[c + 12], Nr[c + 13], Nr[c + 14], Nr[c + 15]);
        });
        sfor<0, (N % 16) / 4>(MK_LAMBDA(qq) {
            constexpr int c = 16 * (N / 16) + 4 * decltype(qq)::value;
            G16::fmac4<c % 16, (c + 1) % 16, (c + 2) % 16, (c + 3) % 16, false, GD>(acc0, acc1, E.XB[c / 16], Nr[c], E.XB[(c + 1) / 16], Nr[c + 1],
                                                                           E.XB[(c + 2) / 16], Nr[c + 2], E.XB[(c + 3) / 16], Nr[c + 3]);
        });
        sfor<16 * (N / 16) + 4 * ((N % 16) / 4), N>(MK_LAMBDA(cc) {
            constexpr int c = decltype(cc)::value;
            G16::fmac<c % 16, false, GD>(c % 2 ? acc1 : acc0, E.XB[c / 16], Nr[c]);
        });
        if constexpr (K == 4) {
            G16::fmac4<0, 1, 2, 3, false, GD>(acc0, acc1, E.XF, Nr[N], E.XF, Nr[N + 1], E.XF, Nr[N + 2], E.XF, Nr[N + 3]);
        } else {
            sfor<0, K>(MK_LAMBDA(kk) {
                constexpr int k = decltype(kk)::value;
                G16::fmac<k, false, GD>(k % 2 ? acc1 : acc0, E.XF, Nr[N + k]);
            });
        }
        return acc0 + acc1;
    };

    const double *tb = a.F + (inst * a.bs + (T - 1) * a.ts) * a.rs; // tape block of the current step (in HBM)
    const long tstep = a.ts * a.rs;
    double *omean = a.sim_means ? a.sim_means + (inst * a.bs + (T - 1) * a.ts) * N + js : nullptr;
    double *ovar = a.sim_vars ? a.sim_vars + (inst * a.bs + (T - 1) * a.ts) * N + js : nullptr;
    const long ostep = a.ts * N;
    [[maybe_unused]] const int sl = lane < n ? lane : n - 1; // STATE: the entry this lane reads (lanes >= n: any valid one)
    // lane a < n writes state a; the lanes beyond hold the r row's replicas and write nothing (two exec-masked stores a step)
    [[maybe_unused]] double *osm = (STATE && a.state_means && lane < n) ? a.state_means + (inst * a.bs + (T - 1) * a.ts) * n + lane : nullptr;
    [[maybe_unused]] double *osv = (STATE && a.state_vars && lane < n) ? a.state_vars + (inst * a.bs + (T - 1) * a.ts) * n + lane : nullptr;
    [[maybe_unused]] const long sstep = a.ts * n;

    fetch_block(tb);
    for (long t = T - 1; t >= 0; --t) {
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this step's block has landed in LDS
        wave_lds_sync();
        // the scalars of this lane's series: (s0, s1, s2); s2 = NaN marks "not observed at this step"
        double s0, s1, s2;
        {
            const double *e = sideb + js * SS + SW - 4;
            if constexpr (PAIRS) {
                const v2d sc = *reinterpret_cast<const v2d *>(e);
                s0 = sc.x;
                s1 = sc.y;
            } else {
                s0 = e[0];
                s1 = e[1];
            }
            s2 = e[2];
        }
        const bool unobs = (s2 != s2);
        const mask_t obsm = (mask_t)__ballot(lane < N && !unobs) & NM; // series observed at this step
        Ent ea, eb;

        // ---- STATE: the factor columns of Vt and the smoothed factor means, with (r, N) before the step's updates and BEFORE the
        // unobserved pass overwrites its entries' LDS slots
        [[maybe_unused]] double st_madd = 0.0, st_vadd = 0.0; // what the state outputs add to the projection's (mean, var)
        if constexpr (STATE) {
            {   // w_k = N pt_{N+k} in the natural layout (lane a: w_k[a]; the r row: r . pt_{N+k}) -> fbuf[k][.]
                Ent ef[2];
                load_ent(N, ef[0]);
                sfor<0, K>(MK_LAMBDA(kk) {
                    constexpr int k = decltype(kk)::value;
                    if constexpr (k + 1 < K) load_ent(N + k + 1, ef[(k + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const double w = matvec(ef[k & 1]);
                    fbuf[k * NP + ra] = w;
                });
            }
            wave_lds_sync();
            // c[k] = (this lane's entry) . w_k: the entry's n doubles are contiguous in the block (series part, factor part),
            // w_k DPP-replicated (lane 16 q + i holds w_k[16 m + i])
            const double *er = tapeb + sl * XS;
            double cacc[K], WBk[K][NB], WFk[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                cacc[k] = 0.0;
#pragma unroll
                for (int m = 0; m < NB; ++m) WBk[k][m] = fbuf[k * NP + xoff_b[m]];
                WFk[k] = fbuf[k * NP + N + (i16 < K ? i16 : K - 1)];
            }
            constexpr int CH = 8; // columns per piece: the row is read a piece ahead of its multiply-adds
            sfor<0, (n + CH - 1) / CH>(MK_LAMBDA(pp) {
                constexpr int c0 = CH * decltype(pp)::value, c1 = c0 + CH < n ? c0 + CH : n;
                double rowp[CH];
                if constexpr (PAIRS) {
#pragma unroll
                    for (int c = c0; c < c1; c += 2) {
                        const v2d t2 = *reinterpret_cast<const v2d *>(er + c);
                        rowp[c - c0] = t2.x;
                        rowp[c - c0 + 1] = t2.y;
                    }
                } else {
#pragma unroll
                    for (int c = c0; c < c1; ++c) rowp[c - c0] = er[c];
                }
                sfor<c0, c1>(MK_LAMBDA(cc) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr (c < N) {
                        if constexpr (K == 4)
                            G16::fmac4x<c % 16, GD>(cacc[0], cacc[1], cacc[2], cacc[3], WBk[0][c / 16], WBk[1][c / 16], WBk[2][c / 16],
                                                 WBk[3][c / 16], rowp[c - c0]);
                        else
                            sfor<0, K>(MK_LAMBDA(k2) { G16::fmac<c % 16, false, GD>(cacc[decltype(k2)::value], WBk[decltype(k2)::value][c / 16], rowp[c - c0]); });
                    } else {
                        if constexpr (K == 4)
                            G16::fmac4x<c - N, GD>(cacc[0], cacc[1], cacc[2], cacc[3], WFk[0], WFk[1], WFk[2], WFk[3], rowp[c - c0]);
                        else
                            sfor<0, K>(MK_LAMBDA(k2) { G16::fmac<c - N, false, GD>(cacc[decltype(k2)::value], WFk[decltype(k2)::value], rowp[c - c0]); });
                    }
                });
            });
            // Vt[a][N+k] = Pt[a][N+k] - c[k]  (Pt[a][N+k] is the factor part of entry a); zero where the series was observed
            double vaf[K];
            const bool live = frow || (lane < N && unobs);
#pragma unroll
            for (int k = 0; k < K; ++k) vaf[k] = live ? er[N + k] - cacc[k] : 0.0;
            const double xf_own = er[n] + fbuf[(frow ? lane - N : 0) * NP + n]; // lane N + k: x_f[N+k] + r . pt_{N+k}
            wave_lds_sync();                                                    // fbuf has been read
            if (frow) {
#pragma unroll
                for (int k = 0; k < K; ++k) ffb[(lane - N) * K + k] = vaf[k];   // row k' = lane - N of Vt_FF
                tbuf[lane - N] = xf_own;
            }
            wave_lds_sync();
            double gl[K];
            if constexpr (K % 2 == 0) load_row<K>(gtab + js * K, gl);
            else {
#pragma unroll
                for (int k = 0; k < K; ++k) gl[k] = gtab[js * K + k];
            }
            double quad = 0.0, cross = 0.0, gx = 0.0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                gx = fma(gl[k], tbuf[k], gx);
                cross = fma(gl[k], vaf[k], cross);
                double rowq = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < K; ++k2) rowq = fma(gl[k2], ffb[k * K + k2], rowq);
                quad = fma(gl[k], rowq, quad);
            }
            const int kf = frow ? lane - N : 0;
            st_madd = frow ? tbuf[kf] : -gx;                              // factor rows: the smoothed factor mean itself
            st_vadd = frow ? ffb[kf * K + kf] : fma(-2.0, cross, quad);   // ... and Vt_FF[k][k]
        }

        // ---- series not observed at this step: mean = s0 + pt.r, var = s1 - pt'N pt with (r, N) before the step's updates.
        // The products x_a w_a go back into the entry's own (consumed) LDS slots, beta into its constant-0 slot; lane u sums
        // its entry afterwards.
        mask_t um = ~obsm & NM;
        auto unobs_step = [&](Ent &E, Ent &Enext) __attribute__((always_inline)) {
            const int u = lowbit(um);
            um &= um - (mask_t)1;
            load_ent(um ? lowbit(um) : u, Enext); // the next entry's LDS reads in flight during this product
            __builtin_amdgcn_sched_barrier(0);
            const double w = matvec(E);
            tapeb[xa_off + u * xa_str] = (E.xa + rrow) * w; // x_a w_a; the r row (its slot of the entry holds 0): beta = w itself
        };
        // (Measured and dropped, round 4: pt vanishes on the coordinates observed at the step, so a column loop over the
        // unobserved series and the factors only -- 13.6 of 36 columns at configs[3], x_c as wavefront-uniform LDS reads for
        // ten entries at a time -- does 0.4x the multiply-adds; it ran the kernel at 71 ms against 56: a uniform LDS read per
        // multiply-add costs more than the 22 broadcast multiply-adds it saves.)
        if (um && !MK_TUNE_SKIP(a, 64)) {
            load_ent(lowbit(um), ea);
            while (true) {
                unobs_step(ea, eb);
                if (!um) break;
                unobs_step(eb, ea);
                if (!um) break;
            }
        }
        wave_lds_sync();
        double mean = s2, var = 0.0;              // observed, R = 0: the observation itself, variance 0 (HASR: set in the observed pass)
        if (unobs) {
            double q0 = 0.0, q1 = 0.0;
            double row[N], rowf[SW];
            load_row<N>(tapeb + js * XS, row);
            load_row<SW>(sideb + js * SS, rowf);
#pragma unroll
            for (int c = 0; c < N; ++c) {
                if (c % 2 == 0) q0 += row[c];
                else q1 += row[c];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (k % 2 == 0) q0 += rowf[k];
                else q1 += rowf[k];
            }
            mean = s0 + rowf[SW - 1];
            var = s1 - (q0 + q1);
        }
        if constexpr (STATE) { // smoothed state means / variances of Metran's basis (kalmanfilter.py:461-474: S, diag(Ps));
            // R = 0: the projection's (mean, var) are final here, before the observed pass
            if (osm) *osm = (frow ? 0.0 : mean) + st_madd;
            if (osv) *osv = (frow ? 0.0 : var) + st_vadd;
        }
        // ---- observed series, last first: the scalar updates of the filter walked backwards
        mask_t om = obsm;
        {
            // the loop over the series is UNROLLED (static j: an entry's LDS addresses are immediates, the column N[.][j] is a
            // named register -- no bit scan, no address arithmetic, no jump tables for N[.][j]); every position prefetches the
            // next entry whether it runs or not, a wavefront-uniform branch skips the bodies of the unobserved series
            if (om) {
                Ent eo[2];
                load_ent(N - 1, eo[(N - 1) & 1]);
                sfor_down<0, N>(MK_LAMBDA(jj) {
                    constexpr int j = decltype(jj)::value;
                    if constexpr (j > 0) load_ent(j - 1, eo[(j - 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (om & ((mask_t)1 << j)) {
                        const Ent &E = eo[j & 1];
                        const double w = matvec(E);
                        [[maybe_unused]] double beta = 0.0;
                        if constexpr (HASR) beta = readlane_f64(w, n);
                        const double alpha = MK_TUNE_SKIP(a, 32) ? E.xa * w : wave_sum_mfma(E.xa * w);   // its MFMA chain runs under what follows
                        // rows a < n: N[a][j] - w_a.  The r row (lanes >= n): r_j + v/f - beta -- and beta = r . kt IS that row's own w, so
                        // one multiply-add with the row indicator replaces the lane read of beta, a subtraction, an addition and two
                        // selects (round 6: the loop is bound by its instruction count, five fewer per entry are 4 % of the step)
                        const double nc = fma(rrow, E.s0, Nr[j] - w);
                        Nr[j] = nc;
                        if (!MK_TUNE_SKIP(a, 16)) {
                            tbuf[ra] = nc;                                    // slot n: the r row's, unused
                            wave_lds_sync();
                            // row j = the new column j, 16 bytes at a time into ONE lane -- every element but the diagonal, whose
                            // register is written below from alpha: no read lands in it, so that write waits for no LDS
                            // operation and the next entry's product starts as soon as the first reads are back
                            if (lane == j) {
                                sfor<0, (n + 1) / 2>(MK_LAMBDA(mm) {
                                    constexpr int c0 = 2 * decltype(mm)::value;
                                    if constexpr (c0 == (j & ~1) || c0 + 1 >= n) {
                                        if constexpr (c0 != j) Nr[c0] = tbuf[c0];
                                        if constexpr (c0 + 1 != j && c0 + 1 < n) Nr[c0 + 1] = tbuf[c0 + 1];
                                    } else {
                                        const v2d t2 = *reinterpret_cast<const v2d *>(tbuf + c0);
                                        Nr[c0] = t2.x;
                                        Nr[c0 + 1] = t2.y;
                                    }
                                });
                            }
                        }
                        // the diagonal N[j][j] - 2 w_j + alpha + 1/f, in lane j's own register, once alpha has arrived
                        if (lane == j) Nr[j] = (nc - w) + (alpha + E.s1);
                        if constexpr (HASR) {
                            if (js == j) { // (lanes >= N replicate series N-1: the same values to the same address)
                                mean = s2 - rvar * (E.s0 - beta);
                                var = rvar * (1.0 - rvar * E.s1) - rvar * rvar * alpha;
                            }
                        }
                    }
                });
            }
        }
        {
            const double v = scale * scale * var;
            if (omean) *omean = fma(scale, mean, offset);
            if (ovar) *ovar = v < 0.0 ? 0.0 : v;  // kalmanfilter.py:601-602 (np.maximum keeps a NaN)
        }

        // every entry of the step has been consumed: next step's block on its way while the transition runs
        tb -= tstep;
        if (t > 0) {
            __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): no LDS read of the old block is still in flight
            fetch_block(tb);
        }

        // ---- transition: r <- Pht'r, N <- Pht'N Pht
        if (t > 0 && !MK_TUNE_SKIP(a, 128)) {
            double Yf[K];
            sfor<0, K>(MK_LAMBDA(kk) { // factor columns: Y[a][N+k] = sum_c N[a][c] C[c][k] + N[a][N+k] phi_{N+k}
                constexpr int k = decltype(kk)::value;
                double y0 = Nr[N + k] * phim[N + k], y1 = 0.0;
                sfor<0, N / 4>(MK_LAMBDA(qq) {
                    constexpr int c = 4 * decltype(qq)::value;
                    G16::fmac4<c % 16, (c + 1) % 16, (c + 2) % 16, (c + 3) % 16, false, GD>(y0, y1, CB[k][c / 16], Nr[c], CB[k][(c + 1) / 16], Nr[c + 1],
                                                                                   CB[k][(c + 2) / 16], Nr[c + 2], CB[k][(c + 3) / 16],
                                                                                   Nr[c + 3]);
                });
                sfor<4 * (N / 4), N>(MK_LAMBDA(cc) {
                    constexpr int c = decltype(cc)::value;
                    G16::fmac<c % 16, false, GD>(c % 2 ? y1 : y0, CB[k][c / 16], Nr[c]);
                });
                Yf[k] = y0 + y1;
            });
            {   // series columns: N[a][c] phi_a phi_c
                double ph[NP];
                load_row<NP>(phim, ph);
#pragma unroll
                for (int c = 0; c < N; ++c) Nr[c] = (Nr[c] * phi_own) * ph[c];
            }
            // factor block sums sum_a C[a][k] Y[a][N+k'] over the series rows, factor columns to the factor rows
            const int qs = lane < N ? lane : N;
            sfor<0, K>(MK_LAMBDA(kk) {
                constexpr int k = decltype(kk)::value;
                sfor<0, K>(MK_LAMBDA(ll) {
                    constexpr int l = decltype(ll)::value;
                    qbuf[(k * K + l) * QS + qs] = Cown[k] * Yf[l];
                });
                fbuf[k * NP + ra] = phi_own * Yf[k];
            });
            wave_lds_sync();
            {
                const int L = lane < K * K ? lane : K * K - 1;
                double row[QS];
                load_row<QS>(qbuf + L * QS, row);
                double q0 = 0.0, q1 = 0.0;
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    if (c % 2 == 0) q0 += row[c];
                    else q1 += row[c];
                }
                ffb[L] = q0 + q1;
            }
            if (frow) { // rows N+k, series part: N'[N+k][c] = N'[c][N+k]
                double row[NP];
                load_row<NP>(fbuf + (lane - N) * NP, row);
#pragma unroll
                for (int c = 0; c < N; ++c) Nr[c] = row[c];
            }
            wave_lds_sync();
            const int kf = frow ? lane - N : 0;
            sfor<0, K>(MK_LAMBDA(ll) {
                constexpr int l = decltype(ll)::value;
                const double add = frow ? ffb[kf * K + l] : 0.0;
                Nr[N + l] = fma(phi_own, Yf[l], add);
            });
        }
        if (omean) omean -= ostep;
        if (ovar) ovar -= ostep;
        if constexpr (STATE) {
            if (osm) osm -= sstep;
            if (osv) osv -= sstep;
        }
    }
}

// Which calls the tape path serves: wide models whose rows and the r row fit a wavefront (16 < n <= 63), projection / state outputs.
template <int N, int K>
static hipError_t launch_dk_nk(const SmootherArgs &a, hipStream_t s)
{
    if constexpr (N + K > 16 && N + K + 1 <= 64 && K <= 16) {
        if (a.tape == 2) { // the STATE tape: smoothed state means / variances (and, if asked for, the projection)
            if (a.rs != state_tape_stride_c(N, K) || a.obsvar) return hipErrorInvalidValue;
            hipLaunchKernelGGL((smoother_dk_kernel<N, K, false, true>), dim3((unsigned)a.B), dim3(64), 0, s, a);
            return hipGetLastError();
        }
        if (a.rs != tape_stride_c(N, K)) return hipErrorInvalidValue;
        if (a.obsvar) hipLaunchKernelGGL((smoother_dk_kernel<N, K, true>), dim3((unsigned)a.B), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((smoother_dk_kernel<N, K, false>), dim3((unsigned)a.B), dim3(64), 0, s, a);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}
#define MK_CASE_DK(NN, KK) \
    if (N == NN && K == KK) return launch_dk_nk<NN, KK>(a, s);
hipError_t launch_smoother_dk(int N, int K, const SmootherArgs &a, hipStream_t s)
{
    MK_SHAPES(MK_CASE_DK)
    return hipErrorNotSupported;
}

} // namespace mk
