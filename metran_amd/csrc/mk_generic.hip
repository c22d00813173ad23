// mk_generic.hip -- size-generic kernels: the same filter / objective / smoother for ANY Metran model shape (N series, K factors)
// up to MK_GENERIC_MAX_STATES states, with no specialisation and no compiler at run time.
// Reference semantics: seqkalmanfilter (/root/reference/metran/kalmanfilter.py:236-400: predict :318-333, scalar updates
// :341-378, compressed bookkeeping :380-382, filtered moments :384-390), get_mle (:550-567), kalmansmoother (:403-476) and
// simulate (:569-603) -- whose loops are size-generic: a 70-series model is as legal there as a 5-series one.
//
// Why they exist (round 5).  The fast kernels of this library are fully unrolled over the state dimension: one state (or series)
// per lane, covariance rows in registers -- n <= 64, one instantiation per (N, K), built ahead of time for a handful of shapes
// and by hipcc at run time for the others.  A model outside that envelope (n > 64), or any non-listed shape on a machine
// without hipcc, used to be refused.  These kernels close the gap: correct for every shape, not tuned for any.
//   mapping      one model per WORKGROUP, time sequential, parallel over the matrix ELEMENTS (a flat index e -> (row, column) through
//                a multiply-high, so every lane carries an element whatever n is): one wavefront for the small models (the
//                barriers between the phases of a step then cost nothing), four for the others
//   filter       the covariance P (n x n), the loadings and the observation variances live in LDS (dynamic: n^2 + 4n + 2N + N K
//                doubles; 160 KiB of LDS per workgroup bound n at MK_GENERIC_MAX_STATES = 128), Z = [I | G] is exploited as in the
//                fast kernels (d = P z' costs 1 + K terms per row); two barriers per scalar update; log det F of a step is ONE
//                logarithm (mantissa product + exponent sum, as in the fast kernels)
//   smoother     the reference's RTS recursion with an LDL^T of the predicted covariance in place of its pinv (same contract as
//                the fast kernels: positive pivots inverted, a pivot <= 0 dropped with MK_FLAG_RANK_DEFICIENT, < -1e-8 is
//                MK_FLAG_NOT_SPD).  Round 6: the factorisation runs on the AUGMENTED matrix [Pp | Phi Pf] -- the forward
//                substitution of all n right-hand sides rides in the pivot's trailing update, the back substitution is n more
//                element-parallel sweeps (one barrier each; round 5 gave every right-hand side to ONE thread: n^2 dependent
//                loads deep) -- and its four n x n work matrices live in LDS where they fit (n <= 64), else in a global
//                workspace (L2-resident for one workgroup)
//   layouts      dense arrays or full-square packed records (mk_outputs.record_stride), both time orders; no packed-symmetric
//                records, no tape
// The smoother also serves kalmansmoother's literal 5-argument form (mk_smooth_dense: the CALLER's predicted moments are used,
// not recomputed -- kalmanfilter.py:453-474 reads predicted_state_means / predicted_state_covariances as they are handed in).
#include <hip/hip_runtime.h>

#include <mutex>

#include "mk_generic.h"

namespace mk {

namespace {
constexpr double kGenLog2Pi = 1.8378770664093454835606594728112;
constexpr double kGenLn2 = 0.69314718055994530941723212145818;

__device__ __forceinline__ long blk_index(long inst, long t, long bs, long ts) { return inst * bs + t * ts; }

// row of the flat index e of a row-major matrix with w columns, without an integer division: floor(e / w) is the high half of
// e * (floor(2^32 / w) + 1) as long as e < 2^32 / w  (here e < 2 * 128^2)
struct RowOf {
    unsigned m;
    __device__ __forceinline__ explicit RowOf(int w) : m(w > 1 ? (unsigned)(4294967296.0 / (double)w) + 1u : 0u) {}   // (a 64-bit integer division is ~100 instructions, and this runs once per pivot)
    __device__ __forceinline__ int operator()(int e) const { return m ? (int)__umulhi((unsigned)e, m) : e; }
};

// 1 / x to ~1 ulp for the O(1) variances and pivots met here (the fast kernels' contract, mk_prims.h: v_rcp_f64 -- relative error
// < 2^-25 -- and one cubically convergent step; the IEEE division is ~35 dependent-ish instructions on this chip, and an update's
// chain waits on it)
__device__ __forceinline__ double gen_rcp(double x)
{
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r0, 1.0);
    return fma(r0, fma(e, e, e), r0);
}

// log of a mantissa m in [1/2, 1) (the running product of a step's innovation variances): 2 atanh((m - 1) / (m + 1)) after moving
// m into [1/sqrt 2, sqrt 2), ten terms (truncation 2e-17); anything else (a NaN product) goes to libm
__device__ __forceinline__ double gen_log_mant(double m)
{
    if (!(m >= 0.5 && m < 1.0)) return log(m);
    const bool low = m < 0.70710678118654752440;
    const double m2 = low ? m + m : m;
    const double t = (m2 - 1.0) * gen_rcp(m2 + 1.0), w = t * t;
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
    return fma(t + t, p, low ? -kGenLn2 : 0.0);
}

// threads of a model's workgroup: one wavefront while a lane has at most ~16 covariance elements to carry, else four
#ifndef MK_GEN_FT
#define MK_GEN_FT 32
#endif
#ifndef MK_GEN_ST
#define MK_GEN_ST 20
#endif
#ifndef MK_GEN_FT2
#define MK_GEN_FT2 64
#endif
#ifndef MK_GEN_ST2
#define MK_GEN_ST2 40
#endif
// ... and sixteen where LDS leaves room for one or two models per CU: the model's own threads are then all the latency hiding there is
inline int filter_threads(int n) { return n <= MK_GEN_FT ? 64 : n <= MK_GEN_FT2 ? 256 : 1024; }
inline int smoother_threads(int n) { return n <= MK_GEN_ST ? 64 : n <= MK_GEN_ST2 ? 256 : 1024; }
} // namespace

// ---------------------------------------------------------------------------------------------------------------- filter
// GL: the loadings table [N][K] is in LDS (it always is unless N K doubles on top of the covariance exceed the device's LDS)
template <int NT, bool GL>
__global__ void __launch_bounds__(NT) filter_generic_kernel(FilterArgs a, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int n = N + K, nn = n * n, tid = threadIdx.x;
    double *P = gsm;                     // [n][n]
    double *x = P + nn;                  // [n]
    double *d = x + n;                   // [n]  P z_j'
    double *phi = d + n;                 // [n]
    double *qv = phi + n;                // [n]
    double *yv = qv + n;                 // [N]  observations of the step
    double *Rs = yv + N;                 // [N]  observation variances
    [[maybe_unused]] double *Gs = Rs + N; // [N][K] loadings (GL)
    const long inst = blockIdx.x, rec = inst % a.R;
    const double *Gg = a.loadings + rec * (long)N * K;
    const long T = a.T;
    const RowOf row(n);
    for (int i = tid; i < n; i += NT) {
        phi[i] = a.phi[inst * n + i];
        qv[i] = a.q[inst * n + i];
        x[i] = a.x0 ? a.x0[inst * n + i] : 0.0;                     // run_filter defaults (:747-750)
    }
    for (int j = tid; j < N; j += NT) Rs[j] = a.obsvar ? a.obsvar[rec * N + j] : 0.0;
    if constexpr (GL)
        for (int i = tid; i < N * K; i += NT) Gs[i] = Gg[i];
    for (int e = tid; e < nn; e += NT) {
        const int r = row(e), c = e - r * n;
        P[e] = a.P0 ? a.P0[inst * (long)nn + e] : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    auto G = [&](int j, int k) __attribute__((always_inline)) -> double {
        if constexpr (GL) return Gs[j * K + k];
        else return Gg[j * K + k];
    };

    const long SF = a.rs > 0 ? a.rs : n;              // doubles between the (b, t) blocks of the mean arrays ...
    const long SP = a.rs > 0 ? a.rs : (long)nn;       // ... and of the covariance arrays (records: d_Pf = d_F + n, same stride)
    auto emit = [&](double *mean, double *cov, long blk) {
        if (mean)
            for (int i = tid; i < n; i += NT) mean[blk * SF + i] = x[i];
        if (cov)
            for (int i = tid; i < nn; i += NT) cov[blk * SP + i] = P[i];
    };

    double sum_sig = 0.0, sum_det = 0.0;   // every thread keeps the same scalars
    bool bad_f = false;
    long nobs = 0, sc = 0;
    for (long t = 0; t < T; ++t) {
        const long blk = blk_index(inst, t, a.bs, a.ts);
        for (int j = tid; j < N; j += NT) yv[j] = a.obs[(rec * a.obs_bs + t * a.obs_ts) * N + j];
        // ---- predict (:318-331; Phi diagonal)
        for (int i = tid; i < n; i += NT) x[i] = phi[i] * x[i];
        for (int e = tid; e < nn; e += NT) {
            const int r = row(e), c = e - r * n;
            P[e] = fma(P[e] * phi[r], phi[c], r == c ? qv[r] : 0.0);
        }
        __syncthreads();
        emit(a.Xp, a.Pp, blk);                         // :332-333
        // ---- sequential scalar updates (:341-378), ascending series order
        double sigma = 0.0, fmant = 1.0, poison = 0.0;  // log det F of the step = log(fmant) + fexp ln 2 (+ log f of a non-positive f)
        int fexp = 0, cnt = 0;
        for (int j = 0; j < N; ++j) {
            const double y = yv[j];
            if (!isfinite(y)) continue;                // NaN / inf = missing (:657); the same for every thread
            ++cnt;
            for (int r = tid; r < n; r += NT) {        // d = P z_j' (:349-357): Z = [I | G]
                double s = P[r * n + j];
                for (int k = 0; k < K; ++k) s = fma(P[r * n + N + k], G(j, k), s);
                d[r] = s;
            }
            double v = y - x[j];                       // :344-347 (x is not written before the barrier)
            for (int k = 0; k < K; ++k) v = fma(-G(j, k), x[N + k], v);
            __syncthreads();
            double f = Rs[j] + d[j];                   // :359-362
            for (int k = 0; k < K; ++k) f = fma(G(j, k), d[N + k], f);
            const double rf = gen_rcp(f);
            for (int i = tid; i < n; i += NT) x[i] = fma(d[i] * rf, v, x[i]);      // :374-375
            for (int e = tid; e < nn; e += NT) {       // P -= k k' f (:368-372)
                const int r = row(e), c = e - r * n;
                P[e] = fma(-(d[r] * rf), d[c], P[e]);
            }
            sigma = fma(v * v, rf, sigma);             // :377
            if (f > 0.0) {                             // :378, detf += log f
                fmant *= f;
                fexp += __builtin_amdgcn_frexp_exp(fmant);
                fmant = __builtin_amdgcn_frexp_mant(fmant);
            } else {                                   // f <= 0 or NaN: the reference's log gives -inf / NaN, and so does this
                poison += log(f);
                bad_f = true;
            }
            __syncthreads();
        }
        if (cnt > 0) {                                 // :380-382, compressed index sc
            const double detf = fma((double)fexp, kGenLn2, gen_log_mant(fmant)) + poison;
            if (tid == 0) {
                if (a.sigmas) a.sigmas[blk_index(inst, sc, a.bs, a.ts) * a.sig_stride] = sigma;
                if (a.detfs) a.detfs[blk_index(inst, sc, a.bs, a.ts) * a.sig_stride] = detf;
            }
            if (sc >= a.warmup) {                      // get_mle: COMPRESSED indices (:563-564)
                sum_sig += sigma;
                sum_det += detf;
            }
            ++sc;
        }
        if (t >= a.warmup) nobs += cnt;                // observation_count[warmup:] is a TIME index (:565)
        emit(a.F, a.Pf, blk);                          // :384-390
        __syncthreads();
    }
    for (long i = sc + tid; i < T; i += NT) {         // zero tail of the compressed arrays (np.zeros, :307-308)
        if (a.sigmas) a.sigmas[blk_index(inst, i, a.bs, a.ts) * a.sig_stride] = 0.0;
        if (a.detfs) a.detfs[blk_index(inst, i, a.bs, a.ts) * a.sig_stride] = 0.0;
    }
    if (tid == 0) {
        if (a.mle) a.mle[inst] = ((double)nobs * kGenLog2Pi + sum_det) + sum_sig;   // :566
        if (a.sigmacount) a.sigmacount[inst] = sc;
        if (a.status) a.status[inst] = bad_f ? MK_FLAG_NONPOSITIVE_F : 0u;
    }
}

// -------------------------------------------------------------------------------------------------------------- smoother
// LW: which of the four n x n work matrices are in LDS -- 3: all of them (n <= 64 on a 160 KiB device); 2: the right-hand sides X
// and the factorised matrix A (n <= 100); 1: X alone (it takes part in every sweep and both products).  The others live in the
// global workspace g.ws (4 of its 5 n^2 doubles per model; L2-resident).
template <int NT, int LW>
__global__ void __launch_bounds__(NT) smoother_generic_kernel(GenericSmootherArgs g)
{
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const SmootherArgs &a = g.a;
    const int N = g.N, K = g.K, n = N + K, tid = threadIdx.x;
    const int nn = n * n;
    const int nv = (6 * n + 1) & ~1;
    double *xs = gsm;          // smoothed mean of step t+1
    double *xsn = xs + n;      // ... of step t
    double *delta = xsn + n;   // xs[t+1] - Xp[t+1]
    double *phi = delta + n;
    double *qv = phi + n;
    double *dinv = qv + n;     // D^+ of the factorisation
    const long inst = blockIdx.x, rec = inst % (a.R > 0 ? a.R : 1);
    double *const wsb = g.ws + inst * 5 * (long)nn;
    // A (row stride sa): Pp[t+1], then its factors (d on the diagonal, d_min(i,j) L off it), last V = Dm J'
    // X (row stride sx): the right-hand sides Phi Pf[t], then J' (X[c][i] = J[i][c])
    // Dm: Ps[t+1] - Pp[t+1];  Psn: smoothed covariance of step t+1, then of step t
    // With both A and X in LDS they are the two halves of ONE augmented matrix [A | X] of row stride 2n: the pivot's trailing update
    // is then a single expression over its columns k+1 .. 2n-1.
    constexpr bool AUG = LW >= 2;
    const int sa = AUG ? 2 * n : n, sx = sa;
    double *A, *X, *Dm, *Psn;
    if constexpr (LW == 3) { A = gsm + nv; X = A + n; Dm = A + 2 * nn; Psn = Dm + nn; }
    else if constexpr (LW == 2) { A = gsm + nv; X = A + n; Dm = wsb; Psn = wsb + nn; }
    else { X = gsm + nv; A = wsb; Dm = wsb + nn; Psn = wsb + 2 * (long)nn; }
    const long T = a.T;
    const long SF = a.rs > 0 ? a.rs : n, SP = a.rs > 0 ? a.rs : (long)nn;
    const double *G = a.loadings ? a.loadings + rec * (long)N * K : nullptr;
    const RowOf row(n);
    const int nh = (n + 1) >> 1, ntile = nh * nh;      // 2 x 2 output tiles of the two products (when every thread gets one)
    const bool tiled = ntile >= NT;
    const RowOf rowh(nh);
    for (int i = tid; i < n; i += NT) {
        phi[i] = a.phi[inst * n + i];
        qv[i] = a.q ? a.q[inst * n + i] : 0.0;
    }
    unsigned flags = 0;

    // outputs of one step from (xs, Psn): smoothed records / dense arrays, VAR_ONLY, fused projection (:569-603)
    auto emit = [&](long blk) {
        if (a.S)
            for (int i = tid; i < n; i += NT) a.S[blk * SF + i] = xs[i];
        if (a.Ps)
            for (int i = tid; i < nn; i += NT) a.Ps[blk * SP + i] = Psn[i];
        if (a.state_means)
            for (int i = tid; i < n; i += NT) a.state_means[blk * n + i] = xs[i];
        if (a.state_vars)
            for (int i = tid; i < n; i += NT) a.state_vars[blk * n + i] = Psn[i * n + i];
        if ((a.sim_means || a.sim_vars) && G) {
            for (int j = tid; j < N; j += NT) {
                const double sc = a.scale ? a.scale[rec * N + j] : 1.0, off = a.offset ? a.offset[rec * N + j] : 0.0;
                double m = xs[j], var = Psn[j * n + j];
                for (int k = 0; k < K; ++k) {
                    const double gk = G[j * K + k];
                    m = fma(gk, xs[N + k], m);
                    double rw = Psn[j * n + N + k] + Psn[(N + k) * n + j];   // (j, N+k) + (N+k, j)
                    for (int k2 = 0; k2 < K; ++k2) rw = fma(G[j * K + k2], Psn[(N + k) * n + N + k2], rw);
                    var = fma(gk, rw, var);
                }
                const double v = sc * sc * var;
                if (a.sim_means) a.sim_means[blk * N + j] = fma(sc, m, off);
                if (a.sim_vars) a.sim_vars[blk * N + j] = v < 0.0 ? 0.0 : v;     // :601-602 (np.maximum keeps a NaN)
            }
        }
    };

    // last step: smoothed = filtered (:450-451)
    {
        const long blk = blk_index(inst, T - 1, a.bs, a.ts);
        for (int i = tid; i < n; i += NT) xs[i] = a.F[blk * SF + i];
        for (int i = tid; i < nn; i += NT) Psn[i] = a.Pf[blk * SP + i];
        __syncthreads();
        emit(blk);
        __syncthreads();
    }
    for (long t = T - 2; t >= 0; --t) {
        const long blk = blk_index(inst, t, a.bs, a.ts), blk1 = blk_index(inst, t + 1, a.bs, a.ts);
        const double *Pft = a.Pf + blk * SP, *Ft = a.F + blk * SF;
        // A = Pp[t+1] (the caller's, or Phi Pf[t] Phi + Q), Dm = Ps[t+1] - Pp[t+1], X = Phi Pf[t] (row r scaled: Pf symmetric)
        for (int e = tid; e < nn; e += NT) {
            const int r = row(e), c = e - r * n;
            const double pr = phi[r], pf = Pft[e];
            const double pp = g.Pp ? g.Pp[blk1 * (long)nn + e] : fma(pr * pf, phi[c], r == c ? qv[r] : 0.0);
            A[r * sa + c] = pp;
            Dm[e] = Psn[e] - pp;
            X[r * sx + c] = pr * pf;
        }
        for (int i = tid; i < n; i += NT) delta[i] = xs[i] - (g.Xp ? g.Xp[blk1 * n + i] : phi[i] * Ft[i]);
        __syncthreads();
        // ---- LDL' of A with the forward substitution of the n right-hand sides in the same trailing update (one barrier per pivot):
        //      rows i > k of [A | X], columns j > k of A and every column of X:  . -= (A[i][k] / d_k) (row k)
        //      (both triangles of A are updated: row k stays d_k L(.,k)' and the sweep reads rows, never columns)
        for (int k = 0; k < n; ++k) {
            const double piv = A[k * sa + k];
            double di = 0.0;
            if (piv > 0.0) di = gen_rcp(piv);
            else {
                flags |= MK_FLAG_RANK_DEFICIENT;       // null direction dropped (the reference's pinv, :455)
                if (piv < -1e-8 || !(piv == piv)) flags |= MK_FLAG_NOT_SPD;
            }
            if (tid == 0) dinv[k] = di;
            const int m = n - 1 - k, w = m + n, tot = m * w;
            const RowOf roww(w);
            for (int e = tid; e < tot; e += NT) {
                const int ri = roww(e), i = k + 1 + ri, jj = e - ri * w;
                const double lik = A[i * sa + k] * di;
                if constexpr (AUG) {
                    const int j = k + 1 + jj;          // columns k+1 .. n-1 of A, then (j >= n) the columns of X = A + n
                    A[i * sa + j] = fma(-lik, A[k * sa + j], A[i * sa + j]);
                } else if (jj < m) {
                    const int j = k + 1 + jj;
                    A[i * sa + j] = fma(-lik, A[k * sa + j], A[i * sa + j]);
                } else {
                    const int c = jj - m;
                    X[i * sx + c] = fma(-lik, X[k * sx + c], X[i * sx + c]);
                }
            }
            __syncthreads();
        }
        // ---- D^+, then L' x = z by right-looking sweeps from the last row up: rows i < k:  X[i][.] -= L(k,i) X[k][.]
        for (int e = tid; e < nn; e += NT) {
            const int r = row(e);
            X[r * sx + (e - r * n)] *= dinv[r];
        }
        __syncthreads();
        for (int k = n - 1; k > 0; --k) {
            const int tot = k * n;
            for (int e = tid; e < tot; e += NT) {
                const int i = row(e), c = e - i * n;
                const double lki = A[k * sa + i] * dinv[i];
                X[i * sx + c] = fma(-lki, X[k * sx + c], X[i * sx + c]);
            }
            __syncthreads();
        }
        // ---- S[t] = F[t] + J delta (:461-464);  V = Dm J' (into A, which the sweeps have left)
        for (int i = tid; i < n; i += NT) {
            double s = Ft[i];
            for (int c = 0; c < n; ++c) s = fma(X[c * sx + i], delta[c], s);
            xsn[i] = s;
        }
        if (tiled) {
            for (int tt = tid; tt < ntile; tt += NT) {
                const int tr = rowh(tt), ti = tt - tr * nh;
                const int r0 = 2 * tr, i0 = 2 * ti, r1 = r0 + 1 < n ? r0 + 1 : r0, i1 = i0 + 1 < n ? i0 + 1 : i0;
                double s00 = 0.0, s01 = 0.0, s10 = 0.0, s11 = 0.0;
                for (int c = 0; c < n; ++c) {
                    const double d0 = Dm[r0 * n + c], d1 = Dm[r1 * n + c], x0 = X[c * sx + i0], x1 = X[c * sx + i1];
                    s00 = fma(d0, x0, s00);
                    s01 = fma(d0, x1, s01);
                    s10 = fma(d1, x0, s10);
                    s11 = fma(d1, x1, s11);
                }
                A[r0 * sa + i0] = s00;
                if (i1 != i0) A[r0 * sa + i1] = s01;
                if (r1 != r0) {
                    A[r1 * sa + i0] = s10;
                    if (i1 != i0) A[r1 * sa + i1] = s11;
                }
            }
        } else {
            for (int e = tid; e < nn; e += NT) {
                const int r = row(e), i = e - r * n;
                double s = 0.0;
                for (int c = 0; c < n; ++c) s = fma(Dm[r * n + c], X[c * sx + i], s);
                A[r * sa + i] = s;
            }
        }
        __syncthreads();
        // ---- Ps[t] = Pf[t] + J (Ps[t+1] - Pp[t+1]) J' (:465-474):  Ps[i][j] = Pf[i][j] + sum_r J[i][r] V[r][j]
        if (tiled) {
            for (int tt = tid; tt < ntile; tt += NT) {
                const int tr = rowh(tt), tj = tt - tr * nh;
                const int i0 = 2 * tr, j0 = 2 * tj, i1 = i0 + 1 < n ? i0 + 1 : i0, j1 = j0 + 1 < n ? j0 + 1 : j0;
                double s00 = Pft[i0 * n + j0], s01 = Pft[i0 * n + j1], s10 = Pft[i1 * n + j0], s11 = Pft[i1 * n + j1];
                for (int r = 0; r < n; ++r) {
                    const double x0 = X[r * sx + i0], x1 = X[r * sx + i1], v0 = A[r * sa + j0], v1 = A[r * sa + j1];
                    s00 = fma(x0, v0, s00);
                    s01 = fma(x0, v1, s01);
                    s10 = fma(x1, v0, s10);
                    s11 = fma(x1, v1, s11);
                }
                Psn[i0 * n + j0] = s00;
                if (j1 != j0) Psn[i0 * n + j1] = s01;
                if (i1 != i0) {
                    Psn[i1 * n + j0] = s10;
                    if (j1 != j0) Psn[i1 * n + j1] = s11;
                }
            }
        } else {
            for (int e = tid; e < nn; e += NT) {
                const int i = row(e), j = e - i * n;
                double s = Pft[e];
                for (int r = 0; r < n; ++r) s = fma(X[r * sx + i], A[r * sa + j], s);
                Psn[e] = s;
            }
        }
        for (int i = tid; i < n; i += NT) xs[i] = xsn[i];
        __syncthreads();
        emit(blk);
        __syncthreads();
    }
    if (tid == 0 && a.status) a.status[inst] |= flags;
}

// ------------------------------------------------------------------------------------------------------------- launchers
size_t generic_filter_lds_bytes(int N, int K)
{
    const size_t n = (size_t)(N + K);
    return (n * n + 4 * n + 2 * (size_t)N + 2) * sizeof(double);   // without the loadings table (N K more where the device has room)
}
size_t generic_smoother_ws_doubles(long B, int n) { return (size_t)B * 5 * (size_t)n * (size_t)n; }

namespace {
// above the 64 KiB default a kernel has to be given its dynamic LDS size explicitly -- PER DEVICE (round-5 advice: the grant
// used to be remembered per thread, so a thread driving a second GPU skipped the call there and its launch failed).  `slot` names
// the kernel instantiation.  *cap: the device's LDS per workgroup.
constexpr int MAXDEV = 64, NSLOT = 15;
hipError_t lds_grant(const void *kernel, int slot, size_t lds, int *cap)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    static std::mutex grant_mutex;
    static size_t granted[MAXDEV][NSLOT] = {};
    static int lds_cap[MAXDEV] = {};
    std::lock_guard<std::mutex> lock(grant_mutex);
    if (dev < 0 || dev >= MAXDEV) return hipErrorInvalidDevice;
    if (!lds_cap[dev]) {
        e = hipDeviceGetAttribute(&lds_cap[dev], hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
        if (e != hipSuccess) return e;
    }
    if (cap) *cap = lds_cap[dev];
    if (!kernel) return hipSuccess;
    if (lds > (size_t)lds_cap[dev]) return hipErrorNotSupported;
    if (lds > granted[dev][slot] && lds > 64 * 1024) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        granted[dev][slot] = lds;
    }
    return hipSuccess;
}

template <int NT, bool GL>
hipError_t launch_filter_nt(int N, int K, const FilterArgs &a, size_t lds, hipStream_t s)
{
    const hipError_t e = lds_grant(reinterpret_cast<const void *>(&filter_generic_kernel<NT, GL>), (NT == 64 ? 0 : NT == 256 ? 2 : 4) + (GL ? 1 : 0), lds, nullptr);
    if (e != hipSuccess) return e;   // hipErrorNotSupported = MK_ERR_SHAPE at the C ABI: N + K too large for this device's LDS
    hipLaunchKernelGGL((filter_generic_kernel<NT, GL>), dim3((unsigned)a.B), dim3(NT), lds, s, a, N, K);
    return hipGetLastError();
}

template <int NT, int LW>
hipError_t launch_smoother_nt(const GenericSmootherArgs &g, size_t lds, hipStream_t s)
{
    const hipError_t e = lds_grant(reinterpret_cast<const void *>(&smoother_generic_kernel<NT, LW>), 6 + (NT == 64 ? 0 : NT == 256 ? 3 : 6) + (LW - 1), lds, nullptr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((smoother_generic_kernel<NT, LW>), dim3((unsigned)g.a.B), dim3(NT), lds, s, g);
    return hipGetLastError();
}
} // namespace

hipError_t launch_filter_generic(int N, int K, const FilterArgs &a, hipStream_t s)
{
    if (N + K > MK_GENERIC_MAX_STATES || a.sym || a.tape) return hipErrorNotSupported;
    int cap = 0;
    hipError_t e = lds_grant(nullptr, 0, 0, &cap);
    if (e != hipSuccess) return e;
    const size_t base = generic_filter_lds_bytes(N, K), withg = base + (size_t)N * K * sizeof(double);
    const bool gl = withg <= (size_t)cap;
    const size_t lds = gl ? withg : base;
    const int nt = filter_threads(N + K);
    if (nt == 64) return gl ? launch_filter_nt<64, true>(N, K, a, lds, s) : launch_filter_nt<64, false>(N, K, a, lds, s);
    if (nt == 256) return gl ? launch_filter_nt<256, true>(N, K, a, lds, s) : launch_filter_nt<256, false>(N, K, a, lds, s);
    return gl ? launch_filter_nt<1024, true>(N, K, a, lds, s) : launch_filter_nt<1024, false>(N, K, a, lds, s);
}

hipError_t launch_smoother_generic(const GenericSmootherArgs &g, hipStream_t s)
{
    const int n = g.N + g.K;
    if (n > MK_GENERIC_MAX_STATES || g.a.sym || g.a.tape || !g.ws) return hipErrorNotSupported;
    int cap = 0;
    hipError_t e = lds_grant(nullptr, 0, 0, &cap);
    if (e != hipSuccess) return e;
    const size_t vec = (size_t)((6 * n + 1) & ~1) * sizeof(double), mat = (size_t)n * n * sizeof(double);
    const int lw = vec + 4 * mat <= (size_t)cap ? 3 : vec + 2 * mat <= (size_t)cap ? 2 : 1;   // X alone: 134 KiB at n = 128
    const size_t lds = vec + (lw == 3 ? 4 : lw) * mat;
    const int nt = smoother_threads(n);
    if (nt == 64) {
        if (lw == 3) return launch_smoother_nt<64, 3>(g, lds, s);
        return lw == 2 ? launch_smoother_nt<64, 2>(g, lds, s) : launch_smoother_nt<64, 1>(g, lds, s);
    }
    if (nt == 256) {
        if (lw == 3) return launch_smoother_nt<256, 3>(g, lds, s);
        return lw == 2 ? launch_smoother_nt<256, 2>(g, lds, s) : launch_smoother_nt<256, 1>(g, lds, s);
    }
    if (lw == 3) return launch_smoother_nt<1024, 3>(g, lds, s);
    return lw == 2 ? launch_smoother_nt<1024, 2>(g, lds, s) : launch_smoother_nt<1024, 1>(g, lds, s);
}

} // namespace mk
