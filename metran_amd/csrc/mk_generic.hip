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
//   mapping      one model per WORKGROUP of 256 threads (four wavefronts), time sequential, parallel over the matrix elements
//   filter       the covariance P (n x n) lives in LDS (dynamic: n^2 + 5n + N doubles; 160 KiB of LDS per workgroup bound n at
//                MK_GENERIC_MAX_STATES = 128), Z = [I | G] is exploited as in the fast kernels (d = P z' costs 1 + K terms per row)
//   smoother     the reference's RTS recursion with an LDL^T of the predicted covariance in place of its pinv (same contract as
//                the fast kernels: positive pivots inverted, a pivot <= 0 dropped with MK_FLAG_RANK_DEFICIENT, < -1e-8 is
//                MK_FLAG_NOT_SPD); its five n x n work matrices live in a global workspace (L2-resident for one workgroup)
//   layouts      dense arrays or full-square packed records (mk_outputs.record_stride), both time orders; no packed-symmetric
//                records, no tape
// The smoother also serves kalmansmoother's literal 5-argument form (mk_smooth_dense: the CALLER's predicted moments are used,
// not recomputed -- kalmanfilter.py:453-474 reads predicted_state_means / predicted_state_covariances as they are handed in).
#include <hip/hip_runtime.h>

#include <mutex>

#include "mk_generic.h"

namespace mk {

namespace {
constexpr int GNT = 256;                 // threads per workgroup
constexpr double kGenLog2Pi = 1.8378770664093454835606594728112;

__device__ __forceinline__ long blk_index(long inst, long t, long bs, long ts) { return inst * bs + t * ts; }
} // namespace

// ---------------------------------------------------------------------------------------------------------------- filter
__global__ void __launch_bounds__(GNT) filter_generic_kernel(FilterArgs a, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int n = N + K, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double *P = gsm;                     // [n][n]
    double *x = P + (long)n * n;         // [n]
    double *d = x + n;                   // [n]  P z_j'
    double *phi = d + n;                 // [n]
    double *qv = phi + n;                // [n]
    double *yv = qv + n;                 // [N]  observations of the step
    const long inst = blockIdx.x, rec = inst % a.R;
    const double *G = a.loadings + rec * (long)N * K;
    const double *Rv = a.obsvar ? a.obsvar + rec * N : nullptr;
    const long T = a.T;
    for (int i = tid; i < n; i += GNT) {
        phi[i] = a.phi[inst * n + i];
        qv[i] = a.q[inst * n + i];
        x[i] = a.x0 ? a.x0[inst * n + i] : 0.0;                     // run_filter defaults (:747-750)
    }
    for (int r = wave; r < n; r += GNT / 64)
        for (int c = lane; c < n; c += 64) P[r * n + c] = a.P0 ? a.P0[(inst * n + r) * n + c] : (r == c ? 1.0 : 0.0);
    __syncthreads();

    const long SF = a.rs > 0 ? a.rs : n;              // doubles between the (b, t) blocks of the mean arrays ...
    const long SP = a.rs > 0 ? a.rs : (long)n * n;    // ... and of the covariance arrays (records: d_Pf = d_F + n, same stride)
    auto emit = [&](double *mean, double *cov, long blk) {
        if (mean)
            for (int i = tid; i < n; i += GNT) mean[blk * SF + i] = x[i];
        if (cov)
            for (int i = tid; i < n * n; i += GNT) cov[blk * SP + i] = P[i];
    };

    double sum_sig = 0.0, sum_det = 0.0;   // every thread keeps the same scalars
    bool bad_f = false;
    long nobs = 0, sc = 0;
    for (long t = 0; t < T; ++t) {
        const long blk = blk_index(inst, t, a.bs, a.ts);
        for (int j = tid; j < N; j += GNT) yv[j] = a.obs[(rec * a.obs_bs + t * a.obs_ts) * N + j];
        // ---- predict (:318-331; Phi diagonal)
        for (int i = tid; i < n; i += GNT) x[i] = phi[i] * x[i];
        for (int r = wave; r < n; r += GNT / 64) {
            const double pr = phi[r];
            for (int c = lane; c < n; c += 64) P[r * n + c] = fma(P[r * n + c] * pr, phi[c], r == c ? qv[r] : 0.0);
        }
        __syncthreads();
        emit(a.Xp, a.Pp, blk);                         // :332-333
        // ---- sequential scalar updates (:341-378), ascending series order
        double sigma = 0.0, detf = 0.0;
        int cnt = 0;
        for (int j = 0; j < N; ++j) {
            const double y = yv[j];
            if (!isfinite(y)) continue;                // NaN / inf = missing (:657); the same for every thread
            ++cnt;
            for (int r = tid; r < n; r += GNT) {       // d = P z_j' (:349-357): Z = [I | G]
                double s = P[r * n + j];
                for (int k = 0; k < K; ++k) s = fma(P[r * n + N + k], G[j * K + k], s);
                d[r] = s;
            }
            __syncthreads();
            double f = (Rv ? Rv[j] : 0.0) + d[j], v = y - x[j];   // :344-347, :359-362
            for (int k = 0; k < K; ++k) {
                f = fma(G[j * K + k], d[N + k], f);
                v = fma(-G[j * K + k], x[N + k], v);
            }
            __syncthreads();                           // every thread has read x and d before x moves
            const double rf = 1.0 / f;
            for (int i = tid; i < n; i += GNT) x[i] = fma(d[i] * rf, v, x[i]);      // :374-375
            for (int r = wave; r < n; r += GNT / 64) { // P -= k k' f (:368-372)
                const double kr = d[r] * rf;
                for (int c = lane; c < n; c += 64) P[r * n + c] = fma(-kr, d[c], P[r * n + c]);
            }
            sigma = fma(v * v, rf, sigma);             // :377
            detf += log(f);                            // :378
            bad_f = bad_f || !(f > 0.0);                // f <= 0 or NaN
            __syncthreads();
        }
        if (cnt > 0) {                                 // :380-382, compressed index sc
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
    for (long i = sc + tid; i < T; i += GNT) {        // zero tail of the compressed arrays (np.zeros, :307-308)
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
__global__ void __launch_bounds__(GNT) smoother_generic_kernel(GenericSmootherArgs g)
{
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const SmootherArgs &a = g.a;
    const int N = g.N, K = g.K, n = N + K, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long nn = (long)n * n;
    double *xs = gsm;          // smoothed mean of step t+1
    double *xsn = xs + n;      // ... of step t
    double *delta = xsn + n;   // xs[t+1] - Xp[t+1]
    double *phi = delta + n;
    double *qv = phi + n;
    double *dinv = qv + n;     // D^+ of the factorisation
    const long inst = blockIdx.x, rec = inst % (a.R > 0 ? a.R : 1);
    double *A = g.ws + inst * 5 * nn;   // Pp[t+1], then its L (strictly lower part) -- row-major
    double *X = A + nn;                 // right-hand sides Phi Pf[t], then J' (X[c][i] = J[i][c])
    double *Dm = X + nn;                // Ps[t+1] - Pp[t+1]
    double *Psn = Dm + nn;              // smoothed covariance of step t+1, then of step t
    double *V = Psn + nn;               // Dm J'
    const long T = a.T;
    const long SF = a.rs > 0 ? a.rs : n, SP = a.rs > 0 ? a.rs : nn;
    const double *G = a.loadings ? a.loadings + rec * (long)N * K : nullptr;
    for (int i = tid; i < n; i += GNT) {
        phi[i] = a.phi[inst * n + i];
        qv[i] = a.q ? a.q[inst * n + i] : 0.0;
    }
    unsigned flags = 0;

    // outputs of one step from (xs, Psn): smoothed records / dense arrays, VAR_ONLY, fused projection (:569-603)
    auto emit = [&](long blk) {
        if (a.S)
            for (int i = tid; i < n; i += GNT) a.S[blk * SF + i] = xs[i];
        if (a.Ps)
            for (long i = tid; i < nn; i += GNT) a.Ps[blk * SP + i] = Psn[i];
        if (a.state_means)
            for (int i = tid; i < n; i += GNT) a.state_means[blk * n + i] = xs[i];
        if (a.state_vars)
            for (int i = tid; i < n; i += GNT) a.state_vars[blk * n + i] = Psn[(long)i * n + i];
        if ((a.sim_means || a.sim_vars) && G) {
            for (int j = tid; j < N; j += GNT) {
                const double sc = a.scale ? a.scale[rec * N + j] : 1.0, off = a.offset ? a.offset[rec * N + j] : 0.0;
                double m = xs[j], var = Psn[(long)j * n + j];
                for (int k = 0; k < K; ++k) {
                    const double gk = G[j * K + k];
                    m = fma(gk, xs[N + k], m);
                    double row = Psn[(long)j * n + N + k] + Psn[(long)(N + k) * n + j];   // (j, N+k) + (N+k, j)
                    for (int k2 = 0; k2 < K; ++k2) row = fma(G[j * K + k2], Psn[(long)(N + k) * n + N + k2], row);
                    var = fma(gk, row, var);
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
        for (int i = tid; i < n; i += GNT) xs[i] = a.F[blk * SF + i];
        for (long i = tid; i < nn; i += GNT) Psn[i] = a.Pf[blk * SP + i];
        __syncthreads();
        emit(blk);
        __syncthreads();
    }
    for (long t = T - 2; t >= 0; --t) {
        const long blk = blk_index(inst, t, a.bs, a.ts), blk1 = blk_index(inst, t + 1, a.bs, a.ts);
        const double *Pft = a.Pf + blk * SP, *Ft = a.F + blk * SF;
        // A = Pp[t+1] (the caller's, or Phi Pf[t] Phi + Q), Dm = Ps[t+1] - Pp[t+1], X = Phi Pf[t] (row r scaled: Pf symmetric)
        for (int r = wave; r < n; r += GNT / 64) {
            const double pr = phi[r];
            for (int c = lane; c < n; c += 64) {
                const long i = (long)r * n + c;
                const double pf = Pft[i];
                const double pp = g.Pp ? g.Pp[blk1 * nn + i] : fma(pr * pf, phi[c], r == c ? qv[r] : 0.0);
                A[i] = pp;
                Dm[i] = Psn[i] - pp;
                X[i] = pr * pf;
            }
        }
        for (int i = tid; i < n; i += GNT) delta[i] = xs[i] - (g.Xp ? g.Xp[blk1 * n + i] : phi[i] * Ft[i]);
        __syncthreads();
        // ---- LDL' of A in place, right-looking, one barrier per pivot: [pivot k] [scale column k-1] [trailing update with column k]
        double dprev = 0.0;
        for (int k = 0; k < n; ++k) {
            const double piv = A[(long)k * n + k];
            double di = 0.0;
            if (piv > 0.0) di = 1.0 / piv;
            else {
                flags |= MK_FLAG_RANK_DEFICIENT;       // null direction dropped (the reference's pinv, :455)
                if (piv < -1e-8 || !(piv == piv)) flags |= MK_FLAG_NOT_SPD;
            }
            if (tid == 0) dinv[k] = di;
            if (k > 0)
                for (int i = k + tid; i < n; i += GNT) A[(long)i * n + (k - 1)] *= dprev;     // L(i, k-1), i >= k (row k-1.. done)
            for (int i = k + 1 + wave; i < n; i += GNT / 64) {
                const double lik = A[(long)i * n + k] * di;
                for (int j = k + 1 + lane; j <= i; j += 64) A[(long)i * n + j] = fma(-lik, A[(long)j * n + k], A[(long)i * n + j]);
            }
            dprev = di;
            __syncthreads();
        }
        // (iteration k scaled column k - 1; column n - 1 has no sub-diagonal part)
        // ---- J' = A^-1 (Phi Pf[t]): one right-hand side per thread, forward, D^+, backward (:458-460 with pinv -> LDL')
        for (int c = tid; c < n; c += GNT) {
            for (int i = 0; i < n; ++i) {
                double s = X[(long)i * n + c];
                for (int k = 0; k < i; ++k) s = fma(-A[(long)i * n + k], X[(long)k * n + c], s);
                X[(long)i * n + c] = s;
            }
            for (int i = n - 1; i >= 0; --i) {
                double s = X[(long)i * n + c] * dinv[i];
                for (int k = i + 1; k < n; ++k) s = fma(-A[(long)k * n + i], X[(long)k * n + c], s);
                X[(long)i * n + c] = s;
            }
        }
        __syncthreads();
        // ---- S[t] = F[t] + J delta (:461-464);  V = Dm J'
        for (int i = tid; i < n; i += GNT) {
            double s = Ft[i];
            for (int c = 0; c < n; ++c) s = fma(X[(long)c * n + i], delta[c], s);
            xsn[i] = s;
        }
        for (int r = wave; r < n; r += GNT / 64)
            for (int i = lane; i < n; i += 64) {
                double s = 0.0;
                for (int c = 0; c < n; ++c) s = fma(Dm[(long)r * n + c], X[(long)c * n + i], s);
                V[(long)r * n + i] = s;
            }
        __syncthreads();
        // ---- Ps[t] = Pf[t] + J (Ps[t+1] - Pp[t+1]) J' (:465-474):  Ps[i][j] = Pf[i][j] + sum_r J[i][r] V[r][j]
        for (int i = wave; i < n; i += GNT / 64)
            for (int j = lane; j < n; j += 64) {
                double s = Pft[(long)i * n + j];
                for (int r = 0; r < n; ++r) s = fma(X[(long)r * n + i], V[(long)r * n + j], s);
                Psn[(long)i * n + j] = s;
            }
        for (int i = tid; i < n; i += GNT) xs[i] = xsn[i];
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
    return (n * n + 4 * n + (size_t)N + 2) * sizeof(double);
}
size_t generic_smoother_ws_doubles(long B, int n) { return (size_t)B * 5 * (size_t)n * (size_t)n; }

hipError_t launch_filter_generic(int N, int K, const FilterArgs &a, hipStream_t s)
{
    if (N + K > MK_GENERIC_MAX_STATES || a.sym || a.tape) return hipErrorNotSupported;
    const size_t lds = generic_filter_lds_bytes(N, K);
    // above the 64 KiB default a kernel has to be given its dynamic LDS size explicitly -- PER DEVICE (round-5 advice: the grant
    // used to be remembered per thread, so a thread driving a second GPU skipped the call there and its launch failed), and a
    // model that needs more LDS than the device has is refused here with the shape error instead of a raw launch failure
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    constexpr int MAXDEV = 64;
    static std::mutex grant_mutex;
    static size_t granted[MAXDEV] = {};
    static int lds_cap[MAXDEV] = {};
    std::lock_guard<std::mutex> lock(grant_mutex);
    if (dev < 0 || dev >= MAXDEV) return hipErrorInvalidDevice;
    if (!lds_cap[dev]) {
        e = hipDeviceGetAttribute(&lds_cap[dev], hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
        if (e != hipSuccess) return e;
    }
    if (lds > (size_t)lds_cap[dev]) return hipErrorNotSupported; // MK_ERR_SHAPE at the C ABI: N + K too large for this device's LDS
    if (lds > granted[dev] && lds > 64 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&filter_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        granted[dev] = lds;
    }
    hipLaunchKernelGGL(filter_generic_kernel, dim3((unsigned)a.B), dim3(GNT), lds, s, a, N, K);
    return hipGetLastError();
}

hipError_t launch_smoother_generic(const GenericSmootherArgs &g, hipStream_t s)
{
    const int n = g.N + g.K;
    if (n > MK_GENERIC_MAX_STATES || g.a.sym || g.a.tape || !g.ws) return hipErrorNotSupported;
    const size_t lds = (size_t)(6 * n + 2) * sizeof(double);
    hipLaunchKernelGGL(smoother_generic_kernel, dim3((unsigned)g.a.B), dim3(GNT), lds, s, g);
    return hipGetLastError();
}

} // namespace mk
