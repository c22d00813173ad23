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

// ---------------------------------------------------------------- cross-lane broadcast
template <int G>
struct Group;

// four models per wavefront, one per 16-lane DPP row: broadcast lane J of each row to its row.
template <>
struct Group<16> {
    template <int J>
    static __device__ __forceinline__ double bcast(double v)
    {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + J, 0xf, 0xf, false); // row_newbcast:J
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + J, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
};

// one model per wavefront: lane J -> SGPR pair.
template <>
struct Group<64> {
    template <int J>
    static __device__ __forceinline__ double bcast(double v)
    {
        int lo = __builtin_amdgcn_readlane(__double2loint(v), J);
        int hi = __builtin_amdgcn_readlane(__double2hiint(v), J);
        return __hiloint2double(hi, lo);
    }
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

constexpr double kLn2 = 0.693147180559945309417232121458;
constexpr double kLog2Pi = 1.837877066409345483560659472811; // log(2*pi)

// =====================================================================================
// Sequential-processing Kalman filter + -2 log L            (kalmanfilter.py:236-400, 550-567)
// =====================================================================================
template <int N, int K, int G>
__global__ void __launch_bounds__(256) filter_kernel(FilterArgs a)
{
    constexpr int n = N + K;
    static_assert(n <= G, "state dimension must fit the lane group");
    using Gp = Group<G>;
    constexpr int GPB = 256 / G; // models per 256-thread workgroup
    const int lane = threadIdx.x % G;
    long inst = (long)blockIdx.x * GPB + threadIdx.x / G;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1; // keep every lane executing (DPP rows stay uniform); stores are masked
    const long rec = inst % a.R;
    const bool rowok = lane < n;
    const int r = rowok ? lane : n - 1;
    const bool st = live && rowok;
    const bool lead = live && lane == 0;
    const long T = a.T;

    // per-model constants: lane r holds phi_r, q_r; lane j < N holds loadings[j,:] and obsvar[j]
    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    double phic[n]; // diag(Phi), replicated in every lane of the group
    sfor<0, n>(MK_LAMBDA(c) { phic[c] = Gp::template bcast<decltype(c)::value>(phi_r); });
    const int jr = lane < N ? lane : N - 1;
    double gam[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gam[k] = a.loadings[(rec * N + jr) * K + k];
    const double rvar = a.obsvar ? a.obsvar[rec * N + jr] : 0.0;

    // initial state (run_filter defaults, kalmanfilter.py:747-750)
    double x = a.x0 ? a.x0[inst * n + r] : 0.0;
    double P[n];
#pragma unroll
    for (int c = 0; c < n; ++c) P[c] = a.P0 ? a.P0[(inst * n + r) * n + c] : (c == r ? 1.0 : 0.0);

    const double *yp = a.obs + rec * T * N + jr; // lane j streams series j
    double ynext = yp[0];
    double sum_sig = 0.0, sum_det = 0.0;
    long nobs = 0, sc = 0;
    unsigned flags = 0;

    for (long t = 0; t < T; ++t) {
        const double y = ynext;
        if (t + 1 < T) ynext = yp[(t + 1) * N]; // prefetch next step's observation

        // ---- predict (:318-331; Phi diagonal) ----
        x = phi_r * x;
#pragma unroll
        for (int c = 0; c < n; ++c) P[c] = (phi_r * P[c]) * phic[c] + (c == r ? q_r : 0.0);
        const long row_t = (inst * T + t) * n + r;
        if (a.Xp && st) a.Xp[row_t] = x;                 // :332
        if (a.Pp && st) store_row<n>(a.Pp + row_t * n, P); // :333

        // ---- sequential scalar updates (:341-378), observations in ascending series order ----
        double sigma = 0.0, fmant = 1.0;
        int fexp = 0, cnt = 0;
        sfor<0, N>(MK_LAMBDA(jc) {
            constexpr int j = decltype(jc)::value;
            const double yj = Gp::template bcast<j>(y);
            if (isfinite(yj)) { // uniform within the model's lane group (:657 masks NaN and inf)
                double g[K];
                sfor<0, K>(MK_LAMBDA(k) { g[k] = Gp::template bcast<j>(gam[k]); });
                // innovation v = y_j - Z_j x,  Z_j = e_j + sum_k g_k e_{N+k}   (:344-347)
                double zx = Gp::template bcast<j>(x);
                sfor<0, K>(MK_LAMBDA(k) { zx = fma(g[k], Gp::template bcast<N + decltype(k)::value>(x), zx); });
                const double v = yj - zx;
                // d = P Z_j^T : lane r computes d_r from its own row (:349-357)
                double dr = P[j];
                sfor<0, K>(MK_LAMBDA(k) { dr = fma(P[N + decltype(k)::value], g[k], dr); });
                double d[n]; // all-gather d across the group
                sfor<0, n>(MK_LAMBDA(c) { d[c] = Gp::template bcast<decltype(c)::value>(dr); });
                // innovation variance f = R_j + Z_j d   (:359-362)
                double zd = d[j];
                sfor<0, K>(MK_LAMBDA(k) { zd = fma(g[k], d[N + decltype(k)::value], zd); });
                const double f = Gp::template bcast<j>(rvar) + zd;
                const double rf = 1.0 / f;
                const double kr = dr * rf; // Kalman gain element r (:364-366)
#pragma unroll
                for (int c = 0; c < n; ++c) P[c] = fma(-kr, d[c], P[c]); // P -= k k^T f (:368-372)
                x = fma(kr, v, x);                                       // :374-375
                sigma = fma(v * v, rf, sigma);                           // :377
                // detf += log f (:378): accumulate prod f as mantissa * 2^exp, one log per step
                fmant *= f;
                fexp += __builtin_amdgcn_frexp_exp(fmant);
                fmant = __builtin_amdgcn_frexp_mant(fmant);
                if (!(f > 0.0)) flags |= MK_FLAG_NONPOSITIVE_F;
                ++cnt;
            }
        });

        if (cnt > 0) { // :380-382 compressed bookkeeping
            const double detf = fma((double)fexp, kLn2, log(fmant));
            if (a.sigmas && lead) a.sigmas[inst * T + sc] = sigma;
            if (a.detfs && lead) a.detfs[inst * T + sc] = detf;
            if (sc >= a.warmup) { // get_mle: detfs[warmup:], sigmas[warmup:] are COMPRESSED indices (:563-564)
                sum_det += detf;
                sum_sig += sigma;
            }
            ++sc;
        }
        if (t >= a.warmup) nobs += cnt; // observation_count[warmup:] is a TIME index (:565)

        if (a.F && st) a.F[row_t] = x;                   // :389
        if (a.Pf && st) store_row<n>(a.Pf + row_t * n, P); // :390
    }

    // zero tail of the compressed arrays (np.zeros init, :307-308)
    if (live) {
        for (long i = sc + lane; i < T; i += G) {
            if (a.sigmas) a.sigmas[inst * T + i] = 0.0;
            if (a.detfs) a.detfs[inst * T + i] = 0.0;
        }
    }
    if (lead) {
        if (a.mle) a.mle[inst] = ((double)nobs * kLog2Pi + sum_det) + sum_sig; // :566
        if (a.sigmacount) a.sigmacount[inst] = sc;
        if (a.status) a.status[inst] = flags;
    }
}

// =====================================================================================
// RTS smoother                                               (kalmanfilter.py:403-476)
//   Pp[t+1] = Phi Pf[t] Phi + Q and Xp[t+1] = Phi F[t] are recomputed (Phi diagonal), so only
//   F and Pf are re-read.  J = Pf Phi^T Pp^{-1} by Cholesky (Pp is SPD whenever q > 0; the
//   reference's pinv (:455) coincides with the inverse there); lane i solves for ROW i of J.
// =====================================================================================
template <int n, bool IN_LDS>
struct LStore;

template <int n>
struct LStore<n, false> { // replicated lower-triangular factor in registers
    double v[n * (n + 1) / 2];
    __device__ __forceinline__ LStore(double *, int) {}
    template <int C, int J>
    __device__ __forceinline__ void set(double x, double /*own*/, int /*lane*/)
    {
        v[C * (C + 1) / 2 + J] = x;
    }
    template <int C, int J>
    __device__ __forceinline__ double get() const
    {
        return v[C * (C + 1) / 2 + J];
    }
    __device__ __forceinline__ void publish() {}
    static constexpr bool per_lane_write = false;
};

template <int n>
struct LStore<n, true> { // one model per wavefront: factor lives in LDS, read by broadcast
    double *base;
    __device__ __forceinline__ LStore(double *lds, int wave) : base(lds + wave * (n * (n + 1) / 2)) {}
    template <int C, int J>
    __device__ __forceinline__ double get() const
    {
        return base[C * (C + 1) / 2 + J];
    }
    // lane `lane` (> J) writes its own L(lane, J)
    template <int J>
    __device__ __forceinline__ void set_own(double own, int lane)
    {
        if (lane > J && lane < n) base[lane * (lane + 1) / 2 + J] = own;
    }
    __device__ __forceinline__ void publish()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

template <int n, int G>
__global__ void __launch_bounds__(256) smoother_kernel(SmootherArgs a)
{
    static_assert(n <= G, "state dimension must fit the lane group");
    using Gp = Group<G>;
    constexpr int GPB = 256 / G;
    constexpr bool LDSL = (G == 64);
    const int lane = threadIdx.x % G;
    long inst = (long)blockIdx.x * GPB + threadIdx.x / G;
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1;
    const bool rowok = lane < n;
    const int r = rowok ? lane : n - 1;
    const bool st = live && rowok;
    const long T = a.T;

    __shared__ double lds_L[LDSL ? (256 / 64) * (n * (n + 1) / 2) : 1];
    LStore<n, LDSL> L(lds_L, threadIdx.x / 64);

    const double phi_r = a.phi[inst * n + r];
    const double q_r = a.q[inst * n + r];
    double phic[n];
    sfor<0, n>(MK_LAMBDA(c) { phic[c] = Gp::template bcast<decltype(c)::value>(phi_r); });

    // last step: smoothed = filtered (:450-451)
    long row_t = (inst * T + (T - 1)) * n + r;
    double xs = a.F[row_t];
    double Psn[n];
    load_row<n>(a.Pf + row_t * n, Psn);
    if (a.S && st) a.S[row_t] = xs;
    if (a.Ps && st) store_row<n>(a.Ps + row_t * n, Psn);
    unsigned flags = 0;

    double Pf[n], xf = 0.0;
    if (T >= 2) { // software prefetch of the next (earlier) time step
        const long rt = (inst * T + (T - 2)) * n + r;
        load_row<n>(a.Pf + rt * n, Pf);
        xf = a.F[rt];
    }

    for (long t = T - 2; t >= 0; --t) {
        row_t = (inst * T + t) * n + r;
        double Pfc[n];
#pragma unroll
        for (int c = 0; c < n; ++c) Pfc[c] = Pf[c];
        const double xfc = xf;
        if (t >= 1) {
            const long rt = row_t - n;
            load_row<n>(a.Pf + rt * n, Pf);
            xf = a.F[rt];
        }

        // W = Pf Phi (column scaling); A = Pp[t+1] = Phi Pf Phi + Q (row r)
        double W[n], A[n];
#pragma unroll
        for (int c = 0; c < n; ++c) {
            W[c] = Pfc[c] * phic[c];
            A[c] = fma(phi_r, W[c], (c == r ? q_r : 0.0));
        }

        // ---- Cholesky A = L L^T, right-looking; every lane keeps the whole factor ----
        double inv[n];
        sfor<0, n>(MK_LAMBDA(jc) {
            constexpr int j = decltype(jc)::value;
            const double piv = Gp::template bcast<j>(A[j]);
            if (!(piv > 0.0)) flags |= MK_FLAG_NOT_SPD;
            const double ij = 1.0 / sqrt(piv);
            inv[j] = ij;
            const double lr = A[j] * ij; // own element of column j: L(r, j) (A symmetric)
            if constexpr (LDSL) L.template set_own<j>(lr, lane);
            sfor<j + 1, n>(MK_LAMBDA(cc) {
                constexpr int c = decltype(cc)::value;
                const double lc = Gp::template bcast<j>(A[c]) * ij; // L(c, j) replicated
                if constexpr (!LDSL) L.template set<c, j>(lc, lr, lane);
                A[c] = fma(-lr, lc, A[c]);
            });
        });
        L.publish();

        // ---- lane i solves A z = W_i  (row i of J = Pf Phi^T A^{-1}, :458-460) ----
        double z[n];
#pragma unroll
        for (int c = 0; c < n; ++c) z[c] = W[c];
        sfor<0, n>(MK_LAMBDA(kc) { // forward: L y = b
            constexpr int k = decltype(kc)::value;
            z[k] *= inv[k];
            sfor<k + 1, n>(MK_LAMBDA(cc) {
                constexpr int c = decltype(cc)::value;
                z[c] = fma(-L.template get<c, k>(), z[k], z[c]);
            });
        });
        sfor_down<0, n>(MK_LAMBDA(kc) { // backward: L^T z = y
            constexpr int k = decltype(kc)::value;
            z[k] *= inv[k];
            sfor<0, k>(MK_LAMBDA(cc) {
                constexpr int c = decltype(cc)::value;
                z[c] = fma(-L.template get<k, c>(), z[k], z[c]);
            });
        });
        // z = J[r, :]

        // ---- smoothed mean (:461-464): xs[t] = F[t] + J (xs[t+1] - Phi F[t]) ----
        const double delta = xs - phi_r * xfc;
        double acc = xfc;
        sfor<0, n>(MK_LAMBDA(c) {
            acc = fma(z[decltype(c)::value], Gp::template bcast<decltype(c)::value>(delta), acc);
        });
        xs = acc;

        // ---- smoothed covariance (:465-474): Ps[t] = Pf[t] + J (Ps[t+1] - Pp[t+1]) J^T ----
        double D[n];
#pragma unroll
        for (int c = 0; c < n; ++c) D[c] = Psn[c] - fma(phi_r, W[c], (c == r ? q_r : 0.0));
        double V[n]; // V = J D (row r)
#pragma unroll
        for (int c = 0; c < n; ++c) V[c] = 0.0;
        sfor<0, n>(MK_LAMBDA(kc) {
            constexpr int k = decltype(kc)::value;
            sfor<0, n>(MK_LAMBDA(cc) {
                constexpr int c = decltype(cc)::value;
                V[c] = fma(z[k], Gp::template bcast<k>(D[c]), V[c]);
            });
        });
        sfor<0, n>(MK_LAMBDA(cc) { // Ps[r, c] = Pf[r, c] + V[r, :] . J[c, :]
            constexpr int c = decltype(cc)::value;
            double s = Pfc[c];
            sfor<0, n>(MK_LAMBDA(kc) {
                constexpr int k = decltype(kc)::value;
                s = fma(V[k], Gp::template bcast<c>(z[k]), s);
            });
            Psn[c] = s;
        });
        L.publish(); // all reads of L done before the next step overwrites it

        if (a.S && st) a.S[row_t] = xs;
        if (a.Ps && st) store_row<n>(a.Ps + row_t * n, Psn);
    }
    if (a.status && live && lane == 0 && flags) atomicOr(a.status + inst, flags);
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
    hipLaunchKernelGGL((filter_kernel<N, K, G>), dim3(grid), dim3(256), 0, s, a);
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
