// mk_factor.hip -- batched factor analysis: what produces the loadings and the
// number of common factors of every model BEFORE the Kalman filter runs, for thousands of models at once.
// Reference: /root/reference/metran/factoranalysis.py
//   fa_corr_kernel      _get_correlations (:404-418)  pairwise-complete Pearson correlation (DataFrame.corr)
//   fa_analyse_kernel   _get_eigval (:420-460) + _maptest (:220-312) + Kaiser fallback / maxfactors (:66-82)
//                       + the start vector of _minres (:188-197)
//   fa_minres_kernel    _minresfun (:315-347), _minresgrad (:349-373), _get_loadings (:375-401)
//   fa_rotate_kernel    communality normalisation, _rotate (varimax, :121-171), sign convention (:84-108)
// One model per wavefront; every N x N matrix lives in a wave-private LDS block (N <= 64).  The reference
// calls LAPACK (eig / eigh / svd / inv) on N x N matrices; here ONE device routine, a parallel-ordering Jacobi
// eigen-decomposition of a symmetric matrix, serves all of them: the correlation matrix is symmetric (eig),
// svd(M) of the K x K varimax step comes from eigh(M^T M) (R = u vh is the orthogonal polar factor of M),
// and diag(inv(S)) = sum_k V_ik^2 / w_k.  This is not the hot path: no MFMA, no tuning beyond keeping the
// wavefront's lanes busy in the row/column updates.
#include <hip/hip_runtime.h>

#include "mk_internal.h"

namespace mk {

namespace {

__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// Jacobi eigen-decomposition of a symmetric n x n matrix A (LDS, row-major, destroyed: the eigenvalues end up on its
// diagonal) accumulating the eigenvectors in the COLUMNS of V, in the PARALLEL (round-robin) ordering: a sweep is
// m - 1 rounds (m = n rounded up to even) of m / 2 rotations on disjoint index pairs -- the circle method: in round r
// index m-1 meets r, and (r + i) mod (m-1) meets (r - i) mod (m-1), i = 1 .. m/2 - 1 -- so that every pair (p, q) is
// met exactly once per sweep, as in the cyclic ordering round 2 shipped (one rotation at a time, three fences each:
// 496 dependent rotations a sweep at n = 32; 31 rounds here).  Within a round
//   1. lane i < m/2 forms (c, s) of its pair from the CURRENT A(p,p), A(q,q), A(p,q)  (Rutishauser's formulas);
//   2. all lanes apply the rotations to the columns of A and of V (element (k, pair i): lanes of one row k take
//      different pairs: no two lanes touch the same element, consecutive lanes read different columns of one row);
//   3. after a fence, to the rows of A (element (pair i, k): consecutive lanes take consecutive columns k).
// The rotations of a round commute (disjoint planes): the result is J^T A J for their product J.  Converges
// quadratically like the cyclic ordering; the loop stops when the off-diagonal mass is at rounding level.
__device__ void jacobi_eigh(double *A, double *V, int n, int lane)
{
    __shared__ double rot_c[32], rot_s[32];
    __shared__ int rot_p[32], rot_q[32];
    for (int i = lane; i < n * n; i += 64) V[i] = (i / n == i % n) ? 1.0 : 0.0;
    wsync();
    if (n < 2) return;
    const int m = n + (n & 1), hp = m / 2;
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int i = lane; i < n * n; i += 64) {
            const double v = A[i];
            if (i / n == i % n) dia += v * v;
            else off += v * v;
        }
        for (int s = 32; s > 0; s >>= 1) {
            off += __shfl_xor(off, s);
            dia += __shfl_xor(dia, s);
        }
        if (off <= 1e-30 * dia || off == 0.0) break;
        for (int r = 0; r < m - 1; ++r) {
            if (lane < hp) {
                int p, q;
                if (lane == 0) {
                    p = m - 1;
                    q = r;
                } else {
                    p = (r + lane) % (m - 1);
                    q = (r - lane + (m - 1)) % (m - 1);
                }
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                double c = 1.0, sn = 0.0;
                if (q < n) { // (an odd n has one idle index per round)
                    const double apq = A[p * n + q];
                    if (apq != 0.0) {
                        const double app = A[p * n + p], aqq = A[q * n + q];
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0);
                        sn = t * c;
                    }
                } else {
                    q = p; // identity on the idle index: c = 1, s = 0 and both "columns" are the same one
                }
                rot_p[lane] = p;
                rot_q[lane] = q;
                rot_c[lane] = c;
                rot_s[lane] = sn;
            }
            wsync();
            // columns of A and V
            for (int idx = lane; idx < n * hp; idx += 64) {
                const int i = idx % hp, k = idx / hp;
                const int p = rot_p[i], q = rot_q[i];
                const double c = rot_c[i], sn = rot_s[i];
                if (sn != 0.0) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - sn * akq;
                    A[k * n + q] = sn * akp + c * akq;
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - sn * vkq;
                    V[k * n + q] = sn * vkp + c * vkq;
                }
            }
            wsync();
            // rows of A
            for (int idx = lane; idx < n * hp; idx += 64) {
                const int k = idx % n, i = idx / n;
                const int p = rot_p[i], q = rot_q[i];
                const double c = rot_c[i], sn = rot_s[i];
                if (sn != 0.0) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - sn * aqk;
                    A[q * n + k] = sn * apk + c * aqk;
                }
            }
            wsync();
        }
    }
    wsync();
}

// rank of eigenvalue k in DESCENDING order (ties broken by index): position it is sorted to
__device__ __forceinline__ int rank_desc(const double *A, int n, int k)
{
    const double wk = A[k * n + k];
    int rnk = 0;
    for (int j = 0; j < n; ++j) {
        const double wj = A[j * n + j];
        rnk += (wj > wk) || (wj == wk && j < k);
    }
    return rnk;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------
// pandas DataFrame.corr (pearson, min_periods = 1): for every pair of series the rows where BOTH are present
// (NaN = missing; the daily-grid padding of Metran.oseries, metran.py:571), two passes (means, then centred
// sums).  One block per record, one thread per pair (i <= j).  corr [R,N,N] symmetric, diagonal 1 (NaN when a
// series has fewer than two common observations or zero variance -- pandas gives NaN there as well).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fa_corr_kernel(long R, long T, int N, long bs, long ts, const double *obs, double *corr)
{
    const long r = blockIdx.x;
    const double *src = obs + r * bs * N;
    const long step = ts * N;
    const int P = N * (N + 1) / 2;
    for (int pr = threadIdx.x; pr < P; pr += blockDim.x) {
        int i = 0, rem = pr; // pair index -> (i, j), i <= j, rows of the upper triangle
        while (rem >= N - i) {
            rem -= N - i;
            ++i;
        }
        const int j = i + rem;
        long cnt = 0;
        double sx = 0.0, sy = 0.0;
        for (long t = 0; t < T; ++t) {
            const double x = src[t * step + i], y = src[t * step + j];
            if (x == x && y == y) {
                sx += x;
                sy += y;
                ++cnt;
            }
        }
        double res = __builtin_nan("");
        if (cnt > 0) {
            const double mx = sx / (double)cnt, my = sy / (double)cnt;
            double sxx = 0.0, syy = 0.0, sxy = 0.0;
            for (long t = 0; t < T; ++t) {
                const double x = src[t * step + i], y = src[t * step + j];
                if (x == x && y == y) {
                    const double dx = x - mx, dy = y - my;
                    sxx += dx * dx;
                    syy += dy * dy;
                    sxy += dx * dy;
                }
            }
            const double div = sqrt(sxx * syy);
            if (div != 0.0) res = sxy / div;
        }
        corr[(r * N + i) * N + j] = res;
        corr[(r * N + j) * N + i] = res;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Per model: eigenvalues (descending, negatives clipped to 0, :448-457), Velicer's MAP test (original and
// revised, :220-312), number of factors (MAP, Kaiser criterion when MAP says 0, capped by maxfactors; :66-82)
// and the start vector of minres psi0 = clip(1 / diag(inv(S)), 0.005, 1)  (:188-203: start = diag(s) - ssmc,
// ssmc = 1 - 1/diag(inv(s)); the (0.005, 1) bounds are L-BFGS-B's).  status: 1 = the correlation matrix has a
// NaN / is not invertible (the reference returns no factors there).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) fa_analyse_kernel(long B, int N, long maxfactors, const double *corr, double *eigval,
                                                        long long *nfact, long long *nfact_map, long long *nfact_map4,
                                                        double *psi0, unsigned *status)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = N, lane = threadIdx.x;
    const long b = blockIdx.x;
    double *A = lds, *V = A + n * n, *C = V + n * n, *w = C + n * n; // C: partial covariance; w: sorted eigenvalues
    int *perm = reinterpret_cast<int *>(w + n); // n ints, then (below) the MAP criteria v2[n], v4[n]
    const double *S = corr + b * n * n;
    bool bad = false;
    for (int i = lane; i < n * n; i += 64) {
        const double v = S[i];
        A[i] = v;
        C[i] = v;
        bad |= !(v == v);
    }
    bad = __any(bad);
    wsync();
    jacobi_eigh(A, V, n, lane);
    for (int k = lane; k < n; k += 64) perm[rank_desc(A, n, k)] = k; // perm[s] = column holding the s-th largest
    wsync();
    for (int s = lane; s < n; s += 64) {
        double v = A[perm[s] * n + perm[s]];
        if (v < 0.0) v = 0.0; // :456
        w[s] = v;
        if (eigval) eigval[b * n + s] = v;
    }
    wsync();
    // diag(inv(S))_i = sum_k V_ik^2 / w_k with the UNCLIPPED eigenvalues
    double wmin = 1.0, wmax = 0.0;
    for (int k = 0; k < n; ++k) {
        wmin = fmin(wmin, A[k * n + k]);
        wmax = fmax(wmax, A[k * n + k]);
    }
    // numpy.linalg.inv raises on a singular matrix -> _minres returns None (:199-200).  LAPACK meets an exactly zero pivot
    // when two series are collinear; the Jacobi eigenvalue of that direction is zero only up to rounding (either sign),
    // so "singular" is a smallest eigenvalue below 1e-14 of the largest
    if (!(wmin > 1e-14 * wmax)) bad = true;
    for (int i = lane; i < n; i += 64) {
        double d = 0.0;
        for (int k = 0; k < n; ++k) d += V[i * n + k] * V[i * n + k] / A[k * n + k];
        double p = S[i * n + i] - (1.0 - 1.0 / d); // start = diag(s) - ssmc
        p = fmin(fmax(p, 0.005), 1.0);
        if (psi0) psi0[b * n + i] = p;
    }
    // ---- Velicer's MAP test.  v_k = (sum(pr^2) - n) / (n (n-1)) with pr the partial correlations after removing
    // the first k principal components (columns of V scaled by sqrt(w)); likewise with pr^4 (revised test).
    // The reference stores v_k with np.put(fm, [k, 1], v) (:282-294) -- FLAT indices -- and reads fm[s, 1] back
    // (:300-311): fm[0,1] = v_{n-1} (the last value written), fm[s,1] = v_{2s+1} where 2s+1 <= n-1, else the
    // initial arange value s.  Restated as it behaves: only criteria of an ODD number of removed components
    // compete, against v_{n-1} (= 1 up to rounding: the residual has rank one).  For n <= 3 rounding alone
    // decides between 0 and 1 there; the factor count that follows (Kaiser criterion when 0) is the same. ----
    const double denom = (double)n * (double)(n - 1);
    double *v2 = w + n + (n + 1) / 2; // behind w and perm (n ints occupy (n+1)/2 doubles)
    double *v4 = v2 + n;
    double s2 = 0.0, s4 = 0.0;
    for (int i = lane; i < n * n; i += 64) {
        const double v = C[i] * C[i];
        s2 += v;
        s4 += v * v;
    }
    for (int s = 32; s > 0; s >>= 1) {
        s2 += __shfl_xor(s2, s);
        s4 += __shfl_xor(s4, s);
    }
    if (lane == 0) {
        v2[0] = (s2 - n) / denom;
        v4[0] = (s4 - n) / denom;
    }
    bool early = false; // "exit function with nfacts=1 if diag partcov contains negatives" (:277-279)
    for (int m = 0; m < n - 1; ++m) {
        const int col = perm[m];
        const double sw = w[m]; // eigvec column scaled by sqrt(eigval): a a^T = w v v^T
        wsync();
        for (int i = lane; i < n * n; i += 64) C[i] -= sw * V[(i / n) * n + col] * V[(i % n) * n + col];
        wsync();
        double dmin = 1.0;
        for (int k = 0; k < n; ++k) dmin = fmin(dmin, C[k * n + k]);
        if (dmin < 0.0) {
            early = true;
            break;
        }
        double t2 = 0.0, t4 = 0.0;
        for (int i = lane; i < n * n; i += 64) {
            const double pr = C[i] / sqrt(C[(i / n) * n + i / n] * C[(i % n) * n + i % n]);
            const double v = pr * pr;
            t2 += v;
            t4 += v * v;
        }
        for (int s = 32; s > 0; s >>= 1) {
            t2 += __shfl_xor(t2, s);
            t4 += __shfl_xor(t4, s);
        }
        if (lane == 0) {
            v2[m + 1] = (t2 - n) / denom;
            v4[m + 1] = (t4 - n) / denom;
        }
    }
    wsync();
    long nf_map = 0, nf_map4 = 0;
    if (early) {
        nf_map = nf_map4 = 1;
    } else {
        double best = v2[n - 1], best4 = v4[n - 1]; // fm[0,1]: NaN compares false, as in the reference
        for (int s = 1; s < n; ++s) {
            const double c2 = 2 * s + 1 <= n - 1 ? v2[2 * s + 1] : (double)s;
            const double c4 = 2 * s + 1 <= n - 1 ? v4[2 * s + 1] : (double)s;
            if (c2 < best) {
                best = c2;
                nf_map = s;
            }
            if (c4 < best4) {
                best4 = c4;
                nf_map4 = s;
            }
        }
    }
    if (lane == 0) {
        long nf = nf_map;
        if (nf == 0) { // Kaiser criterion (:72-77)
            for (int s = 0; s < n; ++s) nf += w[s] > 1.0;
        }
        if (maxfactors > 0 && nf > maxfactors) nf = maxfactors;
        if (bad) nf = 0;
        if (nfact) nfact[b] = nf;
        if (nfact_map) nfact_map[b] = nf_map;
        if (nfact_map4) nfact_map4[b] = nf_map4;
        if (status) status[b] = bad ? 1u : 0u;
    }
}

// ---------------------------------------------------------------------------------------------------------
// minres objective, its "jacobian" and the loadings for B parameter vectors psi (instance b uses correlation
// matrix b % R and its factor count).  Restated exactly as the reference has them -- including that the
// OBJECTIVE (:334-347) builds its model from the nf SMALLEST eigenpairs of S with diagonal 1 - psi (numpy.eigh
// is ascending and the code takes [:nf]) while the JACOBIAN (:370-372) is that of the proper minres fit,
// diag(L L^T + diag(psi) - S) / psi^2 with L = _get_loadings(psi) (:396-400).
// _get_loadings takes eigvec[:, :nf] of numpy.linalg.eig(psi^-1/2 S psi^-1/2) -- LAPACK dgeev's order, which is NOT
// sorted (for a quarter of 20- and 32-series two-factor models the first nf pairs are not the nf largest).  Which
// pairs those are is an INPUT here: order [B,KMAX], the rank (0 = largest eigenvalue) of the pair that becomes
// column f; the host obtains it from the very routine the reference calls (metran_amd/factoranalysis.py::
// eig_order), the arithmetic stays here.  order == nullptr: ranks 0 .. nf-1 (the nf largest, descending).
// loadings [B,N,KMAX], columns >= nf zero.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) fa_minres_kernel(long B, long R, int N, int KMAX, const double *corr,
                                                       const long long *nfact, const double *psi,
                                                       const long long *order, double *fval, double *grad,
                                                       double *loadings)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = N, lane = threadIdx.x;
    const long b = blockIdx.x, rec = b % R;
    double *A = lds, *V = A + n * n, *w = V + n * n, *ld = w + n; // ld: loadings [n, KMAX]
    int *perm = reinterpret_cast<int *>(ld + n * KMAX);
    const double *S = corr + rec * n * n;
    const double *ps = psi + b * n;
    long nf = nfact[rec];
    if (nf > KMAX) nf = KMAX;
    // ---- _get_loadings: eig of sstar = psi^-1/2 S psi^-1/2, the pairs of rank order[b][0 .. nf-1] ----
    for (int i = lane; i < n * n; i += 64) A[i] = S[i] / sqrt(ps[i / n] * ps[i % n]);
    wsync();
    jacobi_eigh(A, V, n, lane);
    for (int k = lane; k < n; k += 64) perm[rank_desc(A, n, k)] = k;
    wsync();
    for (int i = lane; i < n * KMAX; i += 64) {
        const int row = i / KMAX, f = i % KMAX;
        double v = 0.0;
        if (f < nf) {
            long rk = order ? order[b * KMAX + f] : f;
            if (rk < 0 || rk >= n) rk = f; // never index outside perm; the C ABI documents the valid range
            const int col = perm[rk];
            const double ev = A[col * n + col] - 1.0;
            v = sqrt(ps[row]) * V[row * n + col] * sqrt(ev > 0.0 ? ev : 0.0);
        }
        ld[i] = v;
        if (loadings) loadings[b * n * KMAX + i] = v;
    }
    wsync();
    if (grad)
        for (int i = lane; i < n; i += 64) {
            double g = ps[i] - S[i * n + i];
            for (int f = 0; f < nf; ++f) g += ld[i * KMAX + f] * ld[i * KMAX + f];
            grad[b * n + i] = g / (ps[i] * ps[i]);
        }
    if (!fval) return;
    // ---- _minresfun: s2 = S with diagonal 1 - psi; the nf SMALLEST eigenpairs (eigh ascending, [:nf]) ----
    wsync();
    for (int i = lane; i < n * n; i += 64) A[i] = (i / n == i % n) ? 1.0 - ps[i / n] : S[i];
    wsync();
    jacobi_eigh(A, V, n, lane);
    for (int k = lane; k < n; k += 64) perm[n - 1 - rank_desc(A, n, k)] = k; // perm[s] = s-th SMALLEST
    wsync();
    for (int s = lane; s < n; s += 64) {
        double v = A[perm[s] * n + perm[s]];
        if (v < 2.220446049250313e-16) v = 100.0 * 2.220446049250313e-16; // :336-337
        w[s] = v;
    }
    wsync();
    double acc = 0.0;
    for (int i = lane; i < n * n; i += 64) {
        const int row = i / n, col = i % n;
        if (row == col) continue;
        double model = 0.0;
        if (nf > 1) {
            for (int f = 0; f < nf; ++f) model += w[f] * V[row * n + perm[f]] * V[col * n + perm[f]];
        } else {
            model = w[0]; // :341-343: a 1-D loading vector, and np.dot(l, l.T) of a 1-D array is the SCALAR l.l = w[0]
        }
        const double rsd = S[i] - model;
        acc += rsd * rsd;
    }
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    if (lane == 0) fval[b] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// What FactorAnalysis.solve does to the minres loadings (:84-108): when nf > 1 rows are normalised by their
// communality, rotated (varimax: _rotate with gamma = 1, maxiter = 20, tol = 1e-6, :146-171) and scaled back;
// columns whose sum is negative change sign.  In place on loadings [B,N,KMAX].
// svd(M) of the K x K step: M^T M = W diag(s^2) W^T  =>  R = u vh = M W diag(1/s) W^T, d = sum(s).
// Where a singular value is exactly 0 the polar factor is not unique; numpy's u vh completes it with LAPACK's bases of
// the null spaces.  Here: the identity on the null space (+ w_j w_j^T), which is what LAPACK returns in the two ways
// an EXACT zero arises -- M = 0 (u = vh = I: a loading matrix whose only non-zero column has unit-normalised entries
// +-1, fixture mv1 of factor_multi.npz) and zero columns of the loadings (zero rows AND columns of M: the completion
// only ever multiplies those zero columns).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) fa_rotate_kernel(long B, int N, int KMAX, const long long *nfact, double *loadings,
                                                       double gamma, int maxiter, double tol)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x, p = N;
    const long b = blockIdx.x;
    int k = (int)nfact[b];
    if (k > KMAX) k = KMAX;
    if (k < 1) return;
    double *Ld = loadings + b * N * KMAX;
    double *phi = lds;              // [p, k] normalised loadings
    double *Lam = phi + p * KMAX;   // [p, k] phi R
    double *Rm = Lam + p * KMAX;    // [k, k]
    double *M = Rm + KMAX * KMAX;   // [k, k]
    double *A = M + KMAX * KMAX;    // [k, k] M^T M
    double *W = A + KMAX * KMAX;    // [k, k] eigenvectors
    double *comm = W + KMAX * KMAX; // [p]
    double *cs = comm + p;          // [k] column sums of Lam^2
    if (k > 1) {
        for (int i = lane; i < p; i += 64) {
            double c = 0.0;
            for (int j = 0; j < k; ++j) c += Ld[i * KMAX + j] * Ld[i * KMAX + j];
            comm[i] = c;
            const double sc = sqrt(c);
            for (int j = 0; j < k; ++j) phi[i * k + j] = Ld[i * KMAX + j] / sc;
        }
        for (int i = lane; i < k * k; i += 64) Rm[i] = (i / k == i % k) ? 1.0 : 0.0;
        wsync();
        double d = 0.0;
        for (int it = 0; it < maxiter; ++it) {
            const double d_old = d;
            for (int i = lane; i < p * k; i += 64) {
                double v = 0.0;
                for (int j = 0; j < k; ++j) v += phi[(i / k) * k + j] * Rm[j * k + i % k];
                Lam[i] = v;
            }
            wsync();
            for (int j = lane; j < k; j += 64) {
                double c = 0.0;
                for (int i = 0; i < p; ++i) c += Lam[i * k + j] * Lam[i * k + j];
                cs[j] = c;
            }
            wsync();
            for (int i = lane; i < k * k; i += 64) { // M = phi^T (Lam^3 - (gamma/p) Lam diag(colsum(Lam^2)))
                const int a = i / k, c = i % k;
                double v = 0.0;
                for (int r = 0; r < p; ++r) {
                    const double l = Lam[r * k + c];
                    v += phi[r * k + a] * (l * l * l - (gamma / p) * l * cs[c]);
                }
                M[i] = v;
            }
            wsync();
            for (int i = lane; i < k * k; i += 64) {
                double v = 0.0;
                for (int r = 0; r < k; ++r) v += M[r * k + i / k] * M[r * k + i % k];
                A[i] = v;
            }
            wsync();
            jacobi_eigh(A, W, k, lane);
            d = 0.0;
            for (int j = 0; j < k; ++j) d += sqrt(fmax(A[j * k + j], 0.0));
            for (int i = lane; i < k * k; i += 64) { // R = M W diag(1/s) W^T
                const int a = i / k, c = i % k;
                double v = 0.0;
                for (int j = 0; j < k; ++j) {
                    const double sj = sqrt(fmax(A[j * k + j], 0.0));
                    double mw = 0.0;
                    for (int r = 0; r < k; ++r) mw += M[a * k + r] * W[r * k + j];
                    if (sj > 0.0) v += mw / sj * W[c * k + j];
                    else v += W[a * k + j] * W[c * k + j]; // null direction of M: see below
                }
                Rm[i] = v;
            }
            wsync();
            if (d_old != 0.0 && d / d_old < 1.0 + tol) break;
        }
        for (int i = lane; i < p * k; i += 64) {
            double v = 0.0;
            for (int j = 0; j < k; ++j) v += phi[(i / k) * k + j] * Rm[j * k + i % k];
            Ld[(i / k) * KMAX + i % k] = v * sqrt(comm[i / k]);
        }
        wsync();
    }
    // sign convention (:100-108)
    for (int j = lane; j < k; j += 64) {
        double s = 0.0;
        for (int i = 0; i < p; ++i) s += Ld[i * KMAX + j];
        if (s < 0.0)
            for (int i = 0; i < p; ++i) Ld[i * KMAX + j] = -Ld[i * KMAX + j];
    }
}

// symmetric eigen-decomposition for B matrices (exposed for tests and for callers that want the reference's
// _get_eigval pieces): eigenvalues descending, eigenvectors in the columns of vec
__global__ void __launch_bounds__(64) fa_eigh_kernel(long B, int N, const double *sym, double *val, double *vec)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = N, lane = threadIdx.x;
    const long b = blockIdx.x;
    double *A = lds, *V = A + n * n;
    int *perm = reinterpret_cast<int *>(V + n * n);
    for (int i = lane; i < n * n; i += 64) A[i] = sym[b * n * n + i];
    wsync();
    jacobi_eigh(A, V, n, lane);
    for (int k = lane; k < n; k += 64) perm[rank_desc(A, n, k)] = k;
    wsync();
    for (int s = lane; s < n; s += 64) val[b * n + s] = A[perm[s] * n + perm[s]];
    if (vec)
        for (int i = lane; i < n * n; i += 64) vec[b * n * n + i] = V[(i / n) * n + perm[i % n]];
}

hipError_t launch_fa_corr(long R, long T, int N, int time_major, const double *obs, double *corr, hipStream_t s)
{
    const long bs = time_major ? 1 : T, ts = time_major ? R : 1;
    hipLaunchKernelGGL(fa_corr_kernel, dim3((unsigned)R), dim3(256), 0, s, R, T, N, bs, ts, obs, corr);
    return hipGetLastError();
}
hipError_t launch_fa_analyse(long B, int N, long maxfactors, const double *corr, double *eigval, long long *nfact,
                             long long *nfact_map, long long *nfact_map4, double *psi0, unsigned *status, hipStream_t s)
{
    const size_t sh = sizeof(double) * (3 * N * N + 3 * N + (N + 1) / 2) + 16;
    hipLaunchKernelGGL(fa_analyse_kernel, dim3((unsigned)B), dim3(64), sh, s, B, N, maxfactors, corr, eigval, nfact, nfact_map,
                       nfact_map4, psi0, status);
    return hipGetLastError();
}
hipError_t launch_fa_minres(long B, long R, int N, int KMAX, const double *corr, const long long *nfact, const double *psi,
                            const long long *order, double *fval, double *grad, double *loadings, hipStream_t s)
{
    const size_t sh = sizeof(double) * (2 * N * N + N + N * KMAX) + sizeof(int) * N + 16;
    hipLaunchKernelGGL(fa_minres_kernel, dim3((unsigned)B), dim3(64), sh, s, B, R, N, KMAX, corr, nfact, psi, order, fval,
                       grad, loadings);
    return hipGetLastError();
}
hipError_t launch_fa_rotate(long B, int N, int KMAX, const long long *nfact, double *loadings, double gamma, int maxiter,
                            double tol, hipStream_t s)
{
    const size_t sh = sizeof(double) * (2 * N * KMAX + 4 * KMAX * KMAX + N + KMAX) + 16;
    hipLaunchKernelGGL(fa_rotate_kernel, dim3((unsigned)B), dim3(64), sh, s, B, N, KMAX, nfact, loadings, gamma, maxiter, tol);
    return hipGetLastError();
}
hipError_t launch_fa_eigh(long B, int N, const double *sym, double *val, double *vec, hipStream_t s)
{
    const size_t sh = sizeof(double) * (2 * N * N) + sizeof(int) * N + 16;
    hipLaunchKernelGGL(fa_eigh_kernel, dim3((unsigned)B), dim3(64), sh, s, B, N, sym, val, vec);
    return hipGetLastError();
}

} // namespace mk
