/*
 * kalman_fast.c -- an OPTIMISED CPU implementation of the same hot path, for bench.py's `cpu_baseline_optimised` leg.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything under oracle/): the product path never links it.
 *
 * What it is.  oracle/kalman_oracle.c is the FIDELITY checker: it restates the reference operation by operation (dense
 * products against a mostly-zero observation row, np.zeros inside the loops, a Jacobi-eigen pseudo-inverse every smoother
 * step, -ffp-contract=off) and is slower per core than the Python reference it restates.  It says nothing about what a CPU
 * can do on this problem.  This file is the "numba-class or better" CPU leg SURVEY.md section 8d(1) asks for: the same
 * recursions (seqkalmanfilter /root/reference/metran/kalmanfilter.py:236-400, get_mle :550-567, kalmansmoother :403-476,
 * simulate :569-603), written the way one would write them for speed --
 *   - Phi, Q diagonal and Z = [I | G] exploited (what Metran always supplies, metran.py:283-370): d = P z' costs 1 + K terms
 *     per row instead of n, the predict one multiply-add per element;
 *   - symmetric rank-one updates on the lower triangle only, mirrored once per step;
 *   - the smoother's gain from a Cholesky factorisation of the predicted covariance and two triangular solves (the
 *     reference: SVD-based pinv), J (Ps' - Pp) J' as two n^3 products with the inner loops unit-stride;
 *   - no allocation inside the time loop, one model per OpenMP thread, -O3 -march=native -ffp-contract=fast;
 * NOT bit-faithful: parity with the checker is asserted by bench.py at 1e-9 (and by tests/test_oracle_golden.py on the CPU).
 * A predicted covariance that is not positive definite (the heywood fixtures) is outside its scope: the factorisation
 * reports it (status 1) instead of dropping the direction.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define FAST_API __attribute__((visibility("default")))

static const double kLog2Pi = 1.8378770664093454835606594728112;

/* mode: 0 = objective only, 1 = + projected smoothed means / variances [T,N] (sim), 2 = + all six state arrays */
static int one_model(int64_t T, int64_t N, int64_t K, const double *restrict obs, const double *restrict phi, const double *restrict q,
                     const double *restrict G, int64_t warmup, int mode, double *mle, double *F, double *Pf, double *Xp, double *Pp,
                     double *S, double *Ps, double *sim_m, double *sim_v, double *work)
{
    const int64_t n = N + K, nn = n * n;
    double *P = work, *x = P + nn, *d = x + n;            /* filter state */
    double *L = d + n, *X = L + nn, *D = X + nn, *V = D + nn, *Psn = V + nn, *xs = Psn + nn, *dl = xs + n; /* smoother */
    int owns = 0, status = 0;
    if (mode >= 1 && !F) { /* projection mode keeps its own filtered moments */
        F = (double *)malloc(sizeof(double) * (size_t)(T * (n + nn)));
        if (!F) return 2;
        Pf = F + T * n;
        owns = 1;
    }
    for (int64_t i = 0; i < n; ++i) x[i] = 0.0;
    for (int64_t i = 0; i < nn; ++i) P[i] = 0.0;
    for (int64_t i = 0; i < n; ++i) P[i * n + i] = 1.0;
    double sum_sig = 0.0, sum_det = 0.0;
    int64_t nobs = 0, sc = 0;
    for (int64_t t = 0; t < T; ++t) {
        const double *y = obs + t * N;
        for (int64_t i = 0; i < n; ++i) x[i] *= phi[i];
        for (int64_t r = 0; r < n; ++r) {
            const double pr = phi[r];
            double *Pr = P + r * n;
            for (int64_t c = 0; c <= r; ++c) Pr[c] = Pr[c] * pr * phi[c];
            Pr[r] += q[r];
        }
        if (mode == 2) {
            memcpy(Xp + t * n, x, sizeof(double) * (size_t)n);
            double *o = Pp + t * nn;
            for (int64_t r = 0; r < n; ++r)
                for (int64_t c = 0; c <= r; ++c) o[r * n + c] = o[c * n + r] = P[r * n + c];
        }
        double sigma = 0.0, detf = 0.0;
        int64_t cnt = 0;
        for (int64_t j = 0; j < N; ++j) {
            const double yj = y[j];
            if (!isfinite(yj)) continue;
            ++cnt;
            const double *g = G + j * K;
            /* d = P z' with P held as its lower triangle: P[r][c] for c <= r, P[c][r] otherwise */
            for (int64_t r = 0; r < n; ++r) {
                double s = r >= j ? P[r * n + j] : P[j * n + r];
                for (int64_t k = 0; k < K; ++k) s += (r >= N + k ? P[r * n + N + k] : P[(N + k) * n + r]) * g[k];
                d[r] = s;
            }
            double f = d[j], v = yj - x[j];
            for (int64_t k = 0; k < K; ++k) {
                f += g[k] * d[N + k];
                v -= g[k] * x[N + k];
            }
            const double rf = 1.0 / f;
            for (int64_t r = 0; r < n; ++r) {
                const double kr = d[r] * rf;
                x[r] += kr * v;
                double *Pr = P + r * n;
                for (int64_t c = 0; c <= r; ++c) Pr[c] -= kr * d[c];
            }
            sigma += v * v * rf;
            detf += log(f);
            if (!(f > 0.0)) status = 1;
        }
        if (cnt > 0) {
            if (sc >= warmup) {
                sum_sig += sigma;
                sum_det += detf;
            }
            ++sc;
        }
        if (t >= warmup) nobs += cnt;
        if (mode >= 1) {
            memcpy(F + t * n, x, sizeof(double) * (size_t)n);
            double *o = Pf + t * nn;
            for (int64_t r = 0; r < n; ++r)
                for (int64_t c = 0; c <= r; ++c) o[r * n + c] = o[c * n + r] = P[r * n + c];
        }
    }
    *mle = ((double)nobs * kLog2Pi + sum_det) + sum_sig;
    if (mode == 0) return status;

    /* ---- RTS smoother, backwards (:450-474) */
    memcpy(xs, F + (T - 1) * n, sizeof(double) * (size_t)n);
    memcpy(Psn, Pf + (T - 1) * nn, sizeof(double) * (size_t)nn);
    for (int64_t t = T - 1; t >= 0; --t) {
        if (t < T - 1) {
            const double *Pft = Pf + t * nn, *Ft = F + t * n;
            /* L = chol(Pp[t+1]), Pp = Phi Pf Phi + Q;  D = Ps[t+1] - Pp[t+1];  X = Phi Pf (right-hand sides, row r scaled) */
            for (int64_t r = 0; r < n; ++r)
                for (int64_t c = 0; c < n; ++c) {
                    const double pf = Pft[r * n + c], pp = phi[r] * pf * phi[c] + (r == c ? q[r] : 0.0);
                    L[r * n + c] = pp;
                    D[r * n + c] = Psn[r * n + c] - pp;
                    X[r * n + c] = phi[r] * pf;
                }
            for (int64_t j = 0; j < n; ++j) {
                double s = L[j * n + j];
                for (int64_t k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
                if (!(s > 0.0)) {
                    status = 1;
                    s = 1e-300;
                }
                const double ljj = sqrt(s), inv = 1.0 / ljj;
                L[j * n + j] = ljj;
                for (int64_t i = j + 1; i < n; ++i) {
                    double u = L[i * n + j];
                    for (int64_t k = 0; k < j; ++k) u -= L[i * n + k] * L[j * n + k];
                    L[i * n + j] = u * inv;
                }
            }
            /* X <- Pp^-1 X: forward and backward substitution on all n right-hand sides at once (rows of X are unit-stride) */
            for (int64_t i = 0; i < n; ++i) {
                double *Xi = X + i * n;
                for (int64_t k = 0; k < i; ++k) {
                    const double l = L[i * n + k];
                    const double *Xk = X + k * n;
                    for (int64_t c = 0; c < n; ++c) Xi[c] -= l * Xk[c];
                }
                const double inv = 1.0 / L[i * n + i];
                for (int64_t c = 0; c < n; ++c) Xi[c] *= inv;
            }
            for (int64_t i = n - 1; i >= 0; --i) {
                double *Xi = X + i * n;
                for (int64_t k = i + 1; k < n; ++k) {
                    const double l = L[k * n + i];
                    const double *Xk = X + k * n;
                    for (int64_t c = 0; c < n; ++c) Xi[c] -= l * Xk[c];
                }
                const double inv = 1.0 / L[i * n + i];
                for (int64_t c = 0; c < n; ++c) Xi[c] *= inv;
            }
            /* now X = J' (X[c][i] = J[i][c]).  delta = xs - Phi F[t];  xs <- F[t] + J delta */
            for (int64_t c = 0; c < n; ++c) dl[c] = xs[c] - phi[c] * Ft[c];
            for (int64_t i = 0; i < n; ++i) xs[i] = Ft[i];
            for (int64_t c = 0; c < n; ++c) {
                const double dc = dl[c];
                const double *Xc = X + c * n;
                for (int64_t i = 0; i < n; ++i) xs[i] += Xc[i] * dc;
            }
            /* V = D J' ;  Ps = Pf + J V */
            for (int64_t r = 0; r < n; ++r) {
                double *Vr = V + r * n;
                for (int64_t i = 0; i < n; ++i) Vr[i] = 0.0;
                for (int64_t c = 0; c < n; ++c) {
                    const double drc = D[r * n + c];
                    const double *Xc = X + c * n;
                    for (int64_t i = 0; i < n; ++i) Vr[i] += drc * Xc[i];
                }
            }
            memcpy(Psn, Pft, sizeof(double) * (size_t)nn);
            for (int64_t r = 0; r < n; ++r) {
                const double *Xr = X + r * n, *Vr = V + r * n;
                for (int64_t i = 0; i < n; ++i) {
                    const double jir = Xr[i];
                    double *Pi = Psn + i * n;
                    for (int64_t j = 0; j < n; ++j) Pi[j] += jir * Vr[j];
                }
            }
        }
        if (mode == 2) {
            memcpy(S + t * n, xs, sizeof(double) * (size_t)n);
            memcpy(Ps + t * nn, Psn, sizeof(double) * (size_t)nn);
        }
        if (sim_m || sim_v) { /* simulate (:597-602) with Z = [I | G] */
            for (int64_t j = 0; j < N; ++j) {
                const double *g = G + j * K;
                double m = xs[j], var = Psn[j * n + j];
                for (int64_t k = 0; k < K; ++k) {
                    m += g[k] * xs[N + k];
                    double row = 2.0 * Psn[j * n + N + k];
                    for (int64_t k2 = 0; k2 < K; ++k2) row += g[k2] * Psn[(N + k) * n + N + k2];
                    var += g[k] * row;
                }
                if (sim_m) sim_m[t * N + j] = m;
                if (sim_v) sim_v[t * N + j] = var > 0.0 ? var : 0.0;
            }
        }
    }
    if (owns) free(F);
    return status;
}

/* B models, OpenMP over models.  Arrays as in oracle_dfm_batch: obs [B,T,N], phi / q [B,n], loadings [B,N,K], outputs per mode
 * (NULL = skipped): mle [B]; F, Xp, S [B,T,n]; Pf, Pp, Ps [B,T,n,n]; sim_means, sim_vars [B,T,N].  Returns the number of models
 * whose status is non-zero (a non-positive innovation variance or a predicted covariance that is not positive definite). */
FAST_API int64_t fast_dfm_batch(int64_t B, int64_t T, int64_t N, int64_t K, const double *obs, const double *phi, const double *q,
                                const double *loadings, int64_t warmup, int mode, double *mle, double *F, double *Pf, double *Xp, double *Pp,
                                double *S, double *Ps, double *sim_means, double *sim_vars)
{
    const int64_t n = N + K, nn = n * n;
    int64_t bad = 0;
#pragma omp parallel reduction(+ : bad)
    {
        double *work = (double *)malloc(sizeof(double) * (size_t)(6 * nn + 6 * n));
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < B; ++b) {
            const int st = work ? one_model(T, N, K, obs + b * T * N, phi + b * n, q + b * n, loadings + b * N * K, warmup, mode, mle + b,
                                            F ? F + b * T * n : NULL, Pf ? Pf + b * T * nn : NULL, Xp ? Xp + b * T * n : NULL,
                                            Pp ? Pp + b * T * nn : NULL, S ? S + b * T * n : NULL, Ps ? Ps + b * T * nn : NULL,
                                            sim_means ? sim_means + b * T * N : NULL, sim_vars ? sim_vars + b * T * N : NULL, work)
                                : 2;
            bad += st != 0;
        }
        free(work);
    }
    return bad;
}

FAST_API int fast_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* (bench.py: the thread count of the timed pass -- a box whose CPU quota is below its core count is timed per thread count) */
FAST_API void fast_set_num_threads(int t)
{
    if (t > 0) omp_set_num_threads(t);
}

