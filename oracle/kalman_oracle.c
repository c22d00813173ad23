/*
 * kalman_oracle.c -- CPU restatement of the pastas/metran Kalman hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP kernels in
 * metran_amd/csrc: only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it.  The product path (metran_amd) never links,
 * imports or calls anything in oracle/ and fails loudly without its HIP library.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
 * against fixtures generated from the reference itself (tests/golden/make_golden.py
 * runs the unmodified /root/reference package) including the reference's stored
 * notebook known-answers.
 *
 * Each function follows the reference lines it cites, operation by operation and in
 * the same floating-point order (compile with -ffp-contract=off), so that the filter
 * outputs are bit-identical to the un-jitted reference except for libm's log().
 *
 * All matrices are row-major double; integer bookkeeping is int64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------
 * a1: SPKalmanFilter.set_observations          /root/reference/metran/kalmanfilter.py:646-674
 *
 * in : oseries [T,N] (NaN / +-inf = missing, :657)
 * out: observations [T,N] (missing -> 0.0, :661,:673), observation_indices [T,N] stored
 *      as DOUBLE, left-packed ascending (:659,:674), observation_count [T] (:660,:668).
 * The reference finds the valid entries with "(x + 1e10).nonzero()" (:666-667); a finite
 * observation equal to exactly -1e10 is therefore dropped.  Restated faithfully.
 * ---------------------------------------------------------------------------------- */
ORACLE_API void oracle_set_observations(int64_t T, int64_t N, const double *oseries,
                                        double *observations, double *observation_indices,
                                        int64_t *observation_count)
{
    memset(observations, 0, sizeof(double) * (size_t)(T * N));
    memset(observation_indices, 0, sizeof(double) * (size_t)(T * N));
    for (int64_t t = 0; t < T; ++t) {
        int64_t cnt = 0;
        for (int64_t j = 0; j < N; ++j) {
            double y = oseries[t * N + j];
            if (!isfinite(y)) continue;      /* masked (:657) */
            if (y + 1e10 == 0.0) continue;   /* nonzero() quirk (:666-667) */
            observations[t * N + j] = y;
            observation_indices[t * N + cnt] = (double)j;
            ++cnt;
        }
        observation_count[t] = cnt;
    }
}

/* ------------------------------------------------------------------------------------
 * a3: seqkalmanfilter (the numba engine)        metran/kalmanfilter.py:236-400
 *
 * 9 inputs exactly as the numba signature (:236-242); 7 outputs as the returned tuple
 * (:392-400).  sigmas/detfs are written compressed (only steps with >=1 observation,
 * :380-382) and zero elsewhere (:307-308).
 *   - predicted mean : full mat-vec Phi @ x                      (:318-322)
 *   - predicted cov  : Phi[r,r] * P[r,c] * Phi[c,c] + Q[r,c]      (:324-331, diag(Phi) only)
 *   - scalar update per listed observation                        (:341-378)
 * ---------------------------------------------------------------------------------- */
ORACLE_API void oracle_seqkalmanfilter(int64_t T, int64_t N, int64_t n,
                                       const double *observations,        /* [T,N] */
                                       const double *transition_matrix,   /* [n,n] */
                                       const double *transition_cov,      /* [n,n] */
                                       const double *observation_matrix,  /* [N,n] */
                                       const double *observation_var,     /* [N]   */
                                       const double *observation_indices, /* [T,N] */
                                       const int64_t *observation_count,  /* [T]   */
                                       const double *x0,                  /* [n]   */
                                       const double *P0,                  /* [n,n] */
                                       double *sigmas, double *detfs,     /* [T]   */
                                       int64_t *sigmacount_out,
                                       double *F, double *Pf,             /* [T,n], [T,n,n] */
                                       double *Xp, double *Pp)
{
    double *xf = (double *)malloc(sizeof(double) * (size_t)n);
    double *Pc = (double *)malloc(sizeof(double) * (size_t)(n * n));
    double *xp = (double *)malloc(sizeof(double) * (size_t)n);
    double *Pq = (double *)malloc(sizeof(double) * (size_t)(n * n));
    double *dotmat = (double *)malloc(sizeof(double) * (size_t)n);
    double *kgain = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(xf, x0, sizeof(double) * (size_t)n);
    memcpy(Pc, P0, sizeof(double) * (size_t)(n * n));
    for (int64_t t = 0; t < T; ++t) { sigmas[t] = 0.0; detfs[t] = 0.0; }
    int64_t sigmacount = 0;

    for (int64_t t = 0; t < T; ++t) {
        for (int64_t r = 0; r < n; ++r) {            /* :318-322 */
            double summed = 0.0;
            for (int64_t c = 0; c < n; ++c) summed += transition_matrix[r * n + c] * xf[c];
            xp[r] = summed;
        }
        for (int64_t r = 0; r < n; ++r)              /* :324-331 */
            for (int64_t c = 0; c < n; ++c)
                Pq[r * n + c] = transition_matrix[r * n + r] * Pc[r * n + c] * transition_matrix[c * n + c]
                                + transition_cov[r * n + c];
        memcpy(Xp + t * n, xp, sizeof(double) * (size_t)n);          /* :332 */
        memcpy(Pp + t * n * n, Pq, sizeof(double) * (size_t)(n * n)); /* :333 */

        if (observation_count[t] > 0) {              /* :335 */
            double sigma = 0.0, detf = 0.0;
            for (int64_t i = 0; i < observation_count[t]; ++i) {
                int64_t idx = (int64_t)observation_indices[t * N + i];   /* :342 */
                const double *zrow = observation_matrix + idx * n;
                double summed = 0.0;                 /* :344-347 */
                for (int64_t r = 0; r < n; ++r) summed += zrow[r] * xp[r];
                double innovation = observations[t * N + idx] - summed;
                for (int64_t r = 0; r < n; ++r) {    /* :349-357 */
                    double s2 = 0.0;
                    for (int64_t c = 0; c < n; ++c) s2 += Pq[r * n + c] * zrow[c];
                    dotmat[r] = s2;
                }
                summed = 0.0;                        /* :359-362 */
                for (int64_t r = 0; r < n; ++r) summed += zrow[r] * dotmat[r];
                double innovation_variance = observation_var[idx] + summed;
                for (int64_t r = 0; r < n; ++r) kgain[r] = dotmat[r] / innovation_variance; /* :364-366 */
                for (int64_t r = 0; r < n; ++r)      /* :368-372 */
                    for (int64_t c = 0; c < n; ++c)
                        Pq[r * n + c] += -kgain[r] * kgain[c] * innovation_variance;
                for (int64_t r = 0; r < n; ++r) xp[r] += kgain[r] * innovation;            /* :374-375 */
                sigma += innovation * innovation / innovation_variance;                     /* :377 */
                detf += log(innovation_variance);                                           /* :378 */
            }
            sigmas[sigmacount] = sigma;              /* :380-382 */
            detfs[sigmacount] = detf;
            ++sigmacount;
        }
        memcpy(xf, xp, sizeof(double) * (size_t)n);  /* :384-388 */
        memcpy(Pc, Pq, sizeof(double) * (size_t)(n * n));
        memcpy(F + t * n, xf, sizeof(double) * (size_t)n);            /* :389 */
        memcpy(Pf + t * n * n, Pc, sizeof(double) * (size_t)(n * n)); /* :390 */
    }
    *sigmacount_out = sigmacount;
    free(xf); free(Pc); free(xp); free(Pq); free(dotmat); free(kgain);
}

/* ------------------------------------------------------------------------------------
 * a6: SPKalmanFilter.get_mle                    metran/kalmanfilter.py:550-567
 *   detfs[warmup:], sigmas[warmup:] index the COMPRESSED arrays (length sigmacount, :773-774),
 *   observation_count[warmup:] indexes TIME STEPS (:565).  Returned value is -2 log L.
 *   np.sum is a pairwise sum; a plain left-to-right sum differs by O(1e-16*sqrt(T)) relative.
 * ---------------------------------------------------------------------------------- */
ORACLE_API double oracle_get_mle(int64_t T, int64_t sigmacount, const double *sigmas,
                                 const double *detfs, const int64_t *observation_count,
                                 int64_t warmup)
{
    double sd = 0.0, ss = 0.0;
    int64_t nobs = 0;
    for (int64_t i = warmup; i < sigmacount; ++i) { sd += detfs[i]; ss += sigmas[i]; }
    for (int64_t t = warmup; t < T; ++t) nobs += observation_count[t];
    return (double)nobs * log(2.0 * M_PI) + sd + ss;   /* :566 */
}

/* ---- symmetric eigen-decomposition (cyclic Jacobi) used for the pseudo-inverse --------- */
static void jacobi_eigh(int64_t n, double *A /* in: sym, destroyed */, double *V, double *w)
{
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            diag += A[i * n + i] * A[i * n + i];
            for (int64_t j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-300 || off <= 1e-34 * diag) break;
        for (int64_t p = 0; p < n - 1; ++p)
            for (int64_t q = p + 1; q < n; ++q) {
                double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int64_t k = 0; k < n; ++k) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int64_t k = 0; k < n; ++k) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int64_t k = 0; k < n; ++k) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int64_t i = 0; i < n; ++i) w[i] = A[i * n + i];
}

/* np.linalg.pinv(A) for SYMMETRIC A (metran/kalmanfilter.py:455): numpy takes the SVD and
 * discards singular values <= rcond * max(s) with rcond = 1e-15; for a symmetric matrix the
 * singular values are |eigenvalues| and pinv = sum_{|w|>cutoff} v v^T / w. */
static void sym_pinv(int64_t n, const double *A, double *Ainv, double *work /* 2*n*n + n */)
{
    double *M = work, *V = work + n * n, *w = work + 2 * n * n;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) M[i * n + j] = 0.5 * (A[i * n + j] + A[j * n + i]);
    jacobi_eigh(n, M, V, w);
    double smax = 0.0;
    for (int64_t i = 0; i < n; ++i) if (fabs(w[i]) > smax) smax = fabs(w[i]);
    double cutoff = 1e-15 * smax;
    for (int64_t i = 0; i < n * n; ++i) Ainv[i] = 0.0;
    for (int64_t k = 0; k < n; ++k) {
        if (!(fabs(w[k]) > cutoff)) continue;
        double iw = 1.0 / w[k];
        for (int64_t i = 0; i < n; ++i) {
            double vi = V[i * n + k] * iw;
            for (int64_t j = 0; j < n; ++j) Ainv[i * n + j] += vi * V[j * n + k];
        }
    }
}

static void matmul(int64_t n, const double *A, const double *B, double *C) /* C = A @ B */
{
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) {
            double s = 0.0;
            for (int64_t k = 0; k < n; ++k) s += A[i * n + k] * B[k * n + j];
            C[i * n + j] = s;
        }
}

/* ------------------------------------------------------------------------------------
 * a7: kalmansmoother (RTS)                      metran/kalmanfilter.py:403-476
 *   last step = filtered (:450-451); for t = T-2 .. 0:
 *     psc_inv = pinv(Pp[t+1])                                   (:455)
 *     J = Pf[t] @ (Phi^T @ psc_inv)                             (:458-460)
 *     S[t]  = F[t]  + J @ (S[t+1]  - Xp[t+1])                   (:461-464)
 *     Ps[t] = Pf[t] + J @ ((Ps[t+1] - Pp[t+1]) @ J^T)           (:465-474)
 * ---------------------------------------------------------------------------------- */
ORACLE_API void oracle_kalmansmoother(int64_t T, int64_t n, const double *F, const double *Pf,
                                      const double *Xp, const double *Pp,
                                      const double *transition_matrix, double *S, double *Ps)
{
    size_t nn = (size_t)(n * n);
    double *buf = (double *)malloc(sizeof(double) * (8 * nn + 2 * (size_t)n));
    double *inv = buf, *PhiT = buf + nn, *tmp = buf + 2 * nn, *J = buf + 3 * nn, *D = buf + 4 * nn,
           *JT = buf + 5 * nn, *work = buf + 6 * nn; /* work: 2*nn + n */
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) PhiT[i * n + j] = transition_matrix[j * n + i];
    memcpy(S + (T - 1) * n, F + (T - 1) * n, sizeof(double) * (size_t)n);
    memcpy(Ps + (T - 1) * n * n, Pf + (T - 1) * n * n, sizeof(double) * nn);
    for (int64_t t = T - 2; t >= 0; --t) {
        const double *Pp1 = Pp + (t + 1) * n * n, *Pft = Pf + t * n * n;
        sym_pinv(n, Pp1, inv, work);
        matmul(n, PhiT, inv, tmp);
        matmul(n, Pft, tmp, J);
        for (int64_t i = 0; i < n; ++i) {
            double s = 0.0;
            for (int64_t k = 0; k < n; ++k) s += J[i * n + k] * (S[(t + 1) * n + k] - Xp[(t + 1) * n + k]);
            S[t * n + i] = F[t * n + i] + s;
        }
        for (size_t i = 0; i < nn; ++i) D[i] = Ps[(t + 1) * n * n + (int64_t)i] - Pp1[i];
        for (int64_t i = 0; i < n; ++i)
            for (int64_t j = 0; j < n; ++j) JT[i * n + j] = J[j * n + i];
        matmul(n, D, JT, tmp);
        matmul(n, J, tmp, inv);
        for (size_t i = 0; i < nn; ++i) Ps[t * n * n + (int64_t)i] = Pft[i] + inv[i];
    }
    free(buf);
}

/* ------------------------------------------------------------------------------------
 * a9: SPKalmanFilter.simulate                   metran/kalmanfilter.py:569-603
 *   means[t] = Z @ x[t] (:597); vars[t] = max(diag(Z @ (P[t] @ Z^T)), 0) (:598-602)
 * ---------------------------------------------------------------------------------- */
ORACLE_API void oracle_simulate(int64_t T, int64_t N, int64_t n, const double *Z,
                                const double *means, const double *covs,
                                double *sim_means, double *sim_vars)
{
    double *pz = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t t = 0; t < T; ++t) {
        const double *x = means + t * n, *P = covs + t * n * n;
        for (int64_t j = 0; j < N; ++j) {
            const double *z = Z + j * n;
            double m = 0.0;
            for (int64_t c = 0; c < n; ++c) m += z[c] * x[c];
            sim_means[t * N + j] = m;
            for (int64_t r = 0; r < n; ++r) {
                double s = 0.0;
                for (int64_t c = 0; c < n; ++c) s += P[r * n + c] * z[c];
                pz[r] = s;
            }
            double v = 0.0;
            for (int64_t r = 0; r < n; ++r) v += z[r] * pz[r];
            sim_vars[t * N + j] = v > 0.0 ? v : 0.0;
        }
    }
    free(pz);
}

/* ------------------------------------------------------------------------------------
 * a9: SPKalmanFilter.decompose                  metran/kalmanfilter.py:605-644
 *   sdf[t]    = Z[:, :N] @ x[t, :N]           (:634)
 *   cdf[k][t] = Z[:, N+k] * x[t, N+k]         (:641)      (K = n - N)
 * ---------------------------------------------------------------------------------- */
ORACLE_API void oracle_decompose(int64_t T, int64_t N, int64_t n, const double *Z,
                                 const double *means, double *sdf /* [T,N] */,
                                 double *cdf /* [K,T,N] */)
{
    int64_t K = n - N;
    for (int64_t t = 0; t < T; ++t) {
        const double *x = means + t * n;
        for (int64_t j = 0; j < N; ++j) {
            double s = 0.0;
            for (int64_t c = 0; c < N; ++c) s += Z[j * n + c] * x[c];
            sdf[t * N + j] = s;
            for (int64_t k = 0; k < K; ++k) cdf[(k * T + t) * N + j] = Z[j * n + N + k] * x[N + k];
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Batched driver used by tests and by bench.py's cpu_baseline leg: B independent models,
 * Metran's structure (diagonal Phi/Q given as vectors, Z = [I | loadings], metran/metran.py:
 * 265-384), NaN-encoded observations.  Builds the dense 9-argument inputs per model and calls
 * the functions above -- i.e. it times/validates exactly the reference algorithm.
 * Null output pointers are skipped.  OpenMP over models when compiled with -fopenmp.
 * ---------------------------------------------------------------------------------- */
ORACLE_API int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORACLE_API void oracle_set_num_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}


ORACLE_API void oracle_dfm_batch(int64_t B, int64_t T, int64_t N, int64_t K,
                                 const double *obs,      /* [B,T,N] NaN = missing */
                                 const double *phi,      /* [B,n] */
                                 const double *q,        /* [B,n] */
                                 const double *loadings, /* [B,N,K] */
                                 const double *obsvar,   /* [B,N] or NULL (zeros) */
                                 int64_t warmup, int do_smooth,
                                 double *mle,            /* [B] */
                                 double *sigmas, double *detfs, /* [B,T] or NULL */
                                 int64_t *sigmacount,    /* [B] or NULL */
                                 double *F, double *Pf, double *Xp, double *Pp, /* or NULL */
                                 double *S, double *Ps)  /* or NULL */
{
    int64_t n = N + K;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b) {
        size_t nn = (size_t)(n * n);
        double *Phi = (double *)calloc(nn, sizeof(double));
        double *Q = (double *)calloc(nn, sizeof(double));
        double *Z = (double *)calloc((size_t)(N * n), sizeof(double));
        double *R = (double *)calloc((size_t)N, sizeof(double));
        double *x0 = (double *)calloc((size_t)n, sizeof(double));
        double *P0 = (double *)calloc(nn, sizeof(double));
        double *o = (double *)malloc(sizeof(double) * (size_t)(T * N));
        double *oi = (double *)malloc(sizeof(double) * (size_t)(T * N));
        int64_t *oc = (int64_t *)malloc(sizeof(int64_t) * (size_t)T);
        double *sg = (double *)malloc(sizeof(double) * (size_t)T);
        double *df = (double *)malloc(sizeof(double) * (size_t)T);
        double *f = F ? F + b * T * n : (double *)malloc(sizeof(double) * (size_t)(T * n));
        double *pf = Pf ? Pf + b * T * n * n : (double *)malloc(sizeof(double) * (size_t)T * nn);
        double *xp = Xp ? Xp + b * T * n : (double *)malloc(sizeof(double) * (size_t)(T * n));
        double *pp = Pp ? Pp + b * T * n * n : (double *)malloc(sizeof(double) * (size_t)T * nn);
        for (int64_t i = 0; i < n; ++i) {
            Phi[i * n + i] = phi[b * n + i];
            Q[i * n + i] = q[b * n + i];
            P0[i * n + i] = 1.0;                       /* run_filter default, kalmanfilter.py:747-750 */
        }
        for (int64_t j = 0; j < N; ++j) {
            Z[j * n + j] = 1.0;
            for (int64_t k = 0; k < K; ++k) Z[j * n + N + k] = loadings[(b * N + j) * K + k];
            if (obsvar) R[j] = obsvar[b * N + j];
        }
        oracle_set_observations(T, N, obs + b * T * N, o, oi, oc);
        int64_t sc = 0;
        oracle_seqkalmanfilter(T, N, n, o, Phi, Q, Z, R, oi, oc, x0, P0, sg, df, &sc, f, pf, xp, pp);
        if (mle) mle[b] = oracle_get_mle(T, sc, sg, df, oc, warmup);
        if (sigmas) memcpy(sigmas + b * T, sg, sizeof(double) * (size_t)T);
        if (detfs) memcpy(detfs + b * T, df, sizeof(double) * (size_t)T);
        if (sigmacount) sigmacount[b] = sc;
        if (do_smooth) {
            double *s = S ? S + b * T * n : (double *)malloc(sizeof(double) * (size_t)(T * n));
            double *ps = Ps ? Ps + b * T * n * n : (double *)malloc(sizeof(double) * (size_t)T * nn);
            oracle_kalmansmoother(T, n, f, pf, xp, pp, Phi, s, ps);
            if (!S) free(s);
            if (!Ps) free(ps);
        }
        if (!F) free(f);
        if (!Pf) free(pf);
        if (!Xp) free(xp);
        if (!Pp) free(pp);
        free(Phi); free(Q); free(Z); free(R); free(x0); free(P0);
        free(o); free(oi); free(oc); free(sg); free(df);
    }
}
