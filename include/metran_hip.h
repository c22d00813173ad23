/*
 * metran_hip.h -- C ABI of libmetran_hip.so: the MI355X (gfx950) batched Kalman filter,
 * -2 log-likelihood and RTS smoother for Metran's dynamic-factor model.
 *
 * This is the drop-in boundary for the hot path of pastas/metran.
 * The reference has no FFI of its own: its "engine" is a Python callable
 *     SPKalmanFilter.filtermethod(obs, Phi, Q, Z, R, idx, count, x0, P0) -> 7-tuple
 *         /root/reference/metran/kalmanfilter.py:494-504 (binding), :761-771 (call site)
 *     kalmansmoother(F, Pf, Xp, Pp, Phi) -> (S, Ps)
 *         metran/kalmanfilter.py:685-691 (call site), :403-476 (definition)
 *     SPKalmanFilter.get_mle(warmup)            metran/kalmanfilter.py:550-567
 *     SPKalmanFilter.simulate / decompose       metran/kalmanfilter.py:569-644
 *     Metran._get_matrices(p)                   metran/metran.py:246-416
 * Each entry point below names the reference function it replaces.  Python binds this
 * header with ctypes (metran_amd/_lib.py); INTEGRATION.md shows the stub a Metran
 * maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ exceptions cross the ABI.
 *   - every function returns 0 (MK_OK) or a negative mk_status; mk_last_error() gives text.
 *   - "d_" pointers are DEVICE pointers (hipMalloc / torch.Tensor.data_ptr()); "h_" are host.
 *   - all launches are asynchronous on the context's stream (mk_set_stream / mk_sync).
 *   - layouts are the reference's, with a leading batch axis, row-major double:
 *       obs [R,T,N] (NaN or +-inf = missing), phi/q [B,n], loadings [R,N,K], obsvar [R,N],
 *       F/Xp/S [B,T,n], Pf/Pp/Ps [B,T,n,n], sigmas/detfs [B,T], n = N + K;
 *     optionally TIME-MAJOR ([T,B,...], mk_outputs.time_major / mk_problem.obs_time_major): the
 *     same per-model [T,...] arrays, interleaved so that one time step of all models is contiguous.
 *     B = number of filter instances, R = number of observation records; instance i reads
 *     record i % R (so the P+1 finite-difference evaluations of one model, or S parameter sets
 *     per model, share one uploaded observation record).
 *   - state order: N specific factors then K common factors (metran/metran.py:283-290).
 *   - the model is Metran's: Phi = diag(phi), Q = diag(q), Z = [I_N | loadings], R = diag(obsvar).
 */
#ifndef METRAN_HIP_H
#define METRAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MK_API __attribute__((visibility("default")))
#define MK_ABI_VERSION 7

typedef enum mk_status {
    MK_OK = 0,
    MK_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, ...) */
    MK_ERR_SHAPE = -2,       /* (N,K) is not served, or not in the asked-for mode (see mk_shape_supported) */
    MK_ERR_HIP = -3,         /* a HIP runtime call failed */
    MK_ERR_NO_DEVICE = -4,   /* no gfx950 device visible */
    MK_ERR_ALLOC = -5
} mk_status;

/* per-instance status bits written to mk_outputs.d_status */
#define MK_FLAG_NONPOSITIVE_F 1u /* an innovation variance f <= 0 was met in the filter    */
#define MK_FLAG_NOT_SPD 2u       /* ERROR: predicted covariance indefinite in the smoother (an LDL^T
                                    pivot < -1e-8); the smoothed moments of that instance are invalid */
#define MK_FLAG_RANK_DEFICIENT 4u /* INFO: a pivot <= 0 was met and that null direction dropped, as
                                    numpy.linalg.pinv does in the reference (kalmanfilter.py:455): q_i = 0
                                    for a series with communality 1 (metran.py:314-316).  Results valid. */

typedef struct mk_context mk_context; /* opaque; one per (process, device) */

/* Inputs of one batched evaluation.  Replaces the 9 positional arguments of
 * seqkalmanfilter (metran/kalmanfilter.py:243-253) for B models at once. */
typedef struct mk_problem {
    int64_t n_instances; /* B */
    int64_t n_records;   /* R, 1 <= R <= B */
    int64_t T, N, K;
    int64_t warmup;            /* get_mle(warmup), metran/kalmanfilter.py:550; Metran uses 1 */
    const double *d_obs;       /* [R,T,N]  observations, NaN/inf = missing (kalmanfilter.py:657) */
    const double *d_phi;       /* [B,n]    diag of transition_matrix      (metran.py:283-290) */
    const double *d_q;         /* [B,n]    diag of transition_covariance  (metran.py:310-322) */
    const double *d_loadings;  /* [R,N,K]  observation_matrix[:, N:]      (metran.py:365-370) */
    const double *d_obsvar;    /* [R,N] or NULL = zeros                   (metran.py:382-384) */
    const double *d_x0;        /* [B,n] or NULL = zeros       (kalmanfilter.py:747-748) */
    const double *d_P0;        /* [B,n,n] or NULL = identity  (kalmanfilter.py:749-750) */
    int64_t obs_time_major;    /* 0: d_obs is [R,T,N] (reference layout); 1: d_obs is [T,R,N] */
    const double *d_scale;     /* [R,N] or NULL = ones: series standard deviations (Metran.oseries_std); with
                                  d_offset the scaled observation matrix of get_scaled_observation_matrix
                                  (metran.py:944-961) for the fused projection outputs d_sim_means/d_sim_vars */
    const double *d_offset;    /* [R,N] or NULL = zeros: series means (metran.py:788-793)              */
} mk_problem;

/* Outputs; any pointer may be NULL (that output is skipped and costs no HBM traffic).
 * Replaces the returned 7-tuple of seqkalmanfilter (kalmanfilter.py:392-400), the 2-tuple of
 * kalmansmoother (:476) and get_mle (:566). */
typedef struct mk_outputs {
    double *d_mle;         /* [B]   -2 log L with warm-up skip                  (:563-566) */
    double *d_sigmas;      /* [B,T] compressed, zero-filled tail                (:380-382) */
    double *d_detfs;       /* [B,T] compressed, zero-filled tail                           */
    int64_t *d_sigmacount; /* [B]                                                          */
    double *d_F;           /* [B,T,n]   filtered_state_means                    (:389)     */
    double *d_Pf;          /* [B,T,n,n] filtered_state_covariances              (:390)     */
    double *d_Xp;          /* [B,T,n]   predicted_state_means                   (:332)     */
    double *d_Pp;          /* [B,T,n,n] predicted_state_covariances             (:333)     */
    double *d_S;           /* [B,T,n]   smoothed_state_means                    (:461-464) */
    double *d_Ps;          /* [B,T,n,n] smoothed_state_covariances              (:465-474) */
    uint32_t *d_status;    /* [B] MK_FLAG_* bits, or NULL                                  */
    int64_t time_major;    /* 0: per-step arrays are [B,T,...] (reference layout with a leading batch
                              axis); 1: they are [T,B,...] -- all models' step-t blocks contiguous, the
                              HBM-friendly layout (a [B,T,...] strided view of it costs nothing).
                              Applies to d_sigmas, d_detfs, d_F, d_Pf, d_Xp, d_Pp, d_S, d_Ps.        */
    double *d_sim_means;   /* [B,T,N] fused projection epilogue of the smoother (mk_smooth /          */
    double *d_sim_vars;    /* [B,T,N] mk_filter_smooth, record layout): SPKalmanFilter.simulate
                              (kalmanfilter.py:569-603) of the SMOOTHED moments with Z~ = diag(scale)
                              [I | loadings] (+ offset on the means) -- what Metran.get_simulation consumes
                              (metran.py:831-883).  With these set, d_S / d_Ps (and d_Xp / d_Pp) may be
                              NULL: the smoothed states are then never written (7x less output at n = 10,
                              40x at n = 36).  time_major applies.  NULL = not computed.              */
    int64_t record_stride; /* 0: every array above is dense.  RS = mk_record_stride(n): PACKED RECORDS, the
                              fast path.  Each moment set is ONE array of RS doubles per (model, step):
                                [ mean(n) | covariance(n*n) | sigma, detf (filtered set only) | zero pad ]
                              and the pointers are views into it: d_Pp = d_Xp + n, d_Pf = d_F + n,
                              d_Ps = d_S + n, d_sigmas = d_F + n + n*n, d_detfs = d_sigmas + 1 (or NULL);
                              element (b,t) of any of them sits RS doubles after element (b,t)-1 of the
                              layout chosen by time_major.  RS*8 is a multiple of 128 bytes: all stores of
                              the kernels then cover whole cache lines (partial-line stores of the small
                              vectors were measured to throttle HBM writes).  mk_filter uses records when
                              the predicted AND filtered sets are both requested this way; mk_smooth when
                              d_F (and d_S) are record arrays.                                         */
    int64_t flags;         /* MK_OUT_* bits (0 = none):
                              MK_OUT_PACKED_SYM  the records are PACKED-SYMMETRIC: RS = mk_record_stride_sym(n),
                                [ mean(n) | upper triangle by rows, n(n+1)/2 | sigma, detf | zero pad ],
                                640 B instead of 896 B per (model, step) at n = 10, 5632 B instead of 10752 B at
                                n = 36 (c_s = n + n(n+1)/2).  Pointer conventions as for
                                records with n*n replaced by n(n+1)/2: d_Pf = d_F + n, d_sigmas = d_F + n +
                                n(n+1)/2, ...  Element (r,c), r <= c, of a covariance sits at
                                r*n - r(r-1)/2 + (c - r) of its triangle.  Requires record_stride != 0.
                              MK_OUT_VAR_ONLY    mk_smooth / mk_filter_smooth write the smoothed state MEANS to
                                d_S [B,T,n] and the smoothed state VARIANCES (diagonals of the covariances) to
                                d_Ps [B,T,n], both dense (time_major applies), instead of smoothed records --
                                what Metran.get_state_means / get_state_variances / get_state consume
                                (metran.py:655-756).  d_F/d_Pf stay a (full or packed-symmetric) record array;
                                d_Xp/d_Pp must be NULL in mk_filter_smooth (filtered record only).
                              MK_OUT_TAPE        (ABI 5; mk_filter_smooth with d_sim_means / d_sim_vars only, shapes
                                with mk_tape_supported(N, K) = 1) d_F is not a filtered record array
                                but the BACKWARD TAPE of the inverse-free smoother, record_stride =
                                mk_tape_stride(N, K) = N (n + 4) doubles per (model, step), same (b, t) addressing
                                and time_major rule as records: per series one entry [ vector(n) | s0 | s1 | s2 | 0 ]
                                in the observable basis -- the gain, v/f, 1/f and the observation of an observed
                                series, T Pf z_u', the filtered observable, its variance and NaN for a series not
                                observed at that step.  The backward pass (Durbin-Koopman r / N recursion of the
                                sequential filter, kalmanfilter.py:341-378 walked backwards) then produces the same
                                projected smoothed means / variances as kalmansmoother + simulate (:403-476,
                                :569-603) without the pseudo-inverse of :455.  d_Pf, d_Xp, d_Pp, d_S, d_Ps must be
                                NULL; d_sigmas / d_detfs, if given, are DENSE [B,T] arrays.
                              MK_OUT_TAPE | MK_OUT_VAR_ONLY   (ABI 6; mk_filter_smooth, d_obsvar = NULL) the STATE tape:
                                record_stride = mk_state_tape_stride(N, K) = (N + K)(n + 4) doubles per (model, step) --
                                the N series entries above followed by K entries [ T Pf e_{N+k} (n) | x_f[N+k] |
                                Pf[N+k][N+k] | NaN | 0 ], the factor columns of the filtered covariance in the observable
                                basis.  The same backward pass then also writes the smoothed state MEANS to d_S [B,T,n]
                                and VARIANCES to d_Ps [B,T,n] (kalmansmoother's S and diag(Ps), :461-474; what
                                get_state_means / get_state_variances / get_state consume, metran.py:655-756) with no
                                filtered record, no LDL^T of the predicted covariance and no n x n product.
                                d_sim_means / d_sim_vars are optional; d_Pf, d_Xp, d_Pp must be NULL.             */
} mk_outputs;
#define MK_OUT_PACKED_SYM 1
#define MK_OUT_VAR_ONLY 2
#define MK_OUT_TAPE 4

/* ---- library / context ------------------------------------------------------------------ */
MK_API int mk_abi_version(void);
MK_API const char *mk_last_error(void);
MK_API int mk_device_count(int *count);
MK_API int mk_create(int device, mk_context **ctx);
MK_API int mk_destroy(mk_context *ctx);
/* Use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = default. */
MK_API int mk_set_stream(mk_context *ctx, void *hip_stream);
MK_API int mk_sync(mk_context *ctx);
/* Two shape classes have a second, equivalent kernel kept for A/B measurements (both tested against the oracle:
 * tests/test_smoother_variants.py, tests/test_hip_parity.py).  value 0 = the default.  There is no environment
 * variable and no switch that changes results.
 *   MK_VARIANT_SMOOTHER16     n <= 15, packed records: 0 smoother_record_kernel (DPP), 1 smoother_blk_kernel (4x4x4 MFMA)
 *   MK_VARIANT_WIDE_SMOOTHER  n > 16: 0 the blocked MFMA smoother (n <= 36: rows of A folded into the idle lanes),
 *                             1 smoother_wave_kernel (round 1, row per lane), 2 the MFMA smoother without the fold
 *   MK_VARIANT_WIDE_FILTER    n > 16, N <= 32: 0 by batch size -- filter_split_kernel (series on the lanes, factor block
 *                             replicated: 2 or 4 models per wavefront) for more than 2 instances per SIMD of the device,
 *                             filter_kernel<N,K,64> (one state per lane) below; 1 the latter always; 2 the former always
 *                             (the MK_OUT_TAPE path always runs the split kernel: the tape exists in that layout only)
 *   MK_VARIANT_SINGLE_RECORD  mk_filter with ONE record, at most MK_SPARSE_RECORD_MAX_INSTANCES instances, n <= 16 and both record
 *                             sets (the 7-tuple of seqkalmanfilter for a single Metran model): 0 the observed steps walked one
 *                             after the other and the records of the empty steps written in closed form by a second, parallel
 *                             kernel (kalmanfilter.py:335 skips the update on those steps; examples/data: 343 of 6255 steps
 *                             carry data), 1 the batched filter_kernel step by step
 *   MK_VARIANT_KERNEL_FAMILY  0 the specialised kernels where a shape has them, 1 the size-generic kernels (mk_generic.hip) for EVERY
 *                             shape (mk_filter / mk_loglik / mk_smooth / mk_filter_smooth with dense arrays or full-square records):
 *                             the second, independent implementation of the same recursions, for cross-checks and for timing
 *                             what specialisation buys
 *   MK_VARIANT_TAPE_FILTER    the writer of the backward tape (MK_OUT_TAPE), 16 < n, N <= 32: 0 filter_obs_kernel -- the filter run in the
 *                             observable basis, where gains and columns ARE the tape's entries (round 6) --, 1 filter_split_kernel OUT = 4
 *                             (state basis, every entry converted: round 4).  Same tape, same objective */
enum { MK_VARIANT_SMOOTHER16 = 0, MK_VARIANT_WIDE_SMOOTHER = 1, MK_VARIANT_WIDE_FILTER = 2, MK_VARIANT_SINGLE_RECORD = 3,
       MK_VARIANT_KERNEL_FAMILY = 4, MK_VARIANT_TAPE_FILTER = 5, MK_VARIANT_COUNT = 6 };
#define MK_SPARSE_RECORD_MAX_INSTANCES 16
MK_API int mk_set_kernel_variant(mk_context *ctx, int which, int value);
MK_API int mk_get_kernel_variant(mk_context *ctx, int which, int *value);
/* Tell the context that the CONTENTS of an observation buffer it has seen changed in place (same pointer): per-record
 * data derived from it (the observed-step list of the sparse objective, see mk_loglik) is rebuilt on the next call.
 * Not needed when a different buffer is passed. */
MK_API int mk_observations_changed(mk_context *ctx);
/* 1 if the library serves (N,K): a SPECIALISED kernel (compiled ahead of time, or a registered shape module: N + K <= 64,
 * fully unrolled over the state dimension -- the fast path) or, for every other shape with N + K <= mk_generic_max_states()
 * (= 128), the SIZE-GENERIC kernels of mk_generic.hip (one model per workgroup, covariance in LDS; correct for any shape,
 * tuned for none: the reference's loops are size-generic too, kalmanfilter.py:315-390, 453-474).  mk_shape_specialised tells
 * which.  Generic shapes run mk_filter / mk_loglik / mk_smooth / mk_filter_smooth with dense arrays or full-square records,
 * projection and MK_OUT_VAR_ONLY outputs; they have no packed-symmetric records, no tape and no adjoint gradient
 * (MK_ERR_SHAPE). */
MK_API int mk_shape_supported(int64_t N, int64_t K);
MK_API int mk_shape_specialised(int64_t N, int64_t K);
MK_API int64_t mk_generic_max_states(void);
/* Register a run-time shape module: a shared object built from metran_amd/csrc/mk_kernels.hip with
 * -DMK_SHAPE_MODULE '-DMK_SHAPES(X)=X(N,K)' (hipcc --offload-arch=gfx950).  The kernels are fully
 * unrolled over the state dimension, so a model shape outside the ahead-of-time list gets its own
 * specialised kernels this way (metran_amd/jit.py compiles, hazard-checks and caches them). */
MK_API int mk_register_shape_module(const char *path);
/* Doubles per packed record (see mk_outputs.record_stride) for state dimension n = N + K. */
MK_API int64_t mk_record_stride(int64_t n);
/* ... per PACKED-SYMMETRIC record (mk_outputs.flags & MK_OUT_PACKED_SYM). */
MK_API int64_t mk_record_stride_sym(int64_t n);
/* Doubles per (model, step) of the backward tape (mk_outputs.flags & MK_OUT_TAPE): N (N + K + 4); and whether the
 * tape path serves a shape (a specialised shape with 16 < N + K <= 63 -- the backward pass keeps the rows of its n x n matrix and
 * one more on the 64 lanes; written by the split filter of mk_split.hip for N <= 32 and by the lane-per-state filter beyond,
 * read by mk_dk.hip). */
MK_API int64_t mk_tape_stride(int64_t N, int64_t K);
MK_API int mk_tape_supported(int64_t N, int64_t K);
/* ... of the STATE tape (MK_OUT_TAPE | MK_OUT_VAR_ONLY): (N + K)(N + K + 4); served for the same shapes. */
MK_API int64_t mk_state_tape_stride(int64_t N, int64_t K);
/* Writes up to `cap` supported (N,K) pairs into shapes[2*i], shapes[2*i+1]; returns the count. */
MK_API int mk_supported_shapes(int64_t *shapes, int cap);

/* ---- device memory helpers (for hosts without torch) -------------------------------------- */
MK_API int mk_malloc(mk_context *ctx, size_t bytes, void **d_ptr);
MK_API int mk_free(mk_context *ctx, void *d_ptr);
MK_API int mk_memcpy_h2d(mk_context *ctx, void *d_dst, const void *h_src, size_t bytes);
MK_API int mk_memcpy_d2h(mk_context *ctx, void *h_dst, const void *d_src, size_t bytes);
MK_API int mk_memset(mk_context *ctx, void *d_dst, int value, size_t bytes);

/* ---- the hot path --------------------------------------------------------------------------- */
/* Metran._get_matrices restricted to the diagonals (metran/metran.py:246-322):
 *   phi = exp(-dt/alpha);  q_i = (1-phi_i^2)(1 - sum_k loadings[i,k]^2) for i<N, 1-phi_i^2 else. */
MK_API int mk_params_from_alpha(mk_context *ctx, int64_t B, int64_t R, int64_t N, int64_t K,
                                const double *d_alpha /* [B,n] */,
                                const double *d_loadings /* [R,N,K] */, double dt,
                                double *d_phi /* [B,n] */, double *d_q /* [B,n] */);

/* kalmansmoother in its literal 5-argument form for B models (kalmanfilter.py:403-476: filtered_state_means [B,T,n],
 * filtered_state_covariances [B,T,n,n], predicted_state_means, predicted_state_covariances, diag of transition_matrix [B,n])
 * -> d_S [B,T,n], d_Ps [B,T,n,n] (either may be NULL).  The predicted moments are READ, not recomputed (:454-474 use
 * predicted_state_covariances[t+1] and predicted_state_means[t+1] as handed in), so no transition covariance is needed and a
 * caller's own predicted moments are honoured.  Size-generic kernel (mk_generic.hip), n <= mk_generic_max_states(); dense
 * model-major arrays.  The fast path for moments that came out of mk_filter is mk_smooth / mk_filter_smooth. */
MK_API int mk_smooth_dense(mk_context *ctx, int64_t B, int64_t T, int64_t n, const double *d_phi, const double *d_F,
                           const double *d_Pf, const double *d_Xp, const double *d_Pp, double *d_S, double *d_Ps,
                           uint32_t *d_status);

/* ---- lock-step L-BFGS of the batched calibration (mk_lbfgs.hip) ----------------------------------
 * What ScipySolve.solve leaves to scipy's L-BFGS-B for ONE model (metran/solver.py:222-305: bounds alpha >= pmin, m history pairs,
 * pgtol on the projected gradient, ftol on the relative reduction) for R models at once, one thread per model.  Arrays are [R,n]
 * row-major doubles; every model has its own history ring d_Sh / d_Yh [history,R,n], d_rho [history,R] with d_hlen [R] live pairs
 * starting at slot d_hpos [R] (int32); masks are bytes.  A host pointer h_*, if given, receives the kernel's count after a stream
 * synchronisation (NULL: asynchronous).
 *   mk_lbfgs_direction  projected gradient d_pg, d_active &= max|pg| > gtol, two-loop recursion, steepest descent where the
 *                       direction is not a descent direction, unit-scale first step, zero on active bounds -> d_d; #active.
 *                       With d_phase / d_step / d_nback (own line search per model): a model with phase 1 keeps direction and step,
 *                       the others start a new search (step 1, nback 0, phase 1)
 *   mk_lbfgs_trial      d_xt = max(x + step d, lo); d_xe = searching ? xt : x_new
 *   mk_lbfgs_armijo     searching models: accept (x_new, f_new <- xt, ft) if ft <= f + 1e-4 pg.(xt - x) and ft finite, else
 *                       step *= clamp(parabola minimiser, 0.1, 0.5); #still searching.  Lock-step form (d_nback NULL): an
 *                       accepted model leaves d_searching.  Own-line-search form: d_accepted marks the accepted models (they stay
 *                       in d_searching = the active mask), a model that has used max_backtracks trial points leaves it; #accepted
 *   mk_lbfgs_update     for the models of d_mask (NULL: all): pair (s, y, rho) of the accepted point into the model's ring if
 *                       s.y > 1e-10 y.y (skipped otherwise, as scipy does), (x, f, g) <- new (lock-step form: models still in
 *                       d_searching keep their old gradient if asked and leave d_active), d_active &= relative reduction > ftol,
 *                       phase <- 0; #usable pairs.  d_nit [R] int32 (may be NULL): the quasi-Newton ITERATIONS every model has
 *                       taken -- incremented for each active model updated here -- and a model whose count reaches maxiter
 *                       (> 0) leaves d_active: scipy's maxiter (solver.py:248 passes it through) counts iterations of ONE
 *                       model, whatever the driver's loop counts */
MK_API int mk_lbfgs_direction(mk_context *ctx, int64_t R, int64_t n, int64_t history, const double *d_x, const double *d_g, const double *d_lo,
                              uint8_t *d_active, const double *d_Sh, const double *d_Yh, const double *d_rho, const int *d_hlen,
                              const int *d_hpos, double gtol, double *d_pg, double *d_d, uint8_t *d_phase, double *d_step, int *d_nback,
                              int *h_nactive);
MK_API int mk_lbfgs_trial(mk_context *ctx, int64_t R, int64_t n, const double *d_x, const double *d_d, const double *d_step,
                          const double *d_lo, const uint8_t *d_searching, const double *d_x_new, double *d_xt, double *d_xe);
MK_API int mk_lbfgs_armijo(mk_context *ctx, int64_t R, int64_t n, const double *d_ft, const double *d_f, const double *d_pg,
                           const double *d_xt, const double *d_x, uint8_t *d_searching, double *d_step, double *d_x_new,
                           double *d_f_new, int *d_nback, int64_t max_backtracks, uint8_t *d_accepted, int *h_nsearching,
                           int *h_naccepted);
MK_API int mk_lbfgs_update(mk_context *ctx, int64_t R, int64_t n, int64_t history, double *d_x, double *d_f, double *d_g,
                           const double *d_x_new, const double *d_f_new, const double *d_g_new, int keep_old_gradient_if_searching,
                           const uint8_t *d_searching, const uint8_t *d_mask, uint8_t *d_active, double ftol, double *d_Sh,
                           double *d_Yh, double *d_rho, int *d_hlen, int *d_hpos, uint8_t *d_phase, int *d_nit, int64_t maxiter,
                           int *h_ngood);

/* seqkalmanfilter + get_mle for B instances (kalmanfilter.py:236-400, 550-567).
 * Uses d_mle, d_sigmas, d_detfs, d_sigmacount, d_F, d_Pf, d_Xp, d_Pp, d_status of `out`. */
MK_API int mk_filter(mk_context *ctx, const mk_problem *prob, const mk_outputs *out);

/* -2 log L only: mk_filter with every state output NULL (the solver's objective,
 * Metran.get_mle, metran/metran.py:605-622).  d_mle [B] required. */
MK_API int mk_loglik(mk_context *ctx, const mk_problem *prob, double *d_mle);
/* (When every instance shares ONE record -- n_records == 1, the solver's finite-difference points -- and
 * N+K <= 16, mk_loglik walks only the record's observed steps and applies the runs of empty steps in closed
 * form; real Metran records are sparse: examples/data observes 343 of 6255 daily steps.  The list of observed steps is
 * built once per record (pointer, shape, layout) and reused by the following calls: mk_observations_changed.) */

/* Objective AND its gradient in two launches (the reference has no gradient: scipy differences P+1
 * objective evaluations, metran/solver.py:248-255).  Forward: mk_filter writing only the filtered
 * records into d_work (n_instances*T*mk_record_stride(n) doubles; time_major as in mk_outputs);
 * backward: the adjoint kernel re-reads them once.  d_gphi / d_gq [B,n] receive
 * d(-2 log L)/d diag(Phi) and /d diag(Q); d_mle [B], d_sigmacount [B] as in mk_filter; d_status may be
 * NULL.  N+K <= 16: four models per wavefront (adjoint_kernel); 16 < N+K <= 64: one model per wavefront
 * (adjoint_wide_kernel, mk_split.hip) -- at configs[3]'s shape one gradient instead of 37 differenced filter runs. */
MK_API int mk_loglik_grad(mk_context *ctx, const mk_problem *prob, double *d_work, int time_major,
                          double *d_mle, int64_t *d_sigmacount, double *d_gphi, double *d_gq,
                          uint32_t *d_status);
/* The two launches of mk_loglik_grad separately.  A line search evaluates the objective at several trial points and
 * needs the gradient at the accepted one only: with MK_GRAD_FORWARD every trial is the recording forward pass (d_mle,
 * d_sigmacount, d_status, the records in d_work; d_gphi / d_gq may be NULL), and MK_GRAD_BACKWARD then walks the records
 * the LAST forward pass left in d_work -- same prob (parameters included), same d_work, same d_sigmacount -- instead of
 * filtering the accepted point once more.  phases = MK_GRAD_FORWARD | MK_GRAD_BACKWARD is mk_loglik_grad. */
enum { MK_GRAD_FORWARD = 1, MK_GRAD_BACKWARD = 2 };
MK_API int mk_loglik_grad_phases(mk_context *ctx, const mk_problem *prob, double *d_work, int time_major,
                                 double *d_mle, int64_t *d_sigmacount, double *d_gphi, double *d_gq,
                                 uint32_t *d_status, int phases);

/* Wide models (16 < N + K <= 64), optional: an UPDATE TAPE for the adjoint gradient (round 6).  mk_adjoint_update_stride(N, K)
 * doubles per (model, step) -- N slots [ d = P z_j' (n) | pad | 1/f, v ], one per scalar update of the step -- or 0 where the
 * shape has none (n <= 16: the 16-lane kernel recomputes).  With a caller-owned buffer of at least n_instances * T * stride
 * doubles on the context (mk_set_adjoint_updates; NULL detaches), the recording forward pass of mk_loglik_grad /
 * mk_loglik_grad_phases writes the slots (one model per wavefront, whatever the batch size) and the backward walk READS them
 * instead of recomputing every step from the filtered record of the step before -- 38 % of its instructions, and the walk is
 * one wavefront's dependent chain (512 x (32,4), T = 500: backward pass 12.8 -> see DESIGN.md section 6).  Same gradient to
 * rounding.  The buffer must stay valid and unchanged between the two phases of a gradient; too small a buffer for a call
 * is simply not used. */
MK_API int64_t mk_adjoint_update_stride(int64_t N, int64_t K);
MK_API int mk_set_adjoint_updates(mk_context *ctx, double *d_buf, int64_t capacity_doubles);
/* Chain rule of mk_params_from_alpha: d/dalpha = (gphi - 2 phi c gq) phi dt / alpha^2, c = 1 - sum_k
 * loadings^2 for the series, 1 for the factors (metran/metran.py:246-322). */
MK_API int mk_alpha_grad(mk_context *ctx, int64_t B, int64_t R, int64_t N, int64_t K,
                         const double *d_alpha /* [B,n] */, const double *d_loadings /* [R,N,K] */,
                         double dt, const double *d_gphi, const double *d_gq, double *d_galpha /* [B,n] */);

/* kalmansmoother for B instances (kalmanfilter.py:403-476).  Reads out->d_F and out->d_Pf
 * (as written by mk_filter); predicted moments are recomputed from them (Phi diagonal), so
 * d_Xp/d_Pp are not read.  Writes d_S, d_Ps (either may be NULL), d_status. */
MK_API int mk_smooth(mk_context *ctx, const mk_problem *prob, const mk_outputs *out);

/* run_smoother (kalmanfilter.py:676-694): mk_filter then mk_smooth on the same stream. */
MK_API int mk_filter_smooth(mk_context *ctx, const mk_problem *prob, const mk_outputs *out);

/* SPKalmanFilter.simulate (kalmanfilter.py:569-603) for B instances:
 *   sim_means[b,t,:] = Z_b x[b,t];  sim_vars[b,t,j] = max((Z_b P[b,t] Z_b^T)_jj, 0)
 * Z is dense [RZ,N,n] (instance b uses Z[b % RZ]); Metran passes the std-scaled matrix
 * (metran/metran.py:944-961). */
MK_API int mk_simulate(mk_context *ctx, int64_t B, int64_t RZ, int64_t T, int64_t N, int64_t n,
                       const double *d_Z, const double *d_means /* [B,T,n] */,
                       const double *d_covs /* [B,T,n,n] */, double *d_sim_means /* [B,T,N] */,
                       double *d_sim_vars /* [B,T,N] */);

/* SPKalmanFilter.decompose (kalmanfilter.py:605-644):
 *   sdf[b,t,:] = Z[:, :N] x[b,t,:N];  cdf[b,k,t,:] = Z[:, N+k] * x[b,t,N+k] */
MK_API int mk_decompose(mk_context *ctx, int64_t B, int64_t RZ, int64_t T, int64_t N, int64_t n,
                        const double *d_Z, const double *d_means /* [B,T,n] */,
                        double *d_sdf /* [B,T,N] */, double *d_cdf /* [B,n-N,T,N] */);

/* Sum of d_mle over the local batch in a fixed order (deterministic tree), for the summed
 * objective fed back to the solver; mk_allreduce_sum below combines the ranks. */
MK_API int mk_sum(mk_context *ctx, int64_t count, const double *d_values, double *d_result /* [1] */);

/* ---- the one collective of the path (SURVEY.md 8b "mk_allreduce_sum(ctx, buf, count) (RCCL) or accept an external
 * communicator", 8e): one process per GPU, every model independent, so the only exchange is the summed objective of a
 * shared-parameter calibration -- 1 double, or P + 1 with the adjoint gradient -- fed back to the solver
 * (metran/solver.py:42-63 has one process and no counterpart).  The communicator is RCCL's (ncclComm_t, passed as void *
 * so that this header needs no rccl.h); librccl is bound with dlopen at the first of these calls, preferring a copy the
 * process has already loaded (e.g. PyTorch-ROCm's), so a single-GPU caller never maps it.
 *   mk_comm_set_library    (optional, before the first call below) the path of the librccl.so to bind; the library reads
 *                          no environment variable
 *   mk_comm_unique_id      ncclGetUniqueId into a caller buffer of 128 bytes: ONE rank calls it and hands the bytes to
 *                          the others by whatever side channel the job has (a file, MPI, torch.distributed's store)
 *   mk_comm_init_rank      ncclCommInitRank for this context's device; collective (every rank calls it with the same
 *                          id); the context owns the communicator and destroys it in mk_comm_destroy / mk_destroy
 *   mk_set_communicator    use a communicator the CALLER created and keeps owning (a raw librccl one); NULL detaches
 *   mk_allreduce_sum       in-place all-reduce(sum, float64) of d_buf[count] over the communicator's ranks, ordered on
 *                          the context's stream (mk_set_stream); asynchronous like every launch (mk_sync to wait).
 *                          Without a communicator it FAILS (MK_ERR_INVALID): there is no silent single-rank shortcut --
 *                          a group of one rank still runs the collective, as on eight */
MK_API int mk_comm_set_library(const char *path);
MK_API int mk_comm_unique_id(void *id128 /* [128 bytes] */);
MK_API int mk_comm_init_rank(mk_context *ctx, int nranks, int rank, const void *id128);
MK_API int mk_set_communicator(mk_context *ctx, void *nccl_comm);
MK_API int mk_comm_destroy(mk_context *ctx);
MK_API int mk_allreduce_sum(mk_context *ctx, double *d_buf, int64_t count);

/* ---- observation ingestion (what precedes the filter; batched) ------------------------------ */
/* Metran.standardize (metran/metran.py:102-121) for R models: per series, subtract the mean and
 * divide by the standard deviation (pandas semantics: NaN skipped, ddof = 1).  d_in / d_out are
 * [R,T,N] (or [T,R,N] when time_major != 0); d_out may equal d_in or be NULL (statistics only);
 * d_mean / d_std [R,N] may be NULL.  N <= 64. */
MK_API int mk_standardize(mk_context *ctx, int64_t R, int64_t T, int64_t N, int time_major,
                          const double *d_in, double *d_out, double *d_mean, double *d_std);

/* Metran.mask_observations (metran/metran.py:464-494): d_out[i] = d_mask[i] ? NaN : d_obs[i] for
 * `count` observations in whatever layout d_obs has (d_mask: one byte each, non-zero = hide).
 * Unmasking is re-running the filter on the untouched d_obs: no re-upload. */
MK_API int mk_mask_observations(mk_context *ctx, int64_t count, const double *d_obs,
                                const unsigned char *d_mask, double *d_out);

/* SPKalmanFilter.set_observations (metran/kalmanfilter.py:646-674) for R models: the reference's
 * packed arrays from NaN-encoded observations d_obs [R,T,N] (model-major): d_observations [R,T,N]
 * (missing -> 0.0), d_indices [R,T,N] (doubles holding ints, left-packed), d_count [R,T] (int64).
 * Any output may be NULL.  The filter kernels read d_obs directly; this exists for callers that
 * want the reference's representation (the 9-argument engine callable). */
MK_API int mk_pack_observations(mk_context *ctx, int64_t R, int64_t T, int64_t N, const double *d_obs,
                                double *d_observations, double *d_indices, int64_t *d_count);

/* ---- batched factor analysis (what produces the loadings; metran/factoranalysis.py) ----------- */
/* FactorAnalysis._get_correlations (factoranalysis.py:404-418; pandas DataFrame.corr, pearson, pairwise
 * complete): d_corr [R,N,N] from NaN-encoded d_obs [R,T,N] ([T,R,N] when time_major != 0).  N <= 64. */
MK_API int mk_fa_correlation(mk_context *ctx, int64_t R, int64_t T, int64_t N, int time_major,
                             const double *d_obs, double *d_corr);
/* _get_eigval (:420-460), _maptest (:220-312), the factor-count rules of solve (:66-82; maxfactors <= 0 =
 * None) and the start vector of _minres (:188-203) for B correlation matrices.  d_eigval [B,N] descending,
 * clipped at 0; d_nfactors / d_nfactors_map / d_nfactors_map4 [B]; d_psi0 [B,N]; d_status [B]: 1 = no factors
 * can be derived (NaN correlation or singular matrix; the reference returns None).  Outputs may be NULL.
 * "Singular" is decided on the eigenvalues of the device's own decomposition: smallest eigenvalue <= 1e-14 x the largest.
 * DIVERGENCE from the reference: numpy.linalg.inv (factoranalysis.py:195) raises only on an EXACTLY zero pivot, so a
 * correlation matrix with a condition number above ~1e14 that is not exactly singular (nearly collinear series) makes
 * the reference proceed with an inverse that has no correct digits, while this routine reports status 1 (no factors). */
MK_API int mk_fa_analyse(mk_context *ctx, int64_t B, int64_t N, int64_t maxfactors, const double *d_corr,
                         double *d_eigval, int64_t *d_nfactors, int64_t *d_nfactors_map, int64_t *d_nfactors_map4,
                         double *d_psi0, uint32_t *d_status);
/* _minresfun (:315-347), _minresgrad (:349-373) and _get_loadings (:375-401) for B vectors d_psi [B,N];
 * instance b uses correlation matrix and factor count b % R.  d_fval [B], d_grad [B,N], d_loadings [B,N,KMAX]
 * (columns >= nfactors zero); any output may be NULL.
 * d_order [B,KMAX] (may be NULL): _get_loadings takes eigvec[:, :nf] in the order numpy.linalg.eig (LAPACK dgeev)
 * returns the pairs of psi^-1/2 S psi^-1/2 (:396-398), which is NOT sorted; d_order[b][f] is the rank (0 = largest
 * eigenvalue, < N) of the pair that becomes column f of instance b.  NULL = ranks 0 .. nf-1 (the nf largest,
 * descending).  The jacobian depends on it through the loadings; the objective does not. */
MK_API int mk_fa_minres(mk_context *ctx, int64_t B, int64_t R, int64_t N, int64_t KMAX, const double *d_corr,
                        const int64_t *d_nfactors, const double *d_psi, const int64_t *d_order, double *d_fval,
                        double *d_grad, double *d_loadings);
/* Communality normalisation + varimax rotation (_rotate, :121-171; gamma = 1, maxiter = 20, tol = 1e-6 in the
 * reference) + sign convention of FactorAnalysis.solve (:84-108), in place on d_loadings [B,N,KMAX]. */
MK_API int mk_fa_rotate(mk_context *ctx, int64_t B, int64_t N, int64_t KMAX, const int64_t *d_nfactors,
                        double *d_loadings, double gamma, int maxiter, double tol);
/* Symmetric eigen-decomposition of B matrices d_sym [B,N,N] (cyclic Jacobi): d_val [B,N] descending, d_vec
 * [B,N,N] eigenvectors in columns (may be NULL).  The one dense-linear-algebra routine of the kernels above. */
MK_API int mk_fa_eigh(mk_context *ctx, int64_t B, int64_t N, const double *d_sym, double *d_val, double *d_vec);

/* ---- instrumentation --------------------------------------------------------------------- */
/* When enabled, every kernel launch is bracketed by hipEvents on the context's stream.  enable = 1: the most recent
 * launch of each kind is kept (mk_last_kernel_ms); enable = 2: EVERY launch keeps its own event pair until
 * mk_kernel_ms_totals collects them -- no host synchronisation inside a timed loop. */
MK_API int mk_enable_timing(mk_context *ctx, int enable);
/* Duration of the most recent filter / smoother kernel (ms); synchronises on their events. */
MK_API int mk_last_kernel_ms(mk_context *ctx, float *filter_ms, float *smoother_ms);
/* enable = 2: summed duration (ms) and number of the filter (or objective) and smoother launches since the previous
 * call (or since timing was enabled); synchronises on their events and recycles them. */
MK_API int mk_kernel_ms_totals(mk_context *ctx, double *filter_ms, int64_t *filter_launches, double *smoother_ms,
                               int64_t *smoother_launches);

#ifdef __cplusplus
}
#endif
#endif /* METRAN_HIP_H */
