// mk_internal.h -- shared between the kernels (mk_kernels.hip) and the C ABI (mk_capi.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "metran_hip.h"

// (N series, K common factors) shapes compiled ahead of time.  The first is the benchmark
// shape of BASELINE.json (configs[1], configs[2]); (32,4) is configs[3]; (5,1) is the
// examples/data model (configs[0]); the rest cover the golden fixtures and common small models.
#ifndef MK_SHAPES
#define MK_SHAPES(X) \
    X(8, 2)          \
    X(5, 1)          \
    X(2, 1)          \
    X(3, 1)          \
    X(4, 1)          \
    X(6, 2)          \
    X(14, 3)         \
    X(32, 4)
#endif

namespace mk {

// backward tape of the wide models (MK_OUT_TAPE; mk_split.hip writes it, mk_dk.hip reads it).  Block of one (model, step):
// N entries of XS doubles, entry j = [ series part of the vector (N) | side row: factor part (K), s0, s1, s2, 0 ] -- the series
// part of entry j starts at j * tape_xs_c, its side row at tape_so_c + j * tape_ss_c.  (Round 4 also measured the split
// form -- an N x N array of series parts followed by an N x SW array of side rows -- and kept the contiguous entries.)
constexpr int tape_side_c(int K) { return K + 4; }
constexpr int tape_xs_c(int N, int K) { return N + tape_side_c(K); }
constexpr int tape_ss_c(int N, int K) { return tape_xs_c(N, K); }
constexpr int tape_so_c(int N, int K) { return N; }
constexpr int tape_stride_c(int N, int K) { return N * tape_xs_c(N, K); }
// STATE tape (MK_OUT_TAPE | MK_OUT_VAR_ONLY, round 5): K more entries per block, the factor columns of the filtered
// covariance in the observable basis -- entry N + k = [ T Pf e_{N+k} (n) | x_f[N+k] | Pf[N+k][N+k] | NaN | 0 ] -- from which the
// backward pass also gets the smoothed STATE means and variances (mk_dk.hip, STATE = true)
constexpr int state_tape_stride_c(int N, int K) { return (N + K) * tape_xs_c(N, K); }
// update tape of the wide adjoint gradient (round 6): N slots of (n rounded up to even) + 2 doubles per (model, step)
constexpr int adjoint_update_slot_c(int N, int K) { return ((N + K + 1) & ~1) + 2; }
constexpr int adjoint_update_stride_c(int N, int K) { return N * adjoint_update_slot_c(N, K); }

struct FilterArgs {
    long B, R, T, warmup;
    long bs, ts;         // state/bookkeeping outputs: block (b, t) at index b*bs + t*ts
    long rs;             // > 0: packed records of rs doubles (Xp/F are the record arrays), 0: dense arrays
    long sym;            // records are packed-symmetric (MK_OUT_PACKED_SYM): rs = record_stride_sym(n)
    long sig_stride;     // element stride of sigmas/detfs entries (1 for dense [B,T] arrays)
    long obs_bs, obs_ts; // observations: record (r, t) at row r*obs_bs + t*obs_ts
    const double *obs, *phi, *q, *loadings, *obsvar, *x0, *P0;
    double *mle, *sigmas, *detfs;
    long long *sigmacount;
    double *F, *Pf, *Xp, *Pp;
    unsigned *status;
    long variant;        // wide models: 0 = split layout when B > 2 x #SIMDs, else filter_kernel<N,K,64> (one state per lane); 1 = one state per lane always; 2 = split always
    long tape;           // 1: F is the backward tape of the inverse-free smoother (MK_OUT_TAPE; rs = tape_stride(N, K));
                         // 2: the STATE tape (MK_OUT_TAPE | MK_OUT_VAR_ONLY; rs = state_tape_stride(N, K))
    double *upd;         // OUT = 3, one model per wavefront (16 < n): the UPDATE TAPE of the adjoint gradient (round 6), or NULL.  Per
    long us;             // (model, step) a block of us = adjoint_update_stride_c(N, K) doubles at (b*bs + t*ts)*us: slot u = the u-th scalar
                         // update of the step in ascending series order, [ d = P z_j' (n) | pad | 1/f, v ] -- what adjoint_wide_kernel
                         // otherwise recomputes from the filtered record of step t - 1
    long tape_basis;     // tape, N <= 32: 0 = filter_obs_kernel (the filter in the observable basis, round 6), 1 = filter_split_kernel OUT = 4
                         // (state basis, round 4) -- mk_set_kernel_variant(ctx, MK_VARIANT_TAPE_FILTER, .)
};

// Timing experiments that skip phases of a kernel (and so produce wrong numbers) exist only in builds made with
// -DMK_TUNE=<mask> (make EXTRA=-DMK_TUNE=5 OUT=...): a compile-time constant of a separate library.  The shipped library
// has no such switch and reads no environment variable.
#ifdef MK_TUNE
#define MK_TUNE_SKIP(args, bit) (((MK_TUNE) & (bit)) != 0)
#else
#define MK_TUNE_SKIP(args, bit) false
#endif

struct SmootherArgs {
    long B, T;
    long bs, ts;
    long rs;             // > 0: F and S are packed-record arrays (Pf = F + n, Ps = S + n)
    long sym;            // records are packed-symmetric
    double *state_means, *state_vars; // MK_OUT_VAR_ONLY: [., n] smoothed state means / variances (dense)
    long R;              // observation records (loadings / scale / offset are per record)
    const double *loadings, *scale, *offset; // fused projection epilogue (record kernel, optional)
    double *sim_means, *sim_vars;            // [.,N] per (b,t), same (bs, ts) addressing
    const double *phi, *q;
    const double *F, *Pf;
    double *S, *Ps;
    unsigned *status;
    long variant;        // bit 0: n <= 15 records -> smoother_blk_kernel; bit 1: n > 16 -> smoother_wave_kernel (round 1); bit 2: n > 16 -> the MFMA kernel without the lane fold; bits 3 / 4: (-DMK_EXPERIMENTAL_BLK4 builds only) n > 16, n % 4 == 0 -> the 4x4x4 MFMA block path with / without the lane fold;
                         // set by the C ABI from mk_set_kernel_variant.  Every variant is a tested, equivalent kernel.
    long tape;           // 1: F is the backward tape (MK_OUT_TAPE, rs = tape_stride(N, K)): smoother_dk_kernel (mk_dk.hip);
                         // 2: the STATE tape (rs = state_tape_stride(N, K)): state_means / state_vars [., n] are written too
    const double *obsvar; // tape path: observation variances [R,N] or NULL = zeros
};

struct AdjointArgs {
    long B, R, T, warmup;
    long bs, ts, rs;     // filtered records: (b, t) at (b*bs + t*ts)*rs doubles, rs = record_stride(n)
    long obs_bs, obs_ts;
    const double *obs, *phi, *q, *loadings, *obsvar, *x0, *P0;
    const double *F;                 // filtered record array written by the forward filter
    const long long *sigmacount;     // [B] observed steps per instance (forward filter)
    double *gphi, *gq;               // [B,n] gradient of -2 log L w.r.t. diag(Phi), diag(Q)
    const double *upd;               // update tape written by the recording forward pass (FilterArgs.upd), or NULL: recompute
    long us;                         // its stride per (model, step), adjoint_update_stride_c(N, K)
};

struct SparseArgs { // objective of ONE record (all instances share it), observed steps only
    long B, T, warmup;
    long ostep;                      // doubles between consecutive steps of the record
    const double *obs;               // the record: step t at obs + t*ostep
    const double *phi, *q, *loadings, *obsvar, *x0, *P0;
    int *tlist;                      // workspace [T+1]: count, then the observed steps (built by the launch)
    int rebuild;                     // 0: tlist still describes this record (mk_capi caches it per uploaded record)
    double *mle;
    unsigned *status;
    // record outputs (round 5, the single-record engine route: loglik_sparse_kernel<.., REC = true> + fill_gaps_kernel):
    // F / Xp = filtered / predicted record arrays of rs = record_stride(n) doubles per (instance, step), block (b, t) at
    // (b*bs + t*ts)*rs; NULL = objective only
    double *F, *Xp;
    long rs, bs, ts;
    long long *sigmacount;
};

hipError_t launch_filter(int N, int K, const FilterArgs &a, hipStream_t s);
// mk_split.hip: wide models, N series on the lanes + replicated factor block (hipErrorNotSupported: not served)
hipError_t launch_filter_split(int N, int K, const FilterArgs &a, hipStream_t s);
hipError_t launch_adjoint_wide(int N, int K, const AdjointArgs &a, hipStream_t s); // mk_split.hip: n > 16
hipError_t launch_sparse(int N, int K, const SparseArgs &a, hipStream_t s);
hipError_t launch_adjoint(int N, int K, const AdjointArgs &a, hipStream_t s);
hipError_t launch_alpha_grad(long B, long R, int N, int K, const double *alpha, const double *loadings, double dt,
                             const double *gphi, const double *gq, double *galpha, hipStream_t s);
hipError_t launch_smoother(int N, int K, const SmootherArgs &a, hipStream_t s);
hipError_t launch_smoother_wide(int N, int K, const SmootherArgs &a, hipStream_t s); // mk_wide.hip (n > 16)
hipError_t launch_smoother_dk(int N, int K, const SmootherArgs &a, hipStream_t s);   // mk_dk.hip (a.tape)
int record_stride(int n); // doubles per packed record for state dimension n
int record_stride_sym(int n); // ... per packed-symmetric record
int num_shapes();
void get_shape(int i, int *N, int *K);
hipError_t launch_params(long B, long R, int N, int K, const double *alpha, const double *loadings, double dt,
                         double *phi, double *q, hipStream_t s);
hipError_t launch_simulate(long B, long RZ, long T, int N, int n, const double *Z, const double *means,
                           const double *covs, double *sm, double *sv, hipStream_t s);
hipError_t launch_decompose(long B, long RZ, long T, int N, int n, const double *Z, const double *means,
                            double *sdf, double *cdf, hipStream_t s);
hipError_t launch_sum(long count, const double *v, double *out, hipStream_t s);
// mk_ingest.hip
hipError_t launch_standardize(long R, long T, int N, int time_major, const double *in, double *out, double *mean,
                              double *stdev, hipStream_t s);
hipError_t launch_mask(long count, const double *obs, const unsigned char *mask, double *out, hipStream_t s);
hipError_t launch_pack(long RT, int N, const double *obs, double *observations, double *indices, long *count,
                       hipStream_t s);
// mk_factor.hip
hipError_t launch_fa_corr(long R, long T, int N, int time_major, const double *obs, double *corr, hipStream_t s);
hipError_t launch_fa_analyse(long B, int N, long maxfactors, const double *corr, double *eigval, long long *nfact,
                             long long *nfact_map, long long *nfact_map4, double *psi0, unsigned *status, hipStream_t s);
hipError_t launch_fa_minres(long B, long R, int N, int KMAX, const double *corr, const long long *nfact, const double *psi,
                            const long long *order, double *fval, double *grad, double *loadings, hipStream_t s);
hipError_t launch_fa_rotate(long B, int N, int KMAX, const long long *nfact, double *loadings, double gamma, int maxiter,
                            double tol, hipStream_t s);
hipError_t launch_fa_eigh(long B, int N, const double *sym, double *val, double *vec, hipStream_t s);

} // namespace mk
