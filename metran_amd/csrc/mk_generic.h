// mk_generic.h -- the size-generic kernels (mk_generic.hip): declarations shared with the C ABI (mk_capi.hip).  Not part of
// the run-time shape modules (metran_amd/jit.py does not hash this file).
#pragma once
#include "mk_internal.h"

#define MK_GENERIC_MAX_STATES 128 /* n = N + K served by the generic kernels (the filter keeps P, n^2 doubles, in LDS) */

namespace mk {

struct GenericSmootherArgs {
    SmootherArgs a;          // as for the specialised smoothers (F / Pf inputs, S / Ps / VAR_ONLY / projection outputs)
    int N, K;
    const double *Xp, *Pp;   // NULL: predicted moments recomputed from (F, Pf, phi, q); else the CALLER's, dense [B,T,n] / [B,T,n,n]
                             // in the (bs, ts) addressing of `a` (kalmansmoother's 5-argument form, kalmanfilter.py:403-476)
    double *ws;              // workspace, generic_smoother_ws_doubles(B, n) doubles
};

size_t generic_filter_lds_bytes(int N, int K);
size_t generic_smoother_ws_doubles(long B, int n);
hipError_t launch_filter_generic(int N, int K, const FilterArgs &a, hipStream_t s);
hipError_t launch_smoother_generic(const GenericSmootherArgs &g, hipStream_t s);

} // namespace mk
