// mk_lbfgs.hip -- the vector algebra of the batched calibration driver as four small kernels.
// Reference: ScipySolve.solve hands Metran's objective to scipy's L-BFGS-B (/root/reference/metran/solver.py:222-305: bounds
// alpha >= pmin, m = 10 history pairs, ftol = factr * epsmch, pgtol on the projected gradient); calibrate_batch
// (metran_amd/calibrate.py) runs R such optimisations in lock-step on the device.  Until round 5 its two-loop recursion, trial
// points, Armijo test and history update were ~130 separate torch operations and three to five host synchronisations per
// iteration -- at 8192 models two thirds of the wall time of a calibration whose filter / adjoint kernels take 0.3 s
// (scripts/probe.py calibrate --trace).  Here every model is ONE thread and an iteration is four launches:
//     lbfgs_direction_kernel   projected gradient, convergence test on it, two-loop recursion over the model's history ring,
//                              descent and bound safeguards (models in the middle of a line search keep theirs) -> pg, d, #active
//     lbfgs_trial_kernel       x + step d projected on the bounds; settled models stay at their accepted point   -> xt, xe
//     lbfgs_armijo_kernel      sufficient-decrease test, acceptance, next step length from the parabola         -> #searching
//                              (own-line-search form: back-tracking budget per model, accepted mask             -> #accepted)
//     lbfgs_update_kernel      the new (s, y, rho) pair into the model's ring if it is usable (scipy skips it otherwise),
//                              (x, f, g) <- accepted point, scipy's relative-reduction test
// Every model has its OWN ring (hlen, hpos): models accept steps in different iterations when each runs its own line search.
// The arithmetic is that of the torch code it replaces (tests/oracle_engine.py keeps it as the restatement the CPU tests and
// tests/test_lbfgs_gpu.py compare with), sums taken in index order.  All arrays are [R, n] row-major (the ring [H, R, n]); masks
// are bytes; n <= MK_LBFGS_MAX_N.
#include <hip/hip_runtime.h>

#include "mk_lbfgs.h"

namespace mk {

namespace {
constexpr int TPB = 128;
}

__global__ void __launch_bounds__(TPB) lbfgs_direction_kernel(LbfgsArgs a)
{
    const long r = (long)blockIdx.x * TPB + threadIdx.x;
    if (r >= a.R) return;
    const int n = a.n, H = a.H;
    if (a.phase && a.phase[r] && a.active[r]) { // in the middle of its line search: direction, slope and step stay
        atomicAdd(a.counters + 0, 1);
        return;
    }
    const int len = a.hlen[r], pos = a.hpos[r];
    const double *x = a.x + r * n, *g = a.g + r * n, *lo = a.lo + r * n;
    double *pg = a.pg + r * n, *d = a.d + r * n;
    double q[MK_LBFGS_MAX_N], al[MK_LBFGS_MAX_H];
    double pgmax = 0.0;
    for (int c = 0; c < n; ++c) { // gradient with the components pushing into an active bound removed
        const double v = (x[c] <= lo[c] && g[c] > 0.0) ? 0.0 : g[c];
        pg[c] = v;
        q[c] = v;
        pgmax = fmax(pgmax, fabs(v)); // (a NaN component leaves pgmax alone: NaN > gtol is false below only if all are NaN)
    }
    bool nanpg = false;
    for (int c = 0; c < n; ++c) nanpg = nanpg || (q[c] != q[c]);
    // torch: active &= pg.abs().amax(1) > gtol  (amax propagates NaN, and NaN > gtol is false)
    const bool act = a.active[r] && !nanpg && (pgmax > a.gtol);
    a.active[r] = act ? 1 : 0;
    if (!act) {
        for (int c = 0; c < n; ++c) d[c] = 0.0;
        return;
    }
    atomicAdd(a.counters + 0, 1);
    if (a.step) a.step[r] = 1.0;  // a new line search starts at the unit step
    if (a.nback) a.nback[r] = 0;
    if (a.phase) a.phase[r] = 1;
    // two-loop recursion; pair i = 0 is the NEWEST: slot (pos + len - 1 - i) mod H
    for (int i = 0; i < len; ++i) {
        const int slot = (pos + len - 1 - i) % H;
        const double *s = a.Sh + ((long)slot * a.R + r) * n, *y = a.Yh + ((long)slot * a.R + r) * n;
        double dot = 0.0;
        for (int c = 0; c < n; ++c) dot += s[c] * q[c];
        const double ai = a.rho[(long)slot * a.R + r] * dot;
        al[i] = ai;
        for (int c = 0; c < n; ++c) q[c] -= ai * y[c];
    }
    if (len > 0) {
        const int slot = (pos + len - 1) % H;
        const double *s = a.Sh + ((long)slot * a.R + r) * n, *y = a.Yh + ((long)slot * a.R + r) * n;
        double sy = 0.0, yy = 0.0;
        for (int c = 0; c < n; ++c) {
            sy += s[c] * y[c];
            yy += y[c] * y[c];
        }
        const double gamma = sy / fmax(yy, 1e-300);
        for (int c = 0; c < n; ++c) q[c] *= gamma;
    }
    for (int i = len - 1; i >= 0; --i) { // oldest first
        const int slot = (pos + len - 1 - i) % H;
        const double *s = a.Sh + ((long)slot * a.R + r) * n, *y = a.Yh + ((long)slot * a.R + r) * n;
        double dot = 0.0;
        for (int c = 0; c < n; ++c) dot += y[c] * q[c];
        const double b = a.rho[(long)slot * a.R + r] * dot;
        const double coef = al[i] - b;
        for (int c = 0; c < n; ++c) q[c] += coef * s[c];
    }
    double slope = 0.0;
    for (int c = 0; c < n; ++c) slope += -q[c] * pg[c];
    const bool bad = slope >= 0.0; // not a descent direction: steepest descent instead
    const double scale = len == 0 ? 1.0 / fmax(pgmax, 1e-300) : 1.0; // first step: unit-scale move
    for (int c = 0; c < n; ++c) {
        double v = bad ? -pg[c] : -q[c];
        v *= scale;
        // a parameter ON its bound with the gradient pushing into it stays there
        d[c] = (x[c] <= lo[c] && g[c] > 0.0) ? 0.0 : v;
    }
}

__global__ void __launch_bounds__(TPB) lbfgs_trial_kernel(LbfgsArgs a)
{
    const long i = (long)blockIdx.x * TPB + threadIdx.x;
    if (i >= a.R * a.n) return;
    const long r = i / a.n;
    const double xt = fmax(a.x[i] + a.step[r] * a.d[i], a.lo[i]);
    a.xt[i] = xt;
    a.xe[i] = a.searching[r] ? xt : a.x_new[i]; // settled models: at the point they settled on
}

__global__ void __launch_bounds__(TPB) lbfgs_armijo_kernel(LbfgsArgs a)
{
    const long r = (long)blockIdx.x * TPB + threadIdx.x;
    if (r >= a.R) return;
    if (!a.searching[r]) {
        if (a.mask) a.mask[r] = 0;
        return;
    }
    const int n = a.n;
    double gd = 0.0; // directional derivative along the trial displacement (< 0)
    for (int c = 0; c < n; ++c) gd += a.pg[r * n + c] * (a.xt[r * n + c] - a.x[r * n + c]);
    const double ft = a.ft[r], f = a.f[r];
    const bool fin = isfinite(ft);
    const bool ok = fin && (ft <= f + 1e-4 * gd);
    if (a.mask) a.mask[r] = ok ? 1 : 0;
    if (ok) {
        for (int c = 0; c < n; ++c) a.x_new[r * n + c] = a.xt[r * n + c];
        a.f_new[r] = ft;
        if (a.nback) atomicAdd(a.counters + 3, 1); // own line search: the model stays in the flight, its step is taken by the update
        else a.searching[r] = 0;
        return;
    }
    if (a.nback) { // own line search: the budget of trial points is per model; beyond it the model is done (at numerical precision)
        const int nb = a.nback[r] + 1;
        a.nback[r] = nb;
        if (nb >= a.max_backtracks) {
            a.searching[r] = 0;
            return;
        }
    }
    atomicAdd(a.counters + 1, 1);
    // next trial: the minimiser of the parabola through f, its slope and the rejected value, kept inside [0.1, 0.5] of the step
    const double curv = ft - f - gd;
    double theta = (fin && curv > 0.0) ? -gd / (2.0 * curv) : 0.1;
    theta = fmin(fmax(theta, 0.1), 0.5);
    a.step[r] *= theta;
}

__global__ void __launch_bounds__(TPB) lbfgs_update_kernel(LbfgsArgs a)
{
    const long r = (long)blockIdx.x * TPB + threadIdx.x;
    if (r >= a.R) return;
    if (a.mask && !a.mask[r]) return; // own line search: only the models whose trial was accepted take their step now
    const int n = a.n, H = a.H;
    const bool srch = a.searching && !a.mask && a.searching[r] != 0; // lock-step form: no acceptable step -- done, old gradient kept
    const int len = a.hlen[r], pos = a.hpos[r];
    const int slot = (pos + len) % H; // behind the newest pair; with a full ring that is the oldest one, which the new pair replaces
    double sy = 0.0, yy = 0.0;
    for (int c = 0; c < n; ++c) {
        const double gn = (a.keep_old && srch) ? a.g[r * n + c] : a.g_new[r * n + c];
        const double sc = a.x_new[r * n + c] - a.x[r * n + c], yc = gn - a.g[r * n + c];
        sy += sc * yc;
        yy += yc * yc;
    }
    const bool good = sy > 1e-10 * fmax(yy, 1e-300); // (scipy's L-BFGS-B skips the update otherwise)
    if (good) {
        double *s = a.Sh + ((long)slot * a.R + r) * n, *y = a.Yh + ((long)slot * a.R + r) * n;
        for (int c = 0; c < n; ++c) {
            const double gn = (a.keep_old && srch) ? a.g[r * n + c] : a.g_new[r * n + c];
            s[c] = a.x_new[r * n + c] - a.x[r * n + c];
            y[c] = gn - a.g[r * n + c];
        }
        a.rho[(long)slot * a.R + r] = 1.0 / fmax(sy, 1e-300);
        if (len < H) a.hlen[r] = len + 1;
        else a.hpos[r] = (pos + 1) % H;
        atomicAdd(a.counters + 2, 1);
    }
    const double f_prev = a.f[r], f_now = a.f_new[r];
    for (int c = 0; c < n; ++c) {
        const double gn = (a.keep_old && srch) ? a.g[r * n + c] : a.g_new[r * n + c];
        a.x[r * n + c] = a.x_new[r * n + c];
        a.g[r * n + c] = gn;
    }
    a.f[r] = f_now;
    if (a.phase) a.phase[r] = 0; // the next direction call gives this model a new one
    // scipy's relative-reduction test: (f_k - f_{k+1}) / max(|f_k|, |f_{k+1}|, 1) <= ftol stops the model
    const double rel = (f_prev - f_now) / fmax(fmax(fabs(f_prev), fabs(f_now)), 1.0);
    bool act = a.active[r] && !srch && (rel > a.ftol);
    // scipy's maxiter counts ITERATIONS (accepted steps) of one model, not passes of the driver's loop: with every model on its
    // own line search a pass is one TRIAL point, and a model that back-tracks k times per step would get maxiter / (k + 1) steps
    if (a.nit && a.active[r]) {
        const int k = a.nit[r] + 1;
        a.nit[r] = k;
        if (a.maxiter > 0 && k >= a.maxiter) act = false;
    }
    a.active[r] = act ? 1 : 0;
}

hipError_t launch_lbfgs(int which, const LbfgsArgs &a, hipStream_t s)
{
    if (a.n > MK_LBFGS_MAX_N || a.H > MK_LBFGS_MAX_H || a.R <= 0) return hipErrorInvalidValue;
    const unsigned gr = (unsigned)((a.R + TPB - 1) / TPB), ge = (unsigned)((a.R * a.n + TPB - 1) / TPB);
    switch (which) {
    case 0: hipLaunchKernelGGL(lbfgs_direction_kernel, dim3(gr), dim3(TPB), 0, s, a); break;
    case 1: hipLaunchKernelGGL(lbfgs_trial_kernel, dim3(ge), dim3(TPB), 0, s, a); break;
    case 2: hipLaunchKernelGGL(lbfgs_armijo_kernel, dim3(gr), dim3(TPB), 0, s, a); break;
    case 3: hipLaunchKernelGGL(lbfgs_update_kernel, dim3(gr), dim3(TPB), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace mk
