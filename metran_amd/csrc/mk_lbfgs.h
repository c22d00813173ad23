// mk_lbfgs.h -- the lock-step L-BFGS kernels of the batched calibration (mk_lbfgs.hip): argument block shared with the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#define MK_LBFGS_MAX_N 128 /* parameters per model (= states: one alpha per state; the size-generic kernels serve 128) */
#define MK_LBFGS_MAX_H 16  /* history pairs per model (ring slots) */

namespace mk {

struct LbfgsArgs {
    long R;                 // models
    int n, H;               // parameters per model; ring slots (= history pairs kept)
    int keep_old;           // update: models that found no step keep their old gradient (adjoint mode: g_new is only valid elsewhere)
    int max_backtracks;     // armijo (own line search per model): trial points a model may use before it is declared done
    int *hlen, *hpos;       // [R] live pairs of every model's ring and the slot of its oldest one
    unsigned char *phase;   // [R] 0: takes a new direction at the next mk_lbfgs_direction, 1: in the middle of its line search
    int *nback;             // [R] trial points used in the current line search
    unsigned char *mask;    // update: models to update [R] (NULL = all); armijo: receives the accepted models
    double gtol, ftol;
    double *x, *g, *f;      // current point [R,n], gradient [R,n], objective [R]
    const double *lo;       // lower bounds [R,n]
    unsigned char *active, *searching; // [R]
    double *Sh, *Yh, *rho;  // history ring [H,R,n], [H,R,n], [H,R]
    double *pg, *d;         // projected gradient, search direction [R,n]
    double *step;           // [R]
    double *xt, *xe;        // trial point, point to evaluate [R,n]
    double *x_new, *f_new;  // accepted point so far [R,n], its objective [R]
    const double *ft;       // objective at the trial points [R]
    const double *g_new;    // gradient at the accepted points [R,n]
    int *counters;          // device int[4]: #active (direction), #still searching (armijo), #good pairs (update)
    int *nit;               // update: [R] quasi-Newton iterations every model has taken (NULL: not counted)
    int maxiter;            // update: a model whose count reaches it leaves the flight (0: no limit) -- scipy's maxiter, per model
};

hipError_t launch_lbfgs(int which, const LbfgsArgs &a, hipStream_t s); // 0 direction, 1 trial, 2 armijo, 3 update

} // namespace mk
