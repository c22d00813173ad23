"""Batched calibration driver: maximum-likelihood estimation of the
alpha parameters of MANY independent Metran models at once.

The reference calibrates one model per ``Metran.solve()`` call: scipy L-BFGS-B drives
``BaseSolver.objfunction -> Metran.get_mle`` (metran/solver.py:42-63, 222-288), and every gradient is
P+1 sequential filter runs (2-point finite differences, no analytic jac).  Here R models advance in
lock-step: one kernel launch evaluates the objective of every model at its current point AND at its P
forward-difference points ((P+1)*R filter instances sharing the R uploaded records), a bound-projected
L-BFGS update is computed for all models with batched tensor algebra on the device, and the Armijo
back-tracking line search (steps from a quadratic interpolation) evaluates one trial point per still-searching model per launch.

The iteration is not scipy's L-BFGS-B code path (that is ``metran_amd.solver.HipSolve``, which keeps
scipy on the host and reproduces the reference's trajectory for ONE model); it minimises the same
objective -2 log L(alpha) under the same bounds alpha >= pmin (metran/metran.py:439-462) and reaches the
same optimum.  Multi-GPU (one process per GPU): ``calibrate_sharded`` gives every rank a contiguous slice of the
models (``distributed.shard_range``) and gathers the per-model results in rank order -- per-model parameters need no
collective; ``calibrate_shared`` fits ONE parameter vector to all models of all ranks (summed objective and summed
adjoint gradient in one all-reduce of P+1 doubles per evaluation, ``ShardedObjective.value_and_grad``).
"""
import numpy as np

__all__ = ["calibrate_batch", "calibrate_sharded", "calibrate_shared", "CalibrationResult"]


class CalibrationResult(dict):
    """``alpha [R,P]``, ``obj [R]`` (-2 log L), ``grad [R,P]``, ``converged [R]`` (bool; False too where the objective is not finite or
    the model stopped at ``maxiter``), ``nit_model [R]`` (quasi-Newton iterations of every model -- what scipy's ``nit`` counts), ``nit``
    (their maximum), ``passes`` (passes of the driver's loop: with every model on its own line search one pass is one trial point),
    ``nfev`` (filter instances evaluated in total), ``launches``, ``aic [R]`` (= 2P + obj, solver.py:280)."""

    __getattr__ = dict.__getitem__


def calibrate_batch(kf, alpha0=10.0, pmin=1e-5, dt=1.0, warmup=1, maxiter=200, history=10, eps=1e-8,
                    ftol=2.220446049250313e-09, gtol=1e-5, max_backtracks=12, verbose=False, gradient="auto",
                    stderr=False, compact=0.5, compact_min=256, fd_below=0, launch_budget=None, own_search_above=None):
    """Calibrate every record held by ``kf`` (observations + loadings already set).

    ``maxiter`` is scipy's: the number of quasi-Newton ITERATIONS (accepted steps) one model may take, counted per model on the
    device (``mk_lbfgs_update``), whichever schedule runs -- a model that reaches it leaves the flight unconverged.  The
    driver's loop additionally stops after ``maxiter * (max_backtracks + 1)`` passes (every iteration may use its whole
    back-tracking budget).  ``history`` <= 16 and N + K <= 128 (the L-BFGS kernels' ring and parameter limits).

    Parameters mirror scipy's L-BFGS-B defaults used by the reference (``eps`` forward-difference step,
    ``ftol = factr*epsmch`` with factr 1e7, ``gtol = pgtol`` on the projected gradient, ``history = m``);
    ``alpha0`` and ``pmin`` are Metran's initial value and lower bound (metran/metran.py:446-462).
    ``gradient``: "fd" = the reference's forward differences ((n+1)*R filter instances per gradient),
    "adjoint" = ``BatchedKalman.loglik_grad_alpha`` (one forward + one backward launch over R instances,
    exact to rounding), "auto" = adjoint where available (every supported shape).
    ``compact``: models converge after very different numbers of iterations (8192 synthetic models: half within 25, the
    last one after 140), and a lock-step launch costs the same whether a model still moves or not; whenever the active
    models are fewer than ``compact`` x the models in flight (and more than ``compact_min`` are in flight), the active
    records are gathered into a smaller engine (``BatchedKalman.subset``) and the iteration continues on those.
    Every model sees exactly the iterates it would have seen without it: the sub-engine inherits the parent's kernel
    choices (``BatchedKalman.subset`` pins a batch-size-dependent "auto" to what it resolved to for the parent).  0 disables.
    ``fd_below``: with the adjoint gradient, switch to forward differences once (n+1) x the models in flight fit in
    ``fd_below`` instances (0 = never).  A handful of stragglers is latency-bound: one objective launch over (n+1) R
    instances takes as long as one over R, the forward + backward pair of the adjoint 4-5 times as long; the price is
    the differencing error in the last iterations (what scipy's L-BFGS-B works with throughout).
    ``own_search_above``: with the adjoint gradient and MORE than this many models in flight every model runs its own line search
    across iterations (one forward + at most one backward launch per iteration; a model either takes a new direction or its next
    shorter step) instead of the lock-step search, whose every back-tracking round is a launch over the whole flight that only a
    few models need.  Each model's sequence of trial points is the same either way.  Below the threshold a launch is
    latency-bound whatever its size, back-tracking rounds are cheap and backward passes are not (2.5 x a forward pass): lock-step.
    Default (measured, profiles/r05/ab_line_search.log): 2048 for models of at most 16 states, 0 -- always -- for wider ones,
    whose stragglers rarely back-track, so that their own search costs them no iterations while the lock-step one pays a
    back-tracking round of somebody's in most iterations.
    ``compact_min`` holds while the adjoint gradient is in use (a launch of fewer models than wavefront slots is latency-bound: a
    smaller flight is no faster).  It does not stop the compaction that brings the flight under ``fd_below``, nor those after it
    that let at least four step lengths of every model ride in one launch.
    ``launch_budget``: instances one latency-bound objective launch may carry (default: two wavefronts per SIMD of an MI355X --
    8192 instances of the 16-lane kernels, four models per wavefront; 2048 of the wide ones, one model per wavefront: measured, 4096 wide instances take 14 ms where 512 take 5.5).  With differenced gradients, a flight whose (n+1) R instances fit gets the
    gradient of every trial with the trial, and as many step lengths per searching model at once as fit (up to the whole
    back-tracking budget): a straggler that exhausts its budget then costs ONE launch instead of a dozen.
    """
    import torch

    R, n = kf.R, kf.n
    dev = kf.device
    if not 1 <= int(history) <= 16:
        raise ValueError("history must be between 1 and 16 (MK_LBFGS_MAX_H: the L-BFGS kernels keep at most 16 pairs per model); got %r" % (history,))
    if n > 128:
        raise ValueError("calibrate_batch serves N + K <= 128 parameters per model (MK_LBFGS_MAX_N); this engine has %d" % n)
    if int(maxiter) < 1:
        raise ValueError("maxiter must be at least 1")
    if launch_budget is None:
        launch_budget = 8192 if n <= 16 else 2048
    if own_search_above is None:
        own_search_above = 2048 if n <= 16 else 0
    f64 = dict(dtype=torch.float64, device=dev)
    x = torch.full((R, n), float(alpha0), **f64) if np.isscalar(alpha0) else kf._dev(alpha0, (R, n), "alpha0").clone()
    lo = torch.full((R, n), float(pmin), **f64) if np.isscalar(pmin) else kf._dev(pmin, (R, n), "pmin")
    x = torch.maximum(x, lo)
    nfev = launches = 0
    eye = torch.eye(n, **f64) * eps
    kf0, R0 = kf, R                                   # the caller's engine and model count
    orig = torch.arange(R, device=dev)                # original number of each model in flight
    X_all = torch.empty((R, n), **f64)                # results of the models that have left the flight
    F_all = torch.empty(R, **f64)
    G_all = torch.empty((R, n), **f64)
    active_all = torch.zeros(R, dtype=torch.bool, device=dev)

    if gradient == "auto":
        gradient = "adjoint" if kf.has_adjoint() else "fd"
        if gradient == "adjoint":
            # the recording forward pass needs R*T*record_stride(n) doubles (10.7 KB per model-step at n = 36): a large wide
            # flight does not fit where the differenced objective, which has no workspace, does (round-3 advice)
            need = 8.0 * R * kf.T * kf.record_stride()
            have = getattr(kf, "_grad_work", None)
            have = 8.0 * have.numel() if have is not None else 0.0
            free = float(torch.cuda.mem_get_info(dev)[0]) if dev.type == "cuda" else float("inf")
            if need > have + 0.9 * free:
                gradient = "fd"
    if gradient not in ("fd", "adjoint"):
        raise ValueError("gradient must be 'auto', 'fd' or 'adjoint'")
    grad_mode = gradient                              # may switch to "fd" for a small flight (fd_below); the request stays

    def value_and_grad(xc):
        """f and gradient of all R models: ONE launch of (n+1)*R instances (forward differences) or a
        forward + a backward launch of R instances (adjoint)."""
        nonlocal nfev, launches
        if grad_mode == "adjoint":
            f, g = kf.loglik_grad_alpha(xc, dt=dt, warmup=warmup)
            nfev += R
            launches += 2
            return f, g
        pts = torch.cat([xc[None], xc[None] + eye[:, None, :]], 0).reshape((n + 1) * R, n)  # instance s*R + r
        phi, q = kf.params_from_alpha(pts, dt=dt)
        f = kf.loglik(phi, q, warmup=warmup).reshape(n + 1, R)
        nfev += (n + 1) * R
        launches += 1
        return f[0].clone(), ((f[1:] - f[0:1]) / eps).transpose(0, 1).contiguous()

    def value(xc):
        """Objective at a trial point.  With the adjoint gradient it is the RECORDING forward pass: the gradient of the
        accepted point is then the backward walk over the records of the last trial, not a second filter run."""
        nonlocal nfev, launches
        nfev += R
        launches += 1
        if grad_mode == "adjoint":
            return kf.loglik_forward_alpha(xc, dt=dt, warmup=warmup)
        phi, q = kf.params_from_alpha(xc, dt=dt)
        return kf.loglik(phi, q, warmup=warmup).clone()

    def proj_grad(xc, g):  # gradient with the components pushing into an active bound removed
        return torch.where((xc <= lo) & (g > 0), torch.zeros_like(g), g)

    f, g = value_and_grad(x)
    f, g = f.contiguous(), g.contiguous()
    # L-BFGS history: every model has its own ring of `history` pairs [slot, R, n] (hlen live pairs from slot hpos): with the
    # adjoint gradient every model runs its OWN line search (below), so models take their steps in different iterations
    Sh = torch.zeros((history, R, n), **f64)
    Yh = torch.zeros((history, R, n), **f64)
    rho = torch.zeros((history, R), **f64)
    hlen = torch.zeros(R, dtype=torch.int32, device=dev)
    hpos = torch.zeros(R, dtype=torch.int32, device=dev)
    active = torch.ones(R, dtype=torch.bool, device=dev)
    phase = torch.zeros(R, dtype=torch.uint8, device=dev)   # 1: in the middle of its line search
    nback = torch.zeros(R, dtype=torch.int32, device=dev)
    accepted = torch.zeros(R, dtype=torch.bool, device=dev)
    step = torch.ones(R, **f64)
    pg, d = torch.zeros_like(x), torch.zeros_like(x)
    xt, xe = torch.empty_like(x), torch.empty_like(x)
    x_new, f_new = x.clone(), f.clone()
    nit_model = torch.zeros(R, dtype=torch.int32, device=dev)   # quasi-Newton iterations per model (scipy's nit)
    NIT_all = torch.zeros(R, dtype=torch.int32, device=dev)
    passes = 0
    for passes in range(1, int(maxiter) * (int(max_backtracks) + 1) + 1):
        nit = passes
        # projected gradient, convergence test on it, two-loop recursion with its safeguards: ONE launch (mk_lbfgs.hip; one
        # thread per model) and one host synchronisation for the count.  A model in the middle of its line search (adjoint mode)
        # keeps its direction and its shortened step.
        own = grad_mode == "adjoint" and R > own_search_above
        n_act = kf.lbfgs_direction(x, g, lo, active, Sh, Yh, rho, hlen, hpos, gtol, pg, d, phase if own else None,
                                   step if own else None, nback if own else None)
        if n_act == 0:
            break
        to_fd = grad_mode == "adjoint" and fd_below and (n + 1) * n_act <= fd_below < (n + 1) * R
        more_trials = grad_mode == "fd" and launch_budget // ((n + 1) * R) < 4
        if compact and n_act < compact * R and (R > compact_min or to_fd or more_trials):
            X_all[orig], F_all[orig], G_all[orig], NIT_all[orig] = x, f, g, nit_model   # everybody's current state; the inactive ones are final
            keep = active.nonzero().squeeze(1)
            kf = kf.subset(keep)
            x, f, g, lo, pg, d, orig, hlen, hpos, phase, nback, step, x_new, f_new, nit_model = (
                t[keep].contiguous() for t in (x, f, g, lo, pg, d, orig, hlen, hpos, phase, nback, step, x_new, f_new, nit_model))
            Sh, Yh, rho = Sh[:, keep].contiguous(), Yh[:, keep].contiguous(), rho[:, keep].contiguous()
            R = n_act
            active = torch.ones(R, dtype=torch.bool, device=dev)
            accepted = torch.zeros(R, dtype=torch.bool, device=dev)
            xt, xe = torch.empty_like(x), torch.empty_like(x)
        if grad_mode == "adjoint" and fd_below and (n + 1) * R <= fd_below:
            grad_mode = "fd"
        if own and not (grad_mode == "adjoint" and R > own_search_above):
            # the flight has shrunk below the threshold (or switched to differences): lock-step from here on.  A model in the
            # middle of its line search restarts it at the unit step -- same direction, same trial points.
            own = False
            phase.zero_()
        if own:
            # ---- adjoint gradient: every model runs its OWN line search across iterations.  An iteration is ONE recording forward
            # launch -- each active model at its own trial point, a fresh unit step or the next shorter one -- and, if any trial was
            # accepted, ONE backward launch for the gradients there.  (Round 4 ran the line search in lock-step: with thousands of
            # models in flight some model backtracks in most iterations, and every back-tracking round was a launch over the
            # whole flight that only those few needed.)  A model's own sequence of trial points is what it was.
            kf.lbfgs_trial(x, d, step, lo, active, x, xt, xe)
            ft = value(xe)
            n_search, n_acc = kf.lbfgs_armijo(ft.contiguous(), f, pg, xt, x, active, step, x_new, f_new, nback, max_backtracks, accepted)
            if n_acc > 0:
                launches += 1
                g_cand = kf.loglik_backward_alpha()
                kf.lbfgs_update(x, f, g, x_new, f_new, g_cand.contiguous(), False, None, active, ftol, Sh, Yh, rho, hlen, hpos, accepted, phase,
                                nit=nit_model, maxiter=maxiter)
            if verbose:
                print("it %3d  active %5d  mean obj %.6f" % (nit, int(active.sum()), float(f.mean())))
            continue
        # ---- lock-step Armijo back-tracking, one trial per searching model per launch (differenced gradients; the adjoint
        # gradient on a flight of at most `own_search_above` models)
        step = torch.ones(R, **f64)
        searching = active.clone()
        x_new, f_new = x.clone(), f.clone()
        # differenced gradients on a flight that fits one round of wavefronts: the n difference points of every TRIAL ride
        # in its launch (a launch is latency-bound there: T sequential steps whatever the number of instances), so an
        # accepted trial needs no second launch for its gradient
        speculate = grad_mode == "fd" and (n + 1) * R <= launch_budget
        g_acc = g.clone() if speculate else None
        # ... and SEVERAL step lengths of every searching model ride in one launch while they fit (S_tr x (n+1) x R instances
        # <= launch_budget): with models in lock-step some model needs a short step in most iterations, and on a small flight every
        # back-tracking round is a latency-bound launch of its own (round 3: 4.6 launches per iteration over the tail).  The
        # largest of the simultaneous trials that passes the Armijo test is taken.
        S_tr = max(1, min(max_backtracks, launch_budget // ((n + 1) * R))) if speculate else 1
        ratios = torch.tensor([0.35 ** k for k in range(S_tr)], **f64)   # 1, 0.35, 0.12, 0.04, ...: twelve of them reach 1e-5
        # the same budget of trial POINTS per model either way (max_backtracks): a straggler whose differenced gradient no longer
        # yields an acceptable step uses all of it before it is declared done, and on a small flight every round is a launch
        trials_left, first_round, sub, sub_idx = int(max_backtracks), True, None, None
        while trials_left > 0:
            if speculate and not first_round:
                # After the first round only a few models still search (a straggler whose differenced gradient no longer yields an
                # acceptable step uses its whole budget of trial points before it is declared done), and every further round used
                # to be a launch over the WHOLE flight -- 11 more latency-bound launches for one or two models (round 6 trace of
                # 512 wide models: five such iterations cost 85 ms each, a fifth of the calibration).  Those models alone, as a
                # sub-engine, get every remaining step length that fits one launch.
                idx = searching.nonzero().squeeze(1)
                m = int(idx.numel())
                S2 = min(trials_left, launch_budget // ((n + 1) * m)) if m else 0
                if S2 > S_tr and 2 * m <= R:
                    if sub is None or sub_idx.numel() != m or not bool(torch.equal(sub_idx, idx)):
                        if sub is not None:
                            sub.close()
                        sub, sub_idx = kf.subset(idx), idx
                    ratios2 = torch.tensor([0.35 ** k for k in range(S2)], **f64)
                    xs, ds, los, pgs, fs, st = x[idx], d[idx], lo[idx], pg[idx], f[idx], step[idx]
                    steps_s = st[None, :] * ratios2[:, None]                                    # [S2,m]
                    xt_s = torch.maximum(xs[None] + steps_s[:, :, None] * ds[None], los[None])  # [S2,m,n]
                    pts = torch.cat([xt_s[:, None], xt_s[:, None] + eye[None, :, None, :]], 1)  # [S2, n+1, m, n]
                    phi_t, q_t = sub.params_from_alpha(pts.reshape(S2 * (n + 1) * m, n), dt=dt)
                    fv = sub.loglik(phi_t, q_t, warmup=warmup).reshape(S2, n + 1, m)
                    nfev += S2 * (n + 1) * m
                    launches += 1
                    ft_s = fv[:, 0]
                    gd_s = (pgs[None] * (xt_s - xs[None])).sum(2)
                    ok_s = (ft_s <= fs[None] + 1e-4 * gd_s) & torch.isfinite(ft_s)
                    ok = ok_s.any(0)
                    first = torch.argmax(ok_s.to(torch.int8), 0)                                # largest passing step of each model
                    xt1 = torch.gather(xt_s, 0, first[None, :, None].expand(1, m, n))[0]
                    ft = torch.gather(ft_s, 0, first[None])[0]
                    gt = ((torch.gather(fv[:, 1:], 0, first[None, None, :].expand(1, n, m))[0] - ft[None]) / eps).transpose(0, 1)
                    acc = idx[ok]
                    x_new[acc], f_new[acc], g_acc[acc] = xt1[ok], ft[ok], gt[ok]
                    searching[acc] = False
                    trials_left -= S2
                    if not bool(searching.any()):
                        break
                    ft_l, gd_l = ft_s[-1], gd_s[-1]                                             # go on below the shortest one
                    curv = ft_l - fs - gd_l
                    theta = torch.where(torch.isfinite(ft_l) & (curv > 0), -gd_l / (2.0 * curv), torch.full_like(ft_l, 0.1))
                    step[idx] = torch.where(ok, st, st * float(ratios2[-1]) * theta.clamp(0.1, 0.5))
                    continue
            first_round = False
            trials_left -= S_tr
            if S_tr > 1:
                steps_s = step[None, :] * ratios[:, None]                                   # [S,R]
                xt_s = torch.maximum(x[None] + steps_s[:, :, None] * d[None], lo[None])     # [S,R,n]
                xe_s = torch.where(searching[None, :, None], xt_s, x_new[None])
                pts = torch.cat([xe_s[:, None], xe_s[:, None] + eye[None, :, None, :]], 1)  # [S, n+1, R, n]: instance ((s, j), r)
                phi_t, q_t = kf.params_from_alpha(pts.reshape(S_tr * (n + 1) * R, n), dt=dt)
                fv = kf.loglik(phi_t, q_t, warmup=warmup).reshape(S_tr, n + 1, R)
                nfev += S_tr * (n + 1) * R
                launches += 1
                ft_s = fv[:, 0]
                gd_s = (pg[None] * (xt_s - x[None])).sum(2)
                ok_s = searching[None] & (ft_s <= f[None] + 1e-4 * gd_s) & torch.isfinite(ft_s)
                ok = ok_s.any(0)
                first = torch.argmax(ok_s.to(torch.int8), 0)                                # largest passing step of each model
                sel = first[None, :, None].expand(1, R, n)
                xt1 = torch.gather(xt_s, 0, sel)[0]
                ft = torch.gather(ft_s, 0, first[None])[0]
                gt = ((torch.gather(fv[:, 1:], 0, first[None, None, :].expand(1, n, R))[0] - ft[None]) / eps).transpose(0, 1)
                x_new = torch.where(ok[:, None], xt1, x_new)
                f_new = torch.where(ok, ft, f_new)
                g_acc = torch.where(ok[:, None], gt, g_acc)
                searching &= ~ok
                if not bool(searching.any()):
                    break
                # nobody among a model's trials passed: go on below its shortest one
                ft_l, gd_l = ft_s[-1], gd_s[-1]
                curv = ft_l - f - gd_l
                theta = torch.where(torch.isfinite(ft_l) & (curv > 0), -gd_l / (2.0 * curv), torch.full_like(ft_l, 0.1))
                step = torch.where(searching, step * float(ratios[-1]) * theta.clamp(0.1, 0.5), step)
                continue
            # one trial per searching model: trial point, launch, Armijo test + next step length -- two small launches around
            # the filter's (mk_lbfgs_trial / mk_lbfgs_armijo), one synchronisation for the number of models still searching
            kf.lbfgs_trial(x, d, step, lo, searching, x_new, xt, xe)   # settled models: at the point they settled on
            if speculate:
                before = searching.clone()
                ft, gt = value_and_grad(xe)
            else:
                ft = value(xe)
            # next trial of a rejected model: the minimiser of the parabola through f, its slope and the rejected value, kept
            # inside [0.1, 0.5] of the rejected step (plain halving needs log2 of the ratio in launches, and every launch costs
            # the whole flight)
            n_search = kf.lbfgs_armijo(ft.contiguous(), f, pg, xt, x, searching, step, x_new, f_new)
            if speculate:
                g_acc = torch.where((before & ~searching)[:, None], gt, g_acc)
            if n_search == 0:
                break
        if sub is not None:
            sub.close()
        # a model still searching found no acceptable step: it is done (at numerical precision) and keeps its old gradient
        if grad_mode == "adjoint":
            # the last trial launch evaluated every model AT x_new (a model that found no step: at a rejected point)
            launches += 1
            g_cand, f_tmp, keep_old = kf.loglik_backward_alpha(), f_new, True
        elif speculate:
            g_cand, f_tmp, keep_old = g_acc, f_new, False
        else:
            f_tmp, g_cand = value_and_grad(x_new)
            keep_old = False
        # the pair (s, y) of this step into every model's ring (skipped where it is not usable, as scipy does), (x, f, g) <- the
        # accepted point, scipy's relative-reduction test (f_k - f_{k+1}) / max(|f_k|, |f_{k+1}|, 1) <= ftol: ONE launch
        kf.lbfgs_update(x, f, g, x_new.contiguous(), f_tmp.contiguous(), g_cand.contiguous(), keep_old, searching, active, ftol,
                        Sh, Yh, rho, hlen, hpos, nit=nit_model, maxiter=maxiter)
        if verbose:
            print("it %3d  active %5d  mean obj %.6f" % (nit, int(active.sum()), float(f.mean())))
    X_all[orig], F_all[orig], G_all[orig], active_all[orig], NIT_all[orig] = x, f, g, active, nit_model
    kf, R, x, f, g, active = kf0, R0, X_all, F_all, G_all, active_all
    # a model that stopped at maxiter left the flight (active = False) without converging: scipy reports success False there
    hit_cap = NIT_all >= int(maxiter)
    lo = torch.full((R, n), float(pmin), **f64) if np.isscalar(pmin) else kf._dev(pmin, (R, n), "pmin")
    pg = proj_grad(x, g)
    # a model whose objective is not a number (communalities above 1 give negative transition variances, say) leaves the
    # flight at once -- NaN compares false with gtol -- but it has not converged (scipy: ABNORMAL_TERMINATION, success False)
    finite = torch.isfinite(f) & torch.isfinite(g).all(1)
    res = CalibrationResult(alpha=x, obj=f, grad=g, converged=~active & finite & ~hit_cap, nit=int(NIT_all.max().item()) if R else 0,
                            nit_model=NIT_all, passes=passes, nfev=nfev, launches=launches, aic=2 * n + f, pgnorm=pg.abs().amax(1))
    if stderr:
        # the Hessian is differenced from the exact (adjoint) gradient whatever gradient the ITERATION used
        # (gradient="fd", or the fd_below switch for a small flight): eligibility is the adjoint kernel's, not the mode's
        if not kf.has_adjoint():
            raise ValueError("stderr=True differences the adjoint gradient, which this shape (N=%d, K=%d) does not have"
                             % (kf.N, kf.K))
        d = 1e-5 * x.abs().clamp_min(0.1)                                   # [R,n] step per parameter
        # the n + 1 gradient evaluations ride as batch -- as many point sets per launch pair as the recording forward pass
        # has workspace for (R*T*record_stride(n) doubles per set: a wide flight that needed the memory check above for ONE
        # set cannot hold n + 1 of them; round-4 advice: it used to run out of memory after the whole calibration)
        sets = _grad_sets_that_fit(kf, R, n + 1)
        gg = torch.empty((n + 1, R, n), **f64)
        stderr_launches = 0
        for s0 in range(0, n + 1, sets):
            s1 = min(n + 1, s0 + sets)
            pts = x[None].repeat(s1 - s0, 1, 1)                             # [sets,R,n]: set 0 is x itself, set 1 + j steps x_j
            for s_ in range(max(s0, 1), s1):
                pts[s_ - s0, :, s_ - 1] += d[:, s_ - 1]
            _, gch = kf.loglik_grad_alpha(pts.reshape((s1 - s0) * R, n), dt=dt, warmup=warmup)
            gg[s0:s1] = gch.reshape(s1 - s0, R, n)
            stderr_launches += 2
        hess = ((gg[1:] - gg[0:1]) / d.transpose(0, 1)[:, :, None]).permute(1, 0, 2)   # [R, j, :] = d grad / d x_j
        hess = 0.5 * (hess + hess.transpose(1, 2))
        pcov = torch.linalg.pinv(hess)
        res["pcov"] = pcov
        res["stderr"] = torch.sqrt(torch.diagonal(pcov, dim1=1, dim2=2))
        res["nfev"] = nfev + (n + 1) * R
        res["launches"] = launches + stderr_launches
    return res


def _grad_sets_that_fit(kf, R, want):
    """How many parameter sets of R instances one ``loglik_grad_alpha`` launch pair can take (<= ``want``): its recording
    forward pass needs R*T*record_stride(n) doubles per set.  Raises when not even one fits."""
    import torch

    per_set = 8.0 * R * kf.T * kf.record_stride()
    have = getattr(kf, "_grad_work", None)
    have = 8.0 * have.numel() if have is not None else 0.0
    dev = kf.device
    free = float(torch.cuda.mem_get_info(dev)[0]) if getattr(dev, "type", "cpu") == "cuda" else float("inf")
    room = max(have, 0.8 * free)  # a larger workspace replaces the current one (engine._grad_work is reallocated)
    sets = int(min(want, room // per_set)) if per_set > 0 else want
    if sets < 1:
        raise MemoryError("stderr=True needs %.1f GB for the recording forward pass of the adjoint gradient (%d models, T=%d, "
                          "n=%d); %.1f GB are free -- calibrate in smaller flights" % (per_set / 1e9, R, kf.T, kf.n, free / 1e9))
    return sets



def calibrate_sharded(n_models, build_engine, **kwargs):
    """``calibrate_batch`` over the ranks of torch.distributed: ``build_engine(lo, hi)`` returns a ``BatchedKalman``
    holding records ``lo .. hi-1`` (observations + loadings set); every rank calibrates its slice and the per-model
    results (``alpha [R,P]``, ``obj``, ``converged``, ``pgnorm`` and, with ``stderr=True``, ``stderr``) come back
    concatenated in rank order on every rank.  ``nfev``/``launches``/``nit`` are this rank's own."""
    from .distributed import run_sharded

    info = {}

    def local(lo, hi):
        res = calibrate_batch(build_engine(lo, hi), **kwargs)
        info.update(nit=res.nit, nfev=res.nfev, launches=res.launches, shard=(lo, hi))
        out = {k: res[k] for k in ("alpha", "obj", "pgnorm") if k in res}
        out["converged"] = res.converged.to(res.obj.dtype)
        if "stderr" in res:
            out["stderr"] = res["stderr"]
        return out

    g = run_sharded(n_models, local)
    g["converged"] = g["converged"] > 0.5
    return CalibrationResult(aic=2 * g["alpha"].shape[1] + g["obj"], **g, **info)


def calibrate_shared(local_value_and_grad, x0, pmin=1e-5, **kwargs):
    """ONE parameter vector for all models of all ranks: scipy L-BFGS-B on the host (the reference's optimiser,
    metran/solver.py:248-255) minimising the SUMMED objective.  ``local_value_and_grad(alpha [P]) -> (values [b],
    grads [b,P])`` evaluates this rank's models (``BatchedKalman.loglik_grad_alpha`` on the alpha vector repeated for
    every local record: forward filter + adjoint kernel); the ranks are combined by ``ShardedObjective.value_and_grad``
    -- one all-reduce of P+1 float64 per evaluation, after which every rank holds bit-identical values and follows the
    same iteration.  Returns scipy's result object."""
    import torch
    from scipy.optimize import minimize

    from .distributed import ShardedObjective

    obj = ShardedObjective(None)

    def fun(x):
        total, grad = obj.value_and_grad(torch.as_tensor(x, dtype=torch.float64), local_value_and_grad)
        return float(total), grad.detach().cpu().numpy().astype(np.float64)

    x0 = np.asarray(x0, dtype=np.float64)
    return minimize(fun, x0, jac=True, method="l-bfgs-b", bounds=[(pmin, None)] * x0.size, **kwargs)
