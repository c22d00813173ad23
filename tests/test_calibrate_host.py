"""CPU: the HOST logic of the batched calibration driver (metran_amd/calibrate.py -- SURVEY.md section 8f, row f1) with the
device engine replaced by an oracle-backed stand-in that offers the same methods (tests may use the oracle; the product
never does): the bound-projected L-BFGS update, the Armijo search with one trial per launch, the several-trials-per-launch
search of a small differenced flight, the compaction of the flight onto the still-active records, the switch from the
adjoint to differenced gradients, the adjoint's forward / backward bookkeeping.  The optimum is checked against scipy's
L-BFGS-B (the reference's optimiser, metran/solver.py:248-255) on the same objective.  The GPU tier runs the same driver
on the real engine (tests/test_hip_solver.py, tests/test_bench_gpu.py)."""
import numpy as np
import pytest
import torch
from scipy.optimize import minimize

import oracle
from metran_amd.calibrate import calibrate_batch
from oracle_engine import OracleEngine
from metran_amd.params import phi_q_from_alpha
from metran_amd.synthetic import make_dfm_batch


def scipy_optimum(eng, r, alpha0=10.0, pmin=1e-5):
    def fun(a):
        f, g = eng._grad(torch.from_numpy(np.tile(a, (eng.R, 1))), 1.0, 1)
        return float(f[r]), g[r].numpy()

    # (the whole batch is evaluated for one model's value: fine at these sizes)
    lo = np.broadcast_to(np.asarray(pmin, float), (eng.R, eng.n))[r] if not np.isscalar(pmin) else np.full(eng.n, pmin)
    return minimize(fun, np.maximum(np.full(eng.n, alpha0), lo), jac=True, method="l-bfgs-b",
                    bounds=[(l, None) for l in lo], options=dict(ftol=2.220446049250313e-09, gtol=1e-5))


@pytest.fixture(scope="module")
def models():
    d = make_dfm_batch(5, 3, 1, 160, seed=21, missing=0.2)
    return d["obs"], d["loadings"]


@pytest.fixture(scope="module")
def optima(models):
    eng = OracleEngine(*models)
    return [scipy_optimum(eng, r) for r in range(eng.R)]


def check_against_scipy(res, optima):
    assert bool(res.converged.all())
    for r, ref in enumerate(optima):
        assert float(res.obj[r]) <= ref.fun + 1e-6 * max(1.0, abs(ref.fun)), (r, float(res.obj[r]), ref.fun)
        assert abs(float(res.obj[r]) - ref.fun) <= 2e-5 * max(1.0, abs(ref.fun))
    assert np.all(np.asarray(res.aic) == 2 * res.alpha.shape[1] + np.asarray(res.obj))


def test_adjoint_iteration_reaches_scipys_optimum(models, optima):
    eng = OracleEngine(*models)
    res = calibrate_batch(eng, gradient="adjoint", compact=0)
    check_against_scipy(res, optima)
    # bookkeeping (lock-step line search: the flight is small): every trial point is ONE recording forward launch over the R
    # models in flight, every iteration ends in ONE backward launch; the first gradient is the only forward+backward pair
    kinds = [k for k, _ in eng.log]
    assert kinds[0] == "forward+backward" and kinds.count("forward+backward") == 1
    assert res.nit - 1 <= kinds.count("backward") <= res.nit
    assert all(b == eng.R for _, b in eng.log)
    assert res.launches == 2 + kinds.count("forward") + kinds.count("backward")
    assert res.nfev == eng.R * (1 + kinds.count("forward"))
    # every backward pass directly follows the forward pass whose records it walks
    for i, k in enumerate(kinds):
        if k == "backward":
            assert kinds[i - 1] == "forward"


def test_own_line_search_per_model(models, optima):
    """A flight above ``own_search_above``: every model runs its OWN line search across iterations -- an iteration is ONE
    recording forward launch over the R models in flight, each at its own trial point, followed by ONE backward launch if any
    trial was accepted.  Same optimum; and since a model's sequence of trial points does not depend on its neighbours, the
    result of a model is the same whether it is calibrated alone or in the flight."""
    eng = OracleEngine(*models)
    res = calibrate_batch(eng, gradient="adjoint", compact=0, own_search_above=0)
    check_against_scipy(res, optima)
    kinds = [k for k, _ in eng.log]
    assert kinds[0] == "forward+backward" and kinds.count("forward+backward") == 1
    # one pass of the driver's loop = one trial point per model; a model's ITERATIONS (accepted steps, what scipy's maxiter and
    # nit count: round-5 advice) are counted per model on the device and are at most the passes
    assert res.passes - 1 <= kinds.count("forward") <= res.passes and 0 < kinds.count("backward") <= kinds.count("forward")
    assert res.nit == int(res.nit_model.max()) and 0 < res.nit <= res.passes and bool((res.nit_model > 0).all())
    assert all(b == eng.R for _, b in eng.log)
    assert res.launches == 2 + kinds.count("forward") + kinds.count("backward")
    assert res.nfev == eng.R * (1 + kinds.count("forward"))
    for i, k in enumerate(kinds):
        if k == "backward":
            assert kinds[i - 1] == "forward"
    obs, load = models
    alone = calibrate_batch(OracleEngine(obs[3:4], load[3:4]), gradient="adjoint", compact=0, own_search_above=0)
    assert torch.equal(alone.alpha[0], res.alpha[3]) and torch.equal(alone.obj[0], res.obj[3])
    # ... and the lock-step search walks the same trial points: same iterates
    lock = calibrate_batch(OracleEngine(*models), gradient="adjoint", compact=0)
    np.testing.assert_allclose(lock.alpha.numpy(), res.alpha.numpy(), rtol=1e-9)
    # the flight shrinks below the threshold in the middle of the run: lock-step from there on, same optimum
    mixed = calibrate_batch(OracleEngine(*models), gradient="adjoint", compact=0.9, compact_min=2, own_search_above=3)
    check_against_scipy(mixed, optima)


def test_compaction_leaves_every_models_iterates_unchanged(models):
    """``compact``: the flight is gathered onto the still-active records (``subset``); the docstring's promise is that every
    model sees exactly the iterates it would have seen in the full flight."""
    full = OracleEngine(*models)
    a = calibrate_batch(full, gradient="adjoint", compact=0)
    comp = OracleEngine(*models)
    b = calibrate_batch(comp, gradient="adjoint", compact=0.9, compact_min=2)
    assert min(n for _, n in comp.log) < comp.R         # it did compact
    assert torch.equal(a.alpha, b.alpha) and torch.equal(a.obj, b.obj) and torch.equal(a.grad, b.grad)
    assert torch.equal(a.converged, b.converged)
    assert b.nfev < a.nfev


def test_differenced_small_flight_takes_several_trials_per_launch(models, optima):
    """gradient="fd" on a flight whose (n+1) R instances fit the launch budget: every launch carries S_tr step lengths x (n+1)
    difference points x R models -- as many as fit, up to the whole back-tracking budget of 12 (round 5; four before) -- and an
    accepted trial needs no further launch for its gradient."""
    eng = OracleEngine(*models, adjoint=False)
    n, R = eng.n, eng.R
    res = calibrate_batch(eng, gradient="auto", compact=0, launch_budget=4 * (n + 1) * R)
    check_against_scipy(res, optima)
    assert eng.log[0] == ("loglik", (n + 1) * R)
    assert all(k == "loglik" and b == 4 * (n + 1) * R for k, b in eng.log[1:])
    eng12 = OracleEngine(*models, adjoint=False)
    res12 = calibrate_batch(eng12, gradient="auto", compact=0)          # default budget: all twelve step lengths at once
    check_against_scipy(res12, optima)
    assert all(b == 12 * (n + 1) * R for _, b in eng12.log[1:]) and len(eng12.log) <= len(eng.log)
    assert res.launches == len(eng.log) and res.nfev == sum(b for _, b in eng.log)
    # the differenced gradient the result carries is the forward difference at the result
    f, g = res.obj.numpy(), res.grad.numpy()
    for j in range(n):
        x = res.alpha.numpy().copy()
        x[:, j] += 1e-8
        phi, q = phi_q_from_alpha(x, models[1])
        fj = oracle.dfm_batch(models[0], phi, q, models[1], smooth=False, outputs="mle")["mle"]
        np.testing.assert_allclose((fj - f) / 1e-8, g[:, j], atol=1e-12)


def test_switch_to_differences_below_a_flight_size(models, optima):
    eng = OracleEngine(*models)
    res = calibrate_batch(eng, gradient="adjoint", compact=0, fd_below=10 ** 6)
    check_against_scipy(res, optima)
    kinds = [k for k, _ in eng.log]
    assert kinds[0] == "forward+backward" and set(kinds[1:]) == {"loglik"}   # the first gradient is exact, the rest differenced


def test_active_bounds(models):
    """Lower bounds above the unconstrained optimum of some parameters: the projected iteration stops ON the bound with the
    gradient pointing into it, at scipy's bounded optimum."""
    eng = OracleEngine(*models)
    pmin = np.full((eng.R, eng.n), 1e-5)
    pmin[:, 0] = 25.0
    pmin[1, 2] = 40.0
    res = calibrate_batch(eng, pmin=pmin, alpha0=30.0, gradient="adjoint", compact=0)
    assert bool(res.converged.all())
    x, g = res.alpha.numpy(), res.grad.numpy()
    assert np.all(x >= pmin)
    for r in range(eng.R):
        ref = scipy_optimum(eng, r, alpha0=30.0, pmin=pmin)
        assert abs(float(res.obj[r]) - ref.fun) <= 2e-5 * max(1.0, abs(ref.fun)), (r, float(res.obj[r]), ref.fun)
        on = ref.x <= pmin[r] + 1e-12
        assert np.array_equal(x[r] <= pmin[r], on), (r, x[r], ref.x)
        assert np.all(g[r][on] > 0)
    assert float(res.pgnorm.max()) < 1e-3
    assert np.any(x <= pmin)


def test_a_model_without_a_finite_objective_is_not_converged(models):
    """Communalities above 1 (negative transition variances): the objective is NaN from the first evaluation.  The model drops
    out of the flight at once, keeps its start point, and is NOT reported converged; its neighbours are unaffected."""
    obs, load = models
    bad = load.copy()
    bad[1] = 1.3
    res = calibrate_batch(OracleEngine(obs, bad), gradient="adjoint", compact=0)
    good = calibrate_batch(OracleEngine(obs, load), gradient="adjoint", compact=0)
    assert not bool(res.converged[1]) and not np.isfinite(float(res.obj[1])) and torch.equal(res.alpha[1], torch.full((4,), 10.0, dtype=torch.float64))
    keep = [0, 2, 3, 4]
    assert bool(res.converged[keep].all()) and torch.equal(res.alpha[keep], good.alpha[keep])


def test_iteration_limit_and_standard_errors(models):
    """``maxiter`` reached: nobody is converged, the iterates are the first ones of the full run; ``stderr=True`` differences
    the exact gradient into a Hessian -- against a central-difference Hessian of the oracle's objective."""
    obs, load = models
    short = calibrate_batch(OracleEngine(obs, load), gradient="adjoint", compact=0, maxiter=3)
    assert short.nit == 3 and not bool(short.converged.any()) and bool((short.nit_model == 3).all())
    # the same budget of ITERATIONS whichever schedule runs (round-5 advice: with every model on its own line search a pass of
    # the loop is one trial point, and maxiter used to count passes there): a model gets 3 accepted steps in both, and the
    # iterates of the two schedules are the same
    own = calibrate_batch(OracleEngine(obs, load), gradient="adjoint", compact=0, maxiter=3, own_search_above=0)
    assert own.nit == 3 and bool((own.nit_model == 3).all()) and own.passes >= 3 and not bool(own.converged.any())
    assert torch.allclose(own.alpha, short.alpha, rtol=0, atol=1e-12) and torch.allclose(own.obj, short.obj, rtol=0, atol=1e-9)
    with pytest.raises(ValueError, match="history"):
        calibrate_batch(OracleEngine(obs, load), history=17)
    with pytest.raises(ValueError, match="maxiter"):
        calibrate_batch(OracleEngine(obs, load), maxiter=0)
    eng = OracleEngine(obs, load)
    res = calibrate_batch(eng, gradient="adjoint", compact=0, stderr=True)
    assert bool(res.converged.all()) and bool((res.obj < short.obj).all())
    assert eng.log[-1] == ("forward+backward", (eng.n + 1) * eng.R)      # the Hessian's n + 1 gradients: one launch pair
    r = 2

    def f(a):
        phi, q = phi_q_from_alpha(a[None], load[r][None])
        return float(oracle.dfm_batch(obs[r][None], phi, q, load[r][None], smooth=False, outputs="mle")["mle"][0])

    x, h = res.alpha[r].numpy(), 1e-3
    H = np.empty((4, 4))
    for i in range(4):
        for j in range(4):
            ei, ej = np.eye(4)[i] * h * x[i], np.eye(4)[j] * h * x[j]
            H[i, j] = (f(x + ei + ej) - f(x + ei - ej) - f(x - ei + ej) + f(x - ei - ej)) / (4 * h * h * x[i] * x[j])
    want = np.sqrt(np.diag(np.linalg.pinv(H)))
    np.testing.assert_allclose(res.stderr[r].numpy(), want, rtol=2e-2)
    assert res.pcov.shape == (5, 4, 4)


def test_unknown_gradient_mode(models):
    with pytest.raises(ValueError):
        calibrate_batch(OracleEngine(*models), gradient="central")
