"""Solver plug point A: ``Metran.solve(solver=HipSolve)``.

Mirror of ``metran.solver.ScipySolve`` (/root/reference/metran/solver.py:195-305): same constructor
(``solver(mt=...)``, metran/metran.py:1030-1034), same ``solve(method="l-bfgs-b", **kwargs) ->
(success, optimal, stderr)`` contract (:1039-1042) and the attributes ``fit_report`` reads
(``_name, obj_func, nfev, aic, pcov, pcor, result``; metran/metran.py:1036, 1108-1110, 1157).

What changes is WHERE the objective is evaluated.  The reference gives scipy no gradient, so
L-BFGS-B takes 2-point finite differences: P+1 *sequential* ``Metran.get_mle`` calls per gradient, each
rebuilding the matrices through pandas look-ups (metran/metran.py:386-416) and re-running the filter
(SURVEY.md section 3.1).  Here the observations and loadings are uploaded once, and every gradient
is ONE kernel launch: the P+1 parameter vectors ``x, x + h e_1, ..., x + h e_P`` ride as P+1 filter
instances that share the uploaded record (``mk_params_from_alpha`` + ``mk_loglik``).  The finite-difference
rule is scipy's own for L-BFGS-B (forward difference, absolute step ``eps = 1e-8``, flipped at an
upper bound), so the iteration follows the reference's path up to rounding in the objective.
The FD Hessian fallback for the standard errors (``BaseSolver._get_covariance``, :65-140) is batched
the same way: its (P+1)^2 evaluations are one launch.
"""
import logging

import numpy as np

logger = logging.getLogger(__name__)

__all__ = ["HipSolve", "HipSolveAdjoint", "BatchObjective"]


class BatchObjective:
    """-2 log L of one Metran model for many parameter vectors per launch.

    Parameters
    ----------
    obs : array [T,N]       standardised observations, NaN = missing (``mt.oseries.values``)
    loadings : array [N,K]  ``mt.factors``
    order : sequence of int, optional
        ``state index -> position in the parameter vector`` (Metran looks parameters up by NAME,
        metran/metran.py:283-290; after ``solve()`` the order is sdf_1..sdf_N, cdf_1..cdf_K).
    dt : float  time step in days (metran/metran.py:262)
    """

    def __init__(self, obs, loadings, order=None, dt=1.0, engine=None, warmup=1):
        from .engine import BatchedKalman

        self.kf = engine if engine is not None else BatchedKalman()
        obs = np.asarray(obs, dtype=np.float64)
        loadings = np.asarray(loadings, dtype=np.float64)
        self.kf.set_observations(obs[None]).set_loadings(loadings[None])
        n = loadings.shape[0] + loadings.shape[1]
        self.order = np.arange(n) if order is None else np.asarray(order, dtype=np.int64)
        self.dt = float(dt)
        self.warmup = int(warmup)
        self.nfev = 0       # objective evaluations (filter instances), comparable with the reference's count
        self.launches = 0

    def __call__(self, X):
        """X [S,P] (or [P]) parameter vectors in the caller's order -> float64 array [S]."""
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        alpha = X[:, self.order]
        phi, q = self.kf.params_from_alpha(alpha, dt=self.dt)
        mle = self.kf.loglik(phi, q, warmup=self.warmup)
        self.nfev += X.shape[0]
        self.launches += 1
        return mle.cpu().numpy()

    def value_and_grad(self, x):
        """One parameter vector (caller's order) -> (f, df/dx) by the adjoint kernel."""
        x = np.asarray(x, dtype=np.float64)
        mle, galpha = self.kf.loglik_grad_alpha(x[self.order][None], dt=self.dt, warmup=self.warmup)
        g = np.zeros_like(x)
        g[self.order] = galpha[0].cpu().numpy()
        self.nfev += 1
        self.launches += 2
        return float(mle[0]), g


def _state_order(mt):
    """Position of each state's alpha in ``mt.parameters`` (name look-up as in metran.py:283-290)."""
    names = list(mt.parameters.index)
    want = [str(s) + "_sdf_alpha" for s in mt.snames]
    want += ["cdf%d_alpha" % (k + 1) for k in range(mt.nfactors)]
    return np.array([names.index(w) for w in want], dtype=np.int64)


def _obs_from(mt):
    kf = getattr(mt, "kf", None)
    if kf is not None and getattr(kf, "_obs_nan", None) is not None:
        return kf._obs_nan
    if kf is not None and getattr(kf, "observations", None) is not None:
        from .kalmanfilter import observations_to_nan_encoded

        return observations_to_nan_encoded(kf.observations, kf.observation_indices, kf.observation_count)
    return np.asarray(mt.oseries.values, dtype=np.float64)


def _isnone(v):
    return v is None or (isinstance(v, float) and np.isnan(v))


class HipSolve:
    """Drop-in for ``metran.solver.ScipySolve`` with the objective and its finite-difference gradient
    evaluated on the GPU, one launch per gradient."""

    _name = "HipSolve"

    def __init__(self, mt, gradient="fd", **kwargs):
        """``gradient="fd"`` (default) follows the reference's finite-difference path; ``"adjoint"`` hands
        scipy the exact gradient from ``mk_loglik_grad``: the same optimum in ~6x
        fewer filter runs, but not the reference's iteration path."""
        if gradient not in ("fd", "adjoint"):
            raise ValueError("gradient must be 'fd' or 'adjoint'")
        self.gradient = gradient
        self.mt = mt
        self.pcov = None
        self.pcor = None
        self.nfev = None
        self.result = None
        self._obj = None

    # ------------------------------------------------------------------ objective
    def _objective(self):
        if self._obj is None:
            from .params import dt_days

            mt = self.mt
            freq = mt.settings.get("freq", "D") if hasattr(mt, "settings") else "D"
            self._obj = BatchObjective(_obs_from(mt), mt.factors, order=_state_order(mt), dt=dt_days(freq))
        return self._obj

    def _array_todict(self, p):
        """metran/solver.py:290-305"""
        par = self.initial
        par[self.vary] = p
        return par

    def objfunction(self, p, callback=None):
        """metran/solver.py:42-63 (single evaluation)."""
        if callback is not None:
            p = callback(p)
        return float(self._objective()(np.asarray(p, dtype=np.float64))[0])

    def _fun_and_grad(self, x, eps, ub):
        """f(x) and scipy's 2-point forward-difference gradient from ONE launch of P+1 instances."""
        x = np.asarray(x, dtype=np.float64)
        P = x.size
        if self.gradient == "adjoint":
            obj = self._objective()
            f, g = obj.value_and_grad(self._array_todict(x).astype(np.float64))
            return f, g[np.nonzero(self.vary)[0]]
        h = np.full(P, eps)
        h[np.isfinite(ub) & (x + h > ub)] = -eps  # scipy flips the step at an upper bound
        full = np.tile(self._array_todict(x).astype(np.float64), (P + 1, 1))
        vidx = np.nonzero(self.vary)[0]
        for i in range(P):
            full[i + 1, vidx[i]] += h[i]
        dx = full[np.arange(1, P + 1), vidx] - full[0, vidx]  # the REALISED step (x + h) - x, as scipy divides
        f = self._objective()(full)
        return float(f[0]), (f[1:] - f[0]) / dx

    # ------------------------------------------------------------------ solve
    def solve(self, method="l-bfgs-b", **kwargs):
        """metran/solver.py:222-288."""
        from pandas import DataFrame
        from scipy.optimize import minimize

        self.vary = self.mt.parameters.vary.values.astype(bool)
        self.initial = self.mt.parameters.initial.values.astype(np.float64).copy()
        parameters = self.mt.parameters.loc[self.vary]
        raw = [(b[0], b[1]) for b in parameters.loc[:, ["pmin", "pmax"]].values]
        ub = np.array([np.inf if _isnone(b[1]) else float(b[1]) for b in raw])
        bounds = [(None if _isnone(b[0]) else float(b[0]), None if _isnone(b[1]) else float(b[1])) for b in raw]
        opts = kwargs.get("options") or {}
        eps = float(opts.get("eps", 1e-8))

        self.result = minimize(fun=lambda x: self._fun_and_grad(x, eps, ub), jac=True, method=method,
                               x0=parameters.initial.values.astype(np.float64), bounds=bounds, **kwargs)

        _stderr = np.zeros(parameters.shape[0]) * np.nan
        pcov = None
        if hasattr(self.result, "hess_inv"):
            hi = self.result.hess_inv
            pcov = hi.todense() if hasattr(hi, "todense") else np.asarray(hi)
            _stderr = np.sqrt(np.diag(pcov))
        if self.gradient == "adjoint":
            pcov = self._get_covariance_adjoint(self.result.x)  # exact-gradient Hessian instead of L-BFGS's estimate
            _stderr = np.sqrt(np.diag(pcov))
        if pcov is None or np.isnan(_stderr).any():
            pcov = self._get_covariance(self.result.x)  # finite-difference Hessian, (P+1)^2 instances, one launch
            _stderr = np.sqrt(np.diag(pcov))

        optimal = self.initial
        optimal[self.vary] = self.result.x
        stderr = np.zeros(len(optimal)) * np.nan
        stderr[self.vary] = _stderr

        names = parameters.index.values
        self.pcov = DataFrame(pcov, index=names, columns=names)
        self.pcor = self._get_correlations(self.pcov)

        obj = self._objective()
        self.nfev = obj.nfev            # every filter instance evaluated (reference: one per get_mle call)
        self.launches = obj.launches
        self.aic = 2 * parameters.shape[0] + self.result.fun
        self.obj_func = self.result.fun
        success = self.result.success if hasattr(self.result, "success") else True
        self._leave_filter_at(optimal)
        return success, optimal, stderr

    def _leave_filter_at(self, optimal):
        """The reference's solver evaluates ``Metran.get_mle`` (metran/metran.py:619-621), so when it returns
        ``mt.kf`` holds the matrices of the last evaluated point and a filter run; ``Metran.solve`` then
        stores the optimum and the default-``p`` accessors (``get_state_means()``, ``get_simulation(name)``,
        ``decompose_simulation(name)``: ``_run_kalman(p=None)``, :978-989) run the smoother on whatever
        ``mt.kf`` holds.  This solver never touches ``mt.kf`` while iterating, so the optimum is pushed into
        it here (one B=1 filter run through the model's own engine)."""
        mt = self.mt
        if not (hasattr(mt, "_get_matrices") and getattr(mt, "kf", None) is not None):
            return
        from pandas import Series

        p = Series(optimal, index=mt.parameters.index)
        mt.kf.set_matrices(*mt._get_matrices(p))
        if hasattr(mt.kf, "init_states"):
            mt.kf.init_states()  # results of an earlier solve are stale (metran.py:984, 989 test for None)
        mt.kf.run_filter()

    # ------------------------------------------------------------------ covariance helpers
    def _get_covariance_adjoint(self, x0, rel_step=1e-5):
        """Covariance = pinv(Hessian) with the Hessian from forward differences of the ADJOINT gradient:
        P+1 gradient evaluations in one forward + one backward launch (the reference nests two finite
        differences of the objective, (P+1)^2 evaluations, metran/solver.py:65-140)."""
        x0 = np.asarray(x0, dtype=np.float64)
        n = x0.shape[0]
        vidx = np.nonzero(self.vary)[0]
        obj = self._objective()
        d = rel_step * np.maximum(np.abs(x0), 0.1)
        pts = np.tile(self._array_todict(x0).astype(np.float64), (n + 1, 1))
        for j in range(n):
            pts[1 + j, vidx[j]] += d[j]
        _, g = obj.kf.loglik_grad_alpha(pts[:, obj.order], dt=obj.dt, warmup=obj.warmup)
        g = g.cpu().numpy()
        gfull = np.zeros_like(pts)
        gfull[:, obj.order] = g
        gv = gfull[:, vidx]
        obj.nfev += n + 1
        obj.launches += 2
        hessian = (gv[1:] - gv[0]) / d[:, None]      # row j: d grad / d x_j
        hessian = 0.5 * (hessian + hessian.T)
        return np.linalg.pinv(hessian)

    def _get_covariance(self, x0, epsilon=None):
        """``BaseSolver._get_covariance`` (metran/solver.py:65-140, forward scheme) with every objective
        evaluation of the nested finite differences in ONE launch."""
        x0 = np.asarray(x0, dtype=np.float64)
        n = x0.shape[0]
        if epsilon is None:
            epsilon = np.finfo(float).eps ** 0.25
        vidx = np.nonzero(self.vary)[0]
        base = self._array_todict(x0).astype(np.float64)
        cov = None
        for _ in range(4):  # the reference's epsilon-growth retry (:100-138), bounded
            d = epsilon * np.maximum(np.abs(x0), 0.1)
            # rows: 0 = x0; 1..n = x0 + eps e_i; then for each j: x0 + d_j e_j, x0 + d_j e_j + eps e_i
            pts = np.tile(base, (1 + n + n * (n + 1), 1))
            for i in range(n):
                pts[1 + i, vidx[i]] += epsilon
            for j in range(n):
                o = 1 + n + j * (n + 1)
                pts[o:o + n + 1, vidx[j]] += d[j]
                for i in range(n):
                    pts[o + 1 + i, vidx[i]] += epsilon
            f = self._objective()(pts)
            f0 = (f[1:1 + n] - f[0]) / epsilon  # approx_fprime(x0)
            hessian = np.zeros((n, n))
            for j in range(n):
                o = 1 + n + j * (n + 1)
                ff = (f[o + 1:o + 1 + n] - f[o]) / epsilon  # approx_fprime(x0 + d_j e_j)
                hessian[: j + 1, j] = (ff[: j + 1] - f0[: j + 1]) / d[j]
                hessian[j, : j + 1] = hessian[: j + 1, j]
            if not np.isnan(hessian).any():
                cov = np.linalg.pinv(hessian)
                if np.amin(np.diag(cov)) <= 0:  # the reference's repair step (:127-133)
                    try:
                        cov = np.linalg.pinv(self._nearPSD(hessian))
                    except Exception as e:  # noqa: BLE001 -- as the reference: keep going with the raw pinv
                        logger.debug("Could not calculate 'cov': %s", e)
                if np.amin(np.diag(cov)) > 0:
                    return cov
            epsilon *= 10.0
        if cov is None:
            msg = "HipSolve: the finite-difference Hessian is NaN at every step size; no covariance of the estimates"
            logger.error(msg)
            raise Exception(msg)
        logger.warning("HipSolve: covariance of the estimates has a non-positive diagonal (stderr will hold NaN)")
        return cov

    @staticmethod
    def _nearPSD(A, epsilon=0.0):
        """``BaseSolver._nearPSD`` (metran/solver.py:167-192) AS IT EVALUATES on ndarrays: the eigenvalues are clipped at
        ``epsilon`` and the rows rescaled by ``t_i = 1 / sum_k V_ik^2 w_k``, but the reference's ``T @ vec * diag(sqrt(val))``
        multiplies ELEMENT-wise by a diagonal matrix (the expression dates from ``np.matrix``), so only the diagonal of the
        scaled eigenvector matrix survives and the result is the diagonal matrix ``diag(t_i V_ii^2 w_i)`` -- which is what a
        drop-in has to return (tests/golden/solver_covariance.npz holds the reference's own outputs).  ``np.linalg.eig``
        as there: the pairing of row i with eigenpair i follows LAPACK's order."""
        A = np.asarray(A, dtype=np.float64)
        w, V = np.linalg.eig(A)
        w = np.maximum(w, epsilon)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = 1.0 / ((V * V) @ w)
            b = np.sqrt(t) * np.diag(V) * np.sqrt(w)
        return np.diag(b * b)

    @staticmethod
    def _get_correlations(pcov):
        """metran/solver.py:142-165"""
        d = np.sqrt(np.diag(pcov.values))
        pcor = pcov.copy()
        pcor.loc[:, :] = pcov.values / np.outer(d, d)
        return pcor


class HipSolveAdjoint(HipSolve):
    """``HipSolve`` with the exact (adjoint) gradient, as a CLASS: ``Metran.solve`` tests
    ``issubclass(solver, ...)`` (metran/metran.py:1033), so a lambda or ``functools.partial`` cannot be
    passed -- ``mt.solve(solver=HipSolveAdjoint)``."""

    _name = "HipSolveAdjoint"

    def __init__(self, mt, **kwargs):
        kwargs.pop("gradient", None)
        super().__init__(mt, gradient="adjoint", **kwargs)
