"""Batched factor analysis on the MI355X: the loadings ``Metran.factors``, the number
of common factors and the eigenvalue table of EVERY model of a batch in a handful of launches.

Mirror of ``metran.factoranalysis.FactorAnalysis`` (/root/reference/metran/factoranalysis.py:13-460):

* ``FactorAnalysisBatch``  -- R models at once (device tensors in, device tensors out);
* ``FactorAnalysis``       -- the reference class for ONE model (same constructor, ``solve(oseries)``, ``eigval``,
  ``factors``, ``fep``, ``get_eigval_weight`` and the static helpers the reference's tests call), a batch of one;
* ``install(metran_module)`` makes ``Metran.get_factors`` (metran/metran.py:199-226) use it.

All arithmetic runs in ``libmetran_hip.so`` (``mk_fa_*``, csrc/mk_factor.hip); there is no CPU fallback.

What the reference's ``_minres`` (:173-217) returns.  It hands scipy's L-BFGS-B an objective (``_minresfun``,
:315-347) built from the nf SMALLEST eigenpairs of the reduced correlation matrix (``eigh`` is ascending and the code
takes ``[:nf]``; for nf = 1 the "model" is even the scalar ``l.l``, :341-343) together with the jacobian of the PROPER
minres fit (``_minresgrad``, :349-373).  The two do not belong to the same function, so for most models the first line
search finds no decrease and L-BFGS-B returns its START vector after 21 evaluations ("ABNORMAL", nit = 0): examples/data
and 4 of the 6 synthetic models of tests/golden/factor_analysis.npz.  The loadings are then ``_get_loadings(psi0)`` with
``psi0 = clip(1/diag(inv(S)), 0.005, 1)`` -- closed form.  Whether a model stalls is CHECKED on the device
(``stalled``): the reference's own objective is evaluated at 20 step lengths along the projected-gradient direction
and must fail the sufficient-decrease test at each.  A model that does not stall (the seeded notebook model ``g2`` and
``s6k1`` of the fixture) is handed to the SAME scipy routine on the host, objective and jacobian evaluated by the
kernels, so that it follows the reference's iteration path there too (host-driven scipy with device objectives is how
``HipSolve`` works as well).  ``_get_loadings`` takes ``eigvec[:, :nf]`` of LAPACK's UNSORTED ``eig`` (:396-398); the
kernels take the nf largest pairs, which is what ``eig`` returns first for every fixture (and for 95 % of random
matrices); where it does not, the reference's loadings are those of a non-dominant eigenvector and are not reproduced.
"""
import ctypes
import logging

import numpy as np

from ._lib import MetranHipError, check

logger = logging.getLogger(__name__)

__all__ = ["FactorAnalysisBatch", "FactorAnalysis", "FactorResult", "install", "uninstall"]

_PSI_LO, _PSI_HI = 0.005, 1.0  # bounds of the reference's minimisation (:205-207)


class FactorResult(dict):
    """``factors [R,N,KMAX]`` (columns >= nfactors[r] are zero), ``nfactors [R]``, ``nfactors_map``/``nfactors_map4
    [R]`` (Velicer's MAP tests), ``eigval [R,N]`` (descending), ``fep [R]``, ``corr [R,N,N]``, ``psi [R,N]``,
    ``status [R]`` (1 = no factors can be derived), ``stalled [R]`` (bool, see module docstring)."""

    __getattr__ = dict.__getitem__


class FactorAnalysisBatch:
    """Factor analysis of R models with N series each.

    Parameters
    ----------
    maxfactors : int, optional   as ``FactorAnalysis(maxfactors)`` (:29-30)
    engine : BatchedKalman, optional   supplies the device context (default: the process-wide engine)
    """

    def __init__(self, maxfactors=None, engine=None):
        self.maxfactors = maxfactors
        if engine is None:
            from .kalmanfilter import get_engine

            engine = get_engine()
        self.kf = engine

    # ------------------------------------------------------------------ thin wrappers of the C ABI
    def _call(self, name, *args):
        self.kf._bind_stream()
        check(getattr(self.kf._L, name)(self.kf._ctx, *args))

    def correlations(self, obs, time_major=None):
        """``_get_correlations`` (:404-418) for records ``obs [R,T,N]`` (NaN = missing) -> ``[R,N,N]``."""
        import torch

        kf = self.kf
        if not isinstance(obs, torch.Tensor):
            obs = torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float64))
        obs = obs.to(device=kf.device, dtype=torch.float64)
        if obs.ndim == 2:
            obs = obs[None]
        R, T, N = (int(s) for s in obs.shape)
        if N > 64:
            raise MetranHipError("factor analysis supports N <= 64 series per model")
        tm = obs.transpose(0, 1).is_contiguous() and not obs.is_contiguous() if time_major is None else bool(time_major)
        if not tm:
            obs = obs.contiguous()
        corr = torch.empty((R, N, N), dtype=torch.float64, device=kf.device)
        self._call("mk_fa_correlation", R, T, N, 1 if tm else 0, kf._p(obs), kf._p(corr))
        return corr

    def analyse(self, corr):
        """Eigenvalues, MAP tests, factor count and minres start vector of ``corr [R,N,N]``."""
        import torch

        kf = self.kf
        corr = kf._dev(corr)
        R, N = int(corr.shape[0]), int(corr.shape[1])
        out = dict(eigval=torch.empty((R, N), dtype=torch.float64, device=kf.device),
                   psi0=torch.empty((R, N), dtype=torch.float64, device=kf.device),
                   status=torch.zeros(R, dtype=torch.int32, device=kf.device))
        for k in ("nfactors", "nfactors_map", "nfactors_map4"):
            out[k] = torch.zeros(R, dtype=torch.int64, device=kf.device)
        self._call("mk_fa_analyse", R, N, int(self.maxfactors or 0), kf._p(corr), kf._p(out["eigval"]),
                   kf._p(out["nfactors"]), kf._p(out["nfactors_map"]), kf._p(out["nfactors_map4"]), kf._p(out["psi0"]),
                   kf._p(out["status"]))
        return out

    def minres_eval(self, corr, nfactors, psi, kmax, want=("f", "g", "loadings")):
        """``(_minresfun, _minresgrad, _get_loadings)`` (:315-401) at ``psi [B,N]``; instance b uses model b % R."""
        import torch

        kf = self.kf
        psi = kf._dev(psi)
        B, N = int(psi.shape[0]), int(psi.shape[1])
        R = int(corr.shape[0])
        f = torch.empty(B, dtype=torch.float64, device=kf.device) if "f" in want else None
        g = torch.empty((B, N), dtype=torch.float64, device=kf.device) if "g" in want else None
        ld = torch.empty((B, N, kmax), dtype=torch.float64, device=kf.device) if "loadings" in want else None
        self._call("mk_fa_minres", B, R, N, int(kmax), kf._p(corr), kf._p(nfactors), kf._p(psi), kf._p(f), kf._p(g), kf._p(ld))
        return f, g, ld

    def eigh(self, sym):
        """Eigenvalues (descending) and eigenvectors (columns) of symmetric ``sym [B,N,N]``."""
        import torch

        kf = self.kf
        sym = kf._dev(sym)
        B, N = int(sym.shape[0]), int(sym.shape[1])
        val = torch.empty((B, N), dtype=torch.float64, device=kf.device)
        vec = torch.empty((B, N, N), dtype=torch.float64, device=kf.device)
        self._call("mk_fa_eigh", B, N, kf._p(sym), kf._p(val), kf._p(vec))
        return val, vec

    # ------------------------------------------------------------------ FactorAnalysis.solve for R models
    def solve(self, obs=None, corr=None):
        """``FactorAnalysis.solve`` (:42-119) for every record: ``obs [R,T,N]`` (default: the engine's records) or
        precomputed correlation matrices.  Returns a ``FactorResult`` of device tensors."""
        import torch

        kf = self.kf
        if corr is None:
            if obs is None:
                if kf.obs is None:
                    raise MetranHipError("no observations: pass obs or call set_observations on the engine first")
                obs, tm = kf.obs, kf.time_major
            else:
                tm = None
            corr = self.correlations(obs, time_major=tm)
        else:
            corr = kf._dev(corr)
            if corr.ndim == 2:
                corr = corr[None]
        R, N = int(corr.shape[0]), int(corr.shape[1])
        a = self.analyse(corr)
        nf = a["nfactors"]
        kmax = max(1, int(nf.max().item()))
        psi = a["psi0"]
        f0, g0, ld = self.minres_eval(corr, nf, psi, kmax)
        # ---- does the reference's optimiser stall at its start vector? (module docstring) ----
        lo = torch.full_like(psi, _PSI_LO)
        hi = torch.full_like(psi, _PSI_HI)
        d = torch.minimum(torch.maximum(psi - g0, lo), hi) - psi          # projected-gradient step
        pgnorm = d.abs().amax(1)
        steps = 2.0 ** -torch.arange(20, dtype=torch.float64, device=kf.device)
        trial = (psi[None] + steps[:, None, None] * d[None]).reshape(20 * R, N)  # instance k*R + r -> model r
        ft, _, _ = self.minres_eval(corr, nf, trial, kmax, want=("f",))
        ft = ft.reshape(20, R)
        slope = (g0 * d).sum(1)
        armijo = ft <= f0[None] + 1e-3 * steps[:, None] * slope[None]
        stalled = ~(armijo.any(0) & (pgnorm > 1e-5))
        ok = (a["status"] == 0) & (nf > 0)
        moving = (~stalled) & ok
        if bool(moving.any()):
            idx = torch.nonzero(moving).reshape(-1).tolist()
            logger.warning("factor analysis: scipy's L-BFGS-B can leave the start vector for %d model(s); running the "
                           "reference's minimisation for them (objective/jacobian on the device)", len(idx))
            psi = psi.clone()
            for r in idx:
                psi[r] = torch.from_numpy(self._host_minres(corr[r:r + 1], nf[r:r + 1], psi[r].cpu().numpy(), kmax)).to(psi)
            _, _, ld = self.minres_eval(corr, nf, psi, kmax, want=("loadings",))
        # ---- rotation + sign convention (:84-108) ----
        self._call("mk_fa_rotate", R, N, kmax, kf._p(nf), kf._p(ld), 1.0, 20, 1e-6)
        nonzero = (ld != 0).flatten(1).any(1)
        good = ok & nonzero
        nf_out = torch.where(good, nf, torch.zeros_like(nf))
        ld = torch.where(good[:, None, None], ld, torch.zeros_like(ld))
        ev = a["eigval"]
        w = ev / ev.sum(1, keepdim=True)                                  # get_eigval_weight (:32-40)
        fep = 100.0 * (w * (torch.arange(N, device=kf.device)[None, :] < nf_out[:, None])).sum(1)  # :112
        if bool((~good).any()):
            logger.warning("No proper common factors could be derived from series. (%d of %d models)",
                           int((~good).sum()), R)  # the reference's message (:115-116)
        return FactorResult(factors=ld, nfactors=nf_out, nfactors_map=a["nfactors_map"], nfactors_map4=a["nfactors_map4"],
                            eigval=ev, fep=fep, corr=corr, psi=psi, status=torch.where(good, a["status"], torch.ones_like(a["status"])),
                            stalled=stalled, kmax=kmax)

    def _host_minres(self, corr1, nf1, start, kmax):
        """The reference's ``scopt.minimize(..., method="L-BFGS-B", jac=..., bounds=(0.005, 1))`` (:209-216) for ONE
        model, objective and jacobian from ``mk_fa_minres``."""
        import scipy.optimize as scopt

        def fun(x):
            f, _, _ = self.minres_eval(corr1, nf1, np.asarray(x, dtype=np.float64)[None], kmax, want=("f",))
            return float(f[0].item())

        def jac(x):
            _, g, _ = self.minres_eval(corr1, nf1, np.asarray(x, dtype=np.float64)[None], kmax, want=("g",))
            return g[0].cpu().numpy()

        res = scopt.minimize(fun, start, method="L-BFGS-B", jac=jac, bounds=[(_PSI_LO, _PSI_HI)] * len(start))
        return np.asarray(res.x, dtype=np.float64)


class FactorAnalysis:
    """``metran.factoranalysis.FactorAnalysis`` with the MI355X engine behind it (one model = a batch of one)."""

    def __init__(self, maxfactors=None):
        self.maxfactors = maxfactors
        self.factors = None
        self.eigval = None
        self.fep = None

    def _batch(self):
        return FactorAnalysisBatch(maxfactors=self.maxfactors)

    def get_eigval_weight(self):
        """:32-40"""
        return self.eigval / np.sum(self.eigval)

    def solve(self, oseries):
        """:42-119 -- ``oseries``: DataFrame (or array [T,N]); returns the loadings ``[N, nfactors]`` or None."""
        y = np.asarray(getattr(oseries, "values", oseries), dtype=np.float64)
        res = self._batch().solve(obs=y[None])
        self.eigval = res.eigval[0].cpu().numpy()
        nf = int(res.nfactors[0].item())
        self.stalled = bool(res.stalled[0].item())
        if nf == 0:
            self.factors = None
            return None
        self.factors = np.atleast_2d(res.factors[0, :, :nf].cpu().numpy())
        self.fep = float(res.fep[0].item())
        return self.factors

    @staticmethod
    def _get_correlations(oseries):
        """:404-418"""
        y = np.asarray(getattr(oseries, "values", oseries), dtype=np.float64)
        return FactorAnalysisBatch().correlations(y[None])[0].cpu().numpy()

    @staticmethod
    def _get_eigval(correlation):
        """:420-460 -- eigenvalues descending (negatives set to 0) and eigenvectors scaled by their square roots."""
        val, vec = FactorAnalysisBatch().eigh(np.asarray(correlation, dtype=np.float64)[None])
        val = np.maximum(val[0].cpu().numpy(), 0.0)
        return val, np.atleast_2d(vec[0].cpu().numpy() * np.sqrt(val)[None, :])

    @staticmethod
    def _maptest(cov, eigvec=None, eigval=None):
        """:220-312 -- (nfacts, nfacts4).  The decomposition is recomputed on the device from ``cov``."""
        a = FactorAnalysisBatch().analyse(np.asarray(cov, dtype=np.float64)[None])
        return int(a["nfactors_map"][0].item()), int(a["nfactors_map4"][0].item())

    def _minres(self, s, nf, covar=False):
        """:173-217"""
        import torch

        fb = self._batch()
        corr = fb.kf._dev(np.asarray(s, dtype=np.float64)[None])
        a = fb.analyse(corr)
        if int(a["status"][0].item()) != 0:
            return None
        nft = torch.full((1,), int(nf), dtype=torch.int64, device=fb.kf.device)
        _, _, ld = fb.minres_eval(corr, nft, a["psi0"], max(1, int(nf)), want=("loadings",))
        return ld[0].cpu().numpy()


_PATCHED = {}


def install(metran_module=None):
    """``Metran.get_factors`` constructs ``FactorAnalysis()`` by its global name in ``metran.metran``
    (metran/metran.py:217); replacing that name (and ``metran.factoranalysis.FactorAnalysis``) is sufficient."""
    if metran_module is None:
        import metran as metran_module
    mm = metran_module.metran
    if mm not in _PATCHED:
        _PATCHED[mm] = (mm.FactorAnalysis, metran_module.factoranalysis.FactorAnalysis)
    mm.FactorAnalysis = FactorAnalysis
    metran_module.factoranalysis.FactorAnalysis = FactorAnalysis
    return mm


def uninstall(metran_module=None):
    if metran_module is None:
        import metran as metran_module
    mm = metran_module.metran
    if mm in _PATCHED:
        mm.FactorAnalysis, metran_module.factoranalysis.FactorAnalysis = _PATCHED.pop(mm)
