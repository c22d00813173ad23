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
``HipSolve`` works as well); all models that may move run in LOCK-STEP: one scipy minimisation per model on its own
host thread, the objective/jacobian requests of every model still running served by ONE ``mk_fa_minres`` launch per
step (``_lockstep_minres``).  The stall check has a safety band: a model whose best trial point comes within a factor
1000 of the sufficient-decrease threshold is treated as moving (over 300 random models the ratio is <= 1e-12 for the
models scipy leaves at their start vector and >= 0.07 for the others; the threshold is 1e-3, the band starts at 1e-6).

``_get_loadings`` takes ``eigvec[:, :nf]`` of LAPACK's UNSORTED ``eig`` (:396-398).  For a quarter of random 20- and
32-series models with two factors those are not the nf largest pairs, and the reference's loadings (and, through
``_minresgrad``, its optimiser path) are then those of a non-dominant eigenvector.  The order in which a 64 x 64
``dgeev`` deflates its eigenvalues is not something to emulate on the device: ``eig_order`` asks the very routine the
reference calls (one batched ``numpy.linalg.eig`` on the host) for the ORDER only -- the rank of each of its first nf
eigenvalues -- and ``mk_fa_minres`` builds the loadings from the device's own decomposition of those pairs.
"""
import logging
import os
import queue
import threading

import numpy as np

from ._lib import MetranHipError, check

logger = logging.getLogger(__name__)

__all__ = ["FactorAnalysisBatch", "FactorAnalysis", "FactorResult", "eig_order", "install", "uninstall"]

_PSI_LO, _PSI_HI = 0.005, 1.0  # bounds of the reference's minimisation (:205-207)
_STACK_SIZE_LOCK = threading.Lock()
_LOCKSTEP_THREADS = 512        # scipy minimisations in flight at once (one host thread each)


def eig_order(corr, psi, kmax):
    """ORDER in which ``numpy.linalg.eig`` (LAPACK dgeev: the routine ``_get_loadings`` calls, :396) returns the
    eigenpairs of ``psi^-1/2 S psi^-1/2``: ``[B,kmax]`` int64, entry f = the rank (0 = largest eigenvalue) of the
    f-th pair returned.  ``corr [R,N,N]``, ``psi [B,N]`` host arrays; instance b uses matrix b % R.  The matrix is
    formed exactly as the reference forms it (:394-395, products with diagonal matrices) so that the routine sees the
    same input; only the order of its result is used (the loadings themselves are computed by ``mk_fa_minres``)."""
    corr = np.asarray(corr, dtype=np.float64)
    psi = np.asarray(psi, dtype=np.float64)
    B, N = psi.shape
    sc = 1.0 / np.sqrt(psi)
    sstar = (corr[np.arange(B) % corr.shape[0]] * sc[:, None, :]) * sc[:, :, None]
    out = np.tile(np.arange(kmax, dtype=np.int64), (B, 1))
    good = np.isfinite(sstar).all(axis=(1, 2))
    if good.any():
        w = _eigvals_of_eig(sstar[good])
        pos = np.argsort(-w, axis=1, kind="stable")
        rank = np.empty_like(pos)
        np.put_along_axis(rank, pos, np.broadcast_to(np.arange(N), pos.shape), axis=1)
        out[good] = rank[:, :kmax]
    return out


def _eigvals_of_eig(mats):
    """``numpy.linalg.eig(mats)[0].real`` for a stack of matrices, the stack split over host threads (LAPACK releases
    the GIL; at R = 4096 this call was 3/4 of ``FactorAnalysisBatch.solve`` on one core).  Each matrix goes through the
    same dgeev call as in the reference, so the ORDER of the eigenvalues -- the only thing used -- is the same."""
    B = mats.shape[0]
    workers = min(16, os.cpu_count() or 1, max(1, B // 64))
    if workers <= 1:
        return np.linalg.eig(mats)[0].real
    from concurrent.futures import ThreadPoolExecutor

    bounds = np.linspace(0, B, workers + 1).astype(int)
    with ThreadPoolExecutor(workers) as pool:
        parts = list(pool.map(lambda i: np.linalg.eig(mats[bounds[i]:bounds[i + 1]])[0].real, range(workers)))
    return np.concatenate(parts, axis=0)


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

    def minres_eval(self, corr, nfactors, psi, kmax, want=("f", "g", "loadings"), order="lapack", corr_host=None):
        """``(_minresfun, _minresgrad, _get_loadings)`` (:315-401) at ``psi [B,N]``; instance b uses model b % R.

        ``order``: which eigenpairs ``_get_loadings`` uses as its columns -- ``"lapack"`` (default): the first nf in
        ``numpy.linalg.eig``'s order, as the reference (``eig_order``; only evaluated when the jacobian or the
        loadings are wanted, the objective does not depend on it); ``None``: the nf largest, descending; or an
        explicit ``[B,kmax]`` array of ranks.  ``corr_host``: a host copy of ``corr`` (saves the download)."""
        import torch

        kf = self.kf
        psi = kf._dev(psi)
        B, N = int(psi.shape[0]), int(psi.shape[1])
        R = int(corr.shape[0])
        f = torch.empty(B, dtype=torch.float64, device=kf.device) if "f" in want else None
        g = torch.empty((B, N), dtype=torch.float64, device=kf.device) if "g" in want else None
        ld = torch.empty((B, N, kmax), dtype=torch.float64, device=kf.device) if "loadings" in want else None
        od = None
        if order is not None and (g is not None or ld is not None):
            if isinstance(order, str):
                if order != "lapack":
                    raise MetranHipError("minres_eval: order must be 'lapack', None or an array of ranks")
                order = eig_order(corr.cpu().numpy() if corr_host is None else corr_host, psi.cpu().numpy(), int(kmax))
            od = torch.as_tensor(np.ascontiguousarray(order, dtype=np.int64)).to(kf.device).reshape(B, int(kmax)).contiguous()
        self._call("mk_fa_minres", B, R, N, int(kmax), kf._p(corr), kf._p(nfactors), kf._p(psi), kf._p(od), kf._p(f),
                   kf._p(g), kf._p(ld))
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
    def solve(self, obs=None, corr=None, always_scipy=False):
        """``FactorAnalysis.solve`` (:42-119) for every record: ``obs [R,T,N]`` (default: the engine's records) or
        precomputed correlation matrices.  Returns a ``FactorResult`` of device tensors.  ``always_scipy``: skip the
        device-side stall check and run the reference's minimisation for every model (what the one-model mirror
        class does)."""
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
        ok = (a["status"] == 0) & (nf > 0)
        psi, ld, stalled = self._minres_batch(corr, nf, a["psi0"], ok, kmax, always_scipy=always_scipy)
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

    def _minres_batch(self, corr, nf, psi0, ok, kmax, always_scipy=False):
        """``_minres`` (:173-217) for every model: ``(psi [R,N], unrotated loadings [R,N,kmax], stalled [R])``.
        ``stalled[r]``: scipy's L-BFGS-B returns (or would return) its start vector for model r."""
        import torch

        kf = self.kf
        R, N = int(corr.shape[0]), int(corr.shape[1])
        corr_h = corr.cpu().numpy()
        psi = psi0
        f0, g0, ld = self.minres_eval(corr, nf, psi, kmax, corr_host=corr_h)
        if always_scipy:
            maybe = ok.clone()
        else:
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
            # sufficient decrease is ft <= f0 + 1e-3 step slope; the band down to 1e-6 goes to scipy as well
            near = ft <= f0[None] + 1e-6 * steps[:, None] * slope[None]
            maybe = near.any(0) & (pgnorm > 0.5e-5) & ok
        stalled = torch.ones(R, dtype=torch.bool, device=kf.device)
        if bool(maybe.any()):
            idx = torch.nonzero(maybe).reshape(-1)
            logger.info("factor analysis: scipy's L-BFGS-B may leave the start vector for %d of %d model(s); running the "
                        "reference's minimisation for them in lock-step (objective/jacobian on the device)", len(idx), R)
            x = self._lockstep_minres(corr.index_select(0, idx).contiguous(), corr_h[idx.cpu().numpy()],
                                      nf.index_select(0, idx).contiguous(), psi.index_select(0, idx).cpu().numpy(), kmax)
            x = torch.from_numpy(x).to(psi)
            stalled[idx] = (x == psi.index_select(0, idx)).all(1)
            psi = psi.clone()
            psi[idx] = x
            _, _, ld = self.minres_eval(corr, nf, psi, kmax, want=("loadings",), corr_host=corr_h)
        return psi, ld, stalled

    def _lockstep_minres(self, corr, corr_host, nf, start, kmax):
        """The reference's ``scopt.minimize(_minresfun, start, method="L-BFGS-B", jac=_minresgrad, bounds=(0.005, 1))``
        (:209-216) for M models at once: the SAME public scipy routine per model, each on its own host thread, with
        every round of objective/jacobian requests (one per model still running) answered by ONE ``mk_fa_minres``
        launch plus one batched ``eig_order``.  ``corr [M,N,N]`` (device), ``corr_host`` its host copy, ``nf [M]``,
        ``start [M,N]`` (host) -> ``x [M,N]`` (host).  At most ``_LOCKSTEP_THREADS`` models are in flight at once."""
        import scipy.optimize as scopt
        import torch

        M, N = start.shape
        out = np.array(start, dtype=np.float64)
        bounds = [(_PSI_LO, _PSI_HI)] * N
        dev = self.kf.device

        def run_chunk(lo, hi):
            m = hi - lo
            requests = queue.SimpleQueue()
            answers = [None] * m
            wake = [threading.Event() for _ in range(m)]
            errors = []
            sub_corr, sub_host, sub_nf = corr[lo:hi], corr_host[lo:hi], nf[lo:hi]

            def worker(i):
                cache = {}

                def evaluate(x):
                    x = np.array(x, dtype=np.float64)
                    key = x.tobytes()
                    if key not in cache:
                        requests.put((i, x))
                        wake[i].wait()
                        wake[i].clear()
                        if answers[i] is None:
                            raise MetranHipError("factor analysis: the batched minres evaluation failed")
                        cache.clear()
                        cache[key] = answers[i]
                    return cache[key]

                try:
                    res = scopt.minimize(lambda x: evaluate(x)[0], start[lo + i], method="L-BFGS-B",
                                         jac=lambda x: evaluate(x)[1], bounds=bounds)
                    out[lo + i] = res.x
                except BaseException as e:  # noqa: BLE001 -- reported by the serving thread
                    errors.append(e)
                finally:
                    requests.put((i, None))

            with _STACK_SIZE_LOCK:  # threading.stack_size is process-global: one chunk at a time changes it, and puts it back
                old = threading.stack_size(512 * 1024)
                try:
                    threads = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(m)]
                    for t in threads:
                        t.start()
                finally:
                    threading.stack_size(old)
            live = m
            try:
                while live:
                    batch = {}
                    while len(batch) < live:     # every model still running has asked (or has finished)
                        i, x = requests.get()
                        if x is None:
                            live -= 1
                        else:
                            batch[i] = x
                    if not batch:
                        break
                    ids = sorted(batch)
                    try:
                        X = np.stack([batch[i] for i in ids])
                        sel = torch.as_tensor(ids, device=dev)
                        f, g, _ = self.minres_eval(sub_corr.index_select(0, sel).contiguous(),
                                                   sub_nf.index_select(0, sel).contiguous(), X, kmax, want=("f", "g"),
                                                   corr_host=sub_host[ids])
                        f, g = f.cpu().numpy(), g.cpu().numpy()
                        for k, i in enumerate(ids):
                            answers[i] = (float(f[k]), g[k].copy())
                    except BaseException as e:  # noqa: BLE001
                        errors.append(e)
                        for i in ids:
                            answers[i] = None
                    for i in ids:
                        wake[i].set()
            finally:
                # whatever ends the serving loop (KeyboardInterrupt included): no worker stays blocked on its event -- each
                # finds its answer None, raises inside scipy's callback and finishes
                for i in range(m):
                    if threads[i].is_alive():
                        answers[i] = None
                        wake[i].set()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]

        for lo in range(0, M, _LOCKSTEP_THREADS):
            run_chunk(lo, min(M, lo + _LOCKSTEP_THREADS))
        return out


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
        res = self._batch().solve(obs=y[None], always_scipy=True)
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
        """:173-217 -- the same path as ``solve``: scipy's L-BFGS-B on the device-evaluated objective / jacobian."""
        import torch

        fb = self._batch()
        corr = fb.kf._dev(np.asarray(s, dtype=np.float64)[None])
        a = fb.analyse(corr)
        if int(a["status"][0].item()) != 0:
            return None
        nft = torch.full((1,), int(nf), dtype=torch.int64, device=fb.kf.device)
        ok = torch.ones(1, dtype=torch.bool, device=fb.kf.device)
        _, ld, _ = fb._minres_batch(corr, nft, a["psi0"], ok, max(1, int(nf)), always_scipy=True)
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
