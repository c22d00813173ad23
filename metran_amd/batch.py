"""Many Metran models at once: everything a user of ``metran.Metran`` calls after construction (``get_factors``,
``solve``, ``get_mle``, ``get_simulated_means/variances``, ``get_simulation``, ``decompose_simulation``,
``get_state_means/variances``, ``get_state``, ``mask_observations``; /root/reference/metran/metran.py:199-226,
464-506, 605-989, 991-1045) on top of the batched engine.  The orchestration is Python, as in the reference; every
number comes from the HIP kernels:

    ingest.ObservationBatch  ->  mk_standardize             (Metran.__init__: combine, daily grid, standardise)
    FactorAnalysisBatch      ->  mk_fa_*                     (Metran.get_factors for all models)
    calibrate_batch          ->  mk_loglik_grad / mk_loglik  (Metran.solve for all models in lock-step)
    simulate_smoothed        ->  filter + projecting smoother (get_simulated_means / _variances)
    smooth_state_variances   ->  state means / variances / decomposition

Results of the last filter / smoother run are cached per parameter set, like ``Metran._run_kalman`` does
(metran.py:963-989): asking for another series, the variances after the means, or the decomposition after the
simulation does not launch anything.  With ``torch.distributed`` initialised and ``shard=True`` each rank ingests and
owns a contiguous slice of the models (``distributed.shard_range``); no collective is needed (per-model parameters)
and ``gather`` concatenates per-model results in rank order.
"""
import numpy as np

from .calibrate import calibrate_batch
from .engine import BatchedKalman
from .ingest import ObservationBatch

__all__ = ["MetranBatch"]


class MetranBatch:
    """R independent dynamic-factor models with N series and K common factors each.

    Parameters
    ----------
    models : sequence of whatever ``Metran(oseries)`` accepts (one entry per model)
    factors : array ``[R,N,K]`` or ``[N,K]`` (shared), optional
        factor loadings ``Metran.factors``.  Default None: batched factor analysis of every model
        (``FactorAnalysisBatch``, the reference's ``get_factors``); models with fewer factors than the batch maximum
        K get zero columns (a common factor nobody loads on: it changes no likelihood and no projection).
    maxfactors : as ``FactorAnalysis(maxfactors)``
    shard : bool   with torch.distributed initialised, keep only this rank's slice of ``models``
    device, tmin, tmax, min_pairs : as in ``BatchedKalman`` / ``Metran.settings``
    """

    def __init__(self, models, factors=None, device=None, tmin=None, tmax=None, min_pairs=20, dt=1.0, maxfactors=None,
                 shard=False):
        models = list(models)
        self.n_models_total = len(models)
        self.shard = (0, len(models))
        if shard:
            from .distributed import shard_range, world

            rank, size = world()
            self.shard = shard_range(len(models), rank, size)
            models = models[self.shard[0]:self.shard[1]]
            if factors is not None and np.ndim(factors) == 3:
                factors = np.asarray(factors)[self.shard[0]:self.shard[1]]
        self.batch = ObservationBatch(models, tmin=tmin, tmax=tmax, min_pairs=min_pairs)
        self.kf = BatchedKalman(device, layout="time_major")
        self.batch.upload(self.kf)  # standardised records + (std, mean) scaling on the device
        R, T, N = self.batch.shape
        self.eigval = self.fep = self.nfactors = None
        if factors is None:
            from .factoranalysis import FactorAnalysisBatch

            fa = FactorAnalysisBatch(maxfactors=maxfactors, engine=self.kf).solve()
            self.nfactors = fa.nfactors.cpu().numpy()
            if (self.nfactors == 0).any():
                bad = np.nonzero(self.nfactors == 0)[0]
                # the reference cannot solve such a model either (metran.py:1022-1023: factors is None)
                raise Exception("No proper common factors could be derived from series of model(s) %s"
                                % ", ".join(str(int(b) + self.shard[0]) for b in bad))
            factors = fa.factors.cpu().numpy()
            self.eigval = fa.eigval.cpu().numpy()
            self.fep = fa.fep.cpu().numpy()
        factors = np.asarray(factors, dtype=np.float64)
        if factors.ndim == 2:
            factors = np.broadcast_to(factors, (R,) + factors.shape).copy()
        if self.nfactors is None:
            self.nfactors = np.full(R, factors.shape[2], dtype=np.int64)
        self.kf.set_loadings(factors)
        self.factors = factors
        self.R, self.T, self.N, self.K = R, T, N, int(factors.shape[2])
        self.dt = float(dt)
        self._std, self._mean = self.kf.scale, self.kf.offset
        self.alpha = None
        self.fit = None
        self._cache = {}

    # ------------------------------------------------------------------ parameters / objective
    def _alpha(self, alpha):
        if alpha is None:
            if self.alpha is None:
                raise ValueError("no parameters: call solve() first or pass alpha [R,N+K]")
            return self.alpha
        return self.kf._dev(alpha, (self.R, self.N + self.K), "alpha")

    def get_mle(self, alpha=None):
        """``Metran.get_mle`` (metran.py:605-622) for every model: -2 log L ``[R]``."""
        phi, q = self.kf.params_from_alpha(self._alpha(alpha), dt=self.dt)
        return self.kf.loglik(phi, q)

    def solve(self, **kwargs):
        """``Metran.solve`` (metran.py:991-1045) for all models in lock-step (``calibrate_batch``);
        the optimum becomes the default parameter set of the accessors.  Returns the CalibrationResult.
        Keyword arguments go to ``calibrate_batch``; ``fd_below=4096`` (2048 for models of more than 16 states) lets the last
        stragglers finish on differenced gradients -- what the reference's own solver uses throughout -- and is what
        ``bench.py``'s calibration lines are measured with."""
        kwargs.setdefault("dt", self.dt)
        self.fit = calibrate_batch(self.kf, **kwargs)
        self.alpha = self.fit.alpha
        if (self.nfactors < self.K).any():
            # models with fewer factors than the batch maximum carry zero loading columns: their surplus cdf states are
            # decoupled dummies (alpha stays at its start value, zero gradient).  The reference counts N + nfactors[r]
            # parameters (solver.py:280: aic = 2 P + fun) and has no such rows in its parameter table.
            import torch

            nf = torch.as_tensor(self.nfactors, device=self.fit.obj.device)
            self.fit["aic"] = 2.0 * (self.N + nf).to(self.fit.obj.dtype) + self.fit.obj
            dummy = torch.arange(self.N + self.K, device=nf.device)[None, :] >= (self.N + nf)[:, None]
            self.fit["dummy_parameters"] = dummy          # [R, N+K] True where a parameter does not exist for the model
            if "stderr" in self.fit:
                self.fit["stderr"] = torch.where(dummy, torch.full_like(self.fit["stderr"], float("nan")), self.fit["stderr"])
        self._cache.clear()
        return self.fit

    def gather(self, t):
        """Per-model result ``[R_local, ...]`` of this rank -> ``[R_total, ...]`` on every rank, rank order."""
        from .distributed import gather_concat

        tail = tuple(t.shape[1:])
        return gather_concat(t.reshape(-1)).reshape((-1,) + tail)

    # ------------------------------------------------------------------ observations
    def mask_observations(self, mask):
        """``Metran.mask_observations`` (metran.py:464-494) for the batch: ``mask [R,T,N]`` non-zero = hide."""
        self.kf.mask_observations(mask)
        self._cache.clear()

    def unmask_observations(self):
        """``Metran.unmask_observations`` (metran.py:496-506)."""
        self.kf.unmask_observations()
        self._cache.clear()

    # ------------------------------------------------------------------ cached kernel runs (Metran._run_kalman)
    def _run(self, kind, alpha):
        """kind: "project" (filter + projecting smoother: sim_means/sim_vars in ORIGINAL units), "smoother" (filter +
        smoother with the state moments), "filter" (filter with the filtered moments).  One cached result per kind,
        valid for the parameter set it was computed with (metran.py:978-989 keeps one too)."""
        import torch

        from .kalmanfilter import check_status

        a = self._alpha(alpha)
        hit = self._cache.get(kind)
        if hit is not None and hit[0].shape == a.shape and bool(torch.equal(hit[0], a)):
            return hit[1]
        phi, q = self.kf.params_from_alpha(a, dt=self.dt)
        if kind == "project":
            self.kf.set_scaling(self._std, self._mean)
            out = self.kf.simulate_smoothed(phi, q)
        elif kind == "smoother":
            # state means + variances only (MK_OUT_VAR_ONLY): what the accessors below consume; the smoothed covariances
            # [R,T,n,n] are never written (wide models: the state tape, BatchedKalman.state_tape_path)
            out = self.kf.smooth_state_variances(phi, q)
        elif kind == "filter":
            out = self.kf.filter(phi, q, outputs=("F", "Pf"))
        else:
            raise ValueError(kind)
        check_status(out["status"], "MetranBatch(%s)" % kind)
        self._cache[kind] = (a.clone(), out)
        return out

    def _method(self, method):
        if method not in ("smoother", "filter"):
            raise ValueError("method must be 'smoother' or 'filter'")
        return method

    def _observation_matrix(self, standardized):
        """``get_observation_matrix`` / ``get_scaled_observation_matrix`` (metran.py:365-370, 944-961) ``[R,N,n]``."""
        import torch

        Z = torch.zeros((self.R, self.N, self.N + self.K), dtype=torch.float64, device=self.kf.device)
        Z[:, :, : self.N] = torch.eye(self.N, dtype=torch.float64, device=self.kf.device)
        Z[:, :, self.N:] = self.kf.loadings
        return Z if standardized else Z * self._std[:, :, None]

    # ------------------------------------------------------------------ simulation (projection of the states)
    def _simulate(self, alpha, standardized, method):
        if self._method(method) == "smoother":
            out = self._run("project", alpha)  # fused epilogue, original units
            means, variances = out["sim_means"], out["sim_vars"]
            if standardized:
                means = (means - self._mean[:, None, :]) / self._std[:, None, :]
                variances = variances / (self._std * self._std)[:, None, :]
            return means, variances
        out = self._run("filter", alpha)
        key = ("simf", bool(standardized))
        if key not in out:
            m, v = self.kf.simulate(self._observation_matrix(standardized), out["F"], out["Pf"])
            out[key] = (m if standardized else m + self._mean[:, None, :], v)
        return out[key]

    def get_simulated_means(self, alpha=None, standardized=False, method="smoother"):
        """``Metran.get_simulated_means`` (metran.py:758-795): tensor ``[R,T,N]`` (padding steps included)."""
        return self._simulate(alpha, standardized, method)[0]

    def get_simulated_variances(self, alpha=None, standardized=False, method="smoother"):
        """``Metran.get_simulated_variances`` (metran.py:797-829)."""
        return self._simulate(alpha, standardized, method)[1]

    def _series(self, r, name):
        names = list(self.batch.names[r])
        if name not in names:
            raise KeyError("Unknown name: " + str(name))  # the reference logs this and returns None (metran.py:881)
        return names.index(name)

    @staticmethod
    def _band(mean, variance, ci):
        from pandas import concat
        from scipy.stats import norm

        if ci is None:
            return mean
        if not (0 < ci < 1):
            raise Exception("The value of alpha must be between 0 and 1.")  # metran.py:741-744, 868-871
        iv = norm.ppf(1 - ci / 2.0) * np.sqrt(variance)
        out = concat([mean, mean - iv, mean + iv], axis=1)
        out.columns = ["mean", "lower", "upper"]
        return out

    def get_simulation(self, r, name, alpha=None, ci=0.05, standardized=False, method="smoother"):
        """``Metran.get_simulation`` (metran.py:831-883) for series ``name`` of model ``r``: DataFrame with
        ``mean`` (and ``lower``/``upper`` of the 1-ci interval; ``ci=None`` returns the mean Series)."""
        j = self._series(r, name)
        means, variances = self._simulate(alpha, standardized, method)
        L = int(self.batch.lengths[r])
        sim = self.batch.frame(r, means[r].cpu().numpy()).iloc[:L, j]
        return self._band(sim, self.batch.frame(r, variances[r].cpu().numpy()).iloc[:L, j] if ci is not None else None, ci)

    def decompose_simulation(self, r, name, alpha=None, standardized=False, method="smoother"):
        """``Metran.decompose_simulation`` (metran.py:885-942): DataFrame with the specific dynamic component
        ``sdf`` (carrying the series mean) and one column ``cdf<k>`` per common factor of series ``name``."""
        from pandas import DataFrame

        j = self._series(r, name)
        out = self._run(self._method(method), alpha)
        key = ("decomp", bool(standardized))
        if key not in out:
            states = out["S" if method == "smoother" else "F"]
            out[key] = self.kf.decompose(self._observation_matrix(standardized), states)
        sdf, cdf = out[key]
        L = int(self.batch.lengths[r])
        cols = {"sdf": sdf[r, :L, j].cpu().numpy() + (0.0 if standardized else float(self._mean[r, j]))}
        for k in range(int(self.nfactors[r])):
            cols["cdf%d" % (k + 1)] = cdf[r, k, :L, j].cpu().numpy()
        return DataFrame(cols, index=self.batch.index[r])

    # ------------------------------------------------------------------ states
    def _state_columns(self, r):
        return [str(n) + "_sdf" for n in self.batch.names[r]] + ["cdf%d" % (k + 1) for k in range(self.K)]

    def get_state_means(self, r, alpha=None, method="smoother"):
        """``Metran.get_state_means`` (metran.py:655-681) of model ``r``: DataFrame [T, N+K] with the reference's
        column names (``<series>_sdf`` ..., ``cdf1`` ...)."""
        from pandas import DataFrame

        out = self._run(self._method(method), alpha)
        L = int(self.batch.lengths[r])
        return DataFrame(out["S" if method == "smoother" else "F"][r, :L].cpu().numpy(), index=self.batch.index[r],
                         columns=self._state_columns(r))

    def get_state_variances(self, r, alpha=None, method="smoother"):
        """``Metran.get_state_variances`` (metran.py:683-711): the diagonals of the state covariances."""
        import torch
        from pandas import DataFrame

        out = self._run(self._method(method), alpha)
        L = int(self.batch.lengths[r])
        var = out["var"][r, :L] if method == "smoother" else torch.diagonal(out["Pf"][r, :L], dim1=1, dim2=2)
        return DataFrame(var.cpu().numpy(), index=self.batch.index[r], columns=self._state_columns(r))

    def get_state(self, r, i, alpha=None, ci=0.05, method="smoother"):
        """``Metran.get_state`` (metran.py:713-756): state ``i`` of model ``r`` with its 1-ci band."""
        if i < 0 or i >= self.N + self.K:
            raise IndexError("Value of i must be >=0 and <%d" % (self.N + self.K))  # the reference logs and returns None
        mean = self.get_state_means(r, alpha, method).iloc[:, i]
        return self._band(mean, self.get_state_variances(r, alpha, method).iloc[:, i] if ci is not None else None, ci)
