"""Many Metran models at once: the accessors a user of ``metran.Metran`` calls after construction
(``solve``, ``get_mle``, ``get_simulated_means/variances``, ``get_simulation``, ``get_state_means``,
``decompose_simulation``; /root/reference/metran/metran.py:605-989, 991-1045) on top of the batched
engine.  The orchestration is Python, as in the reference; every number comes from the HIP kernels:

    ingest.ObservationBatch  ->  mk_standardize            (Metran.__init__: combine, daily grid, standardise)
    calibrate_batch          ->  mk_loglik_grad / mk_loglik (Metran.solve for all models in lock-step)
    simulate_smoothed        ->  filter + projecting smoother (get_simulated_means / _variances)
    filter_smooth            ->  state means / decomposition

What the reference does per model and this class does not: the factor analysis that produces the loadings
(``metran/factoranalysis.py``) -- pass ``factors`` ``[R,N,K]`` (e.g. from the
reference's ``FactorAnalysis().solve`` per model, or a shared loading matrix).
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
    factors : array ``[R,N,K]`` or ``[N,K]`` (shared)   factor loadings, ``Metran.factors``
    device, tmin, tmax, min_pairs : as in ``BatchedKalman`` / ``Metran.settings``
    """

    def __init__(self, models, factors, device=None, tmin=None, tmax=None, min_pairs=20, dt=1.0):
        self.batch = ObservationBatch(models, tmin=tmin, tmax=tmax, min_pairs=min_pairs)
        self.kf = BatchedKalman(device, layout="time_major")
        self.batch.upload(self.kf)  # standardised records + (std, mean) scaling on the device
        R, T, N = self.batch.shape
        factors = np.asarray(factors, dtype=np.float64)
        if factors.ndim == 2:
            factors = np.broadcast_to(factors, (R,) + factors.shape).copy()
        self.kf.set_loadings(factors)
        self.factors = factors
        self.R, self.T, self.N, self.K = R, T, N, int(factors.shape[2])
        self.dt = float(dt)
        self._std, self._mean = self.kf.scale, self.kf.offset
        self.alpha = None
        self.fit = None

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
        the optimum becomes the default parameter set of the accessors.  Returns the CalibrationResult."""
        kwargs.setdefault("dt", self.dt)
        self.fit = calibrate_batch(self.kf, **kwargs)
        self.alpha = self.fit.alpha
        return self.fit

    # ------------------------------------------------------------------ simulation (projection of the states)
    def _scaling(self, standardized):
        if standardized:
            self.kf.set_scaling(None, None)
        else:
            self.kf.set_scaling(self._std, self._mean)

    def _simulate(self, alpha, standardized, method):
        phi, q = self.kf.params_from_alpha(self._alpha(alpha), dt=self.dt)
        self._scaling(standardized)
        if method == "smoother":
            out = self.kf.simulate_smoothed(phi, q)
            means, variances = out["sim_means"], out["sim_vars"]
        elif method == "filter":
            import torch

            out = self.kf.filter(phi, q, outputs=("F", "Pf"))
            Z = torch.zeros((self.R, self.N, self.N + self.K), dtype=torch.float64, device=self.kf.device)
            Z[:, :, : self.N] = torch.eye(self.N, dtype=torch.float64, device=self.kf.device)
            Z[:, :, self.N:] = self.kf.loadings
            if not standardized:  # get_scaled_observation_matrix (metran.py:944-961)
                Z = Z * self._std[:, :, None]
            means, variances = self.kf.simulate(Z, out["F"], out["Pf"])
            if not standardized:
                means = means + self._mean[:, None, :]
        else:
            raise ValueError("method must be 'smoother' or 'filter'")
        self.kf.set_scaling(self._std, self._mean)
        return means, variances

    def get_simulated_means(self, alpha=None, standardized=False, method="smoother"):
        """``Metran.get_simulated_means`` (metran.py:758-795): tensor ``[R,T,N]`` (padding steps included)."""
        return self._simulate(alpha, standardized, method)[0]

    def get_simulated_variances(self, alpha=None, standardized=False, method="smoother"):
        """``Metran.get_simulated_variances`` (metran.py:797-829)."""
        return self._simulate(alpha, standardized, method)[1]

    def get_simulation(self, r, name, alpha=None, ci=0.05, standardized=False, method="smoother"):
        """``Metran.get_simulation`` (metran.py:831-883) for series ``name`` of model ``r``: DataFrame with
        ``mean`` (and ``lower``/``upper`` of the 1-ci interval; ``ci=None`` returns the mean Series)."""
        from pandas import concat
        from scipy.stats import norm

        names = list(self.batch.names[r])
        if name not in names:
            raise KeyError("Unknown name: " + str(name))
        j = names.index(name)
        means, variances = self._simulate(alpha, standardized, method)
        L = int(self.batch.lengths[r])
        sim = self.batch.frame(r, means[r].cpu().numpy()).iloc[:L, j]
        if ci is None:
            return sim
        if not (0 < ci < 1):
            raise Exception("The value of alpha must be between 0 and 1.")  # metran.py:868-871
        z = norm.ppf(1 - ci / 2.0)
        iv = z * np.sqrt(self.batch.frame(r, variances[r].cpu().numpy()).iloc[:L, j])
        out = concat([sim, sim - iv, sim + iv], axis=1)
        out.columns = ["mean", "lower", "upper"]
        return out

    # ------------------------------------------------------------------ states
    def get_state_means(self, r, alpha=None, method="smoother"):
        """``Metran.get_state_means`` (metran.py:655-688) of model ``r``: DataFrame [T, N+K] with the
        reference's column names (``<series>_sdf`` ..., ``cdf1`` ...)."""
        from pandas import DataFrame

        phi, q = self.kf.params_from_alpha(self._alpha(alpha), dt=self.dt)
        key = "S" if method == "smoother" else "F"
        out = (self.kf.filter_smooth if method == "smoother" else self.kf.filter)(phi, q, outputs=(key,) if key == "F" else ("F", "Pf", "S"))
        L = int(self.batch.lengths[r])
        cols = [str(n) + "_sdf" for n in self.batch.names[r]] + ["cdf%d" % (k + 1) for k in range(self.K)]
        return DataFrame(out[key][r, :L].cpu().numpy(), index=self.batch.index[r], columns=cols)
