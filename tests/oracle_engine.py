"""Test infrastructure: the slice of ``metran_amd.engine.BatchedKalman`` that the HOST-side drivers touch
(``calibrate_batch``, ``HipSolve`` / ``BatchObjective``), on the CPU: values from the C oracle (``oracle_dfm_batch``),
gradients from the numpy adjoint restatement (tests/adjoint_ref.py).  It lets the CPU tier run those drivers unchanged
(tests/test_calibrate_host.py, tests/test_solver_host.py); the product never sees it -- ``BatchedKalman`` has no CPU path."""
import numpy as np
import torch

import adjoint_ref
import oracle
from metran_amd.params import phi_q_from_alpha


class OracleEngine:
    """The slice of ``BatchedKalman`` that ``calibrate_batch`` touches, on the CPU, values from the C oracle
    (``oracle_dfm_batch``) and gradients from the numpy adjoint restatement (tests/adjoint_ref.py)."""

    def __init__(self, obs=None, loadings=None, adjoint=True, log=None):
        self.device = torch.device("cpu")
        if obs is not None:
            self.set_observations(obs).set_loadings(loadings)
        self._adjoint = adjoint
        self._pending = None
        self.log = log if log is not None else []       # (what, instances) per launch, shared with the subsets

    n = property(lambda self: self.N + self.K)

    def set_observations(self, obs):
        self.obs_np = np.asarray(obs, float)
        self.R, self.T, self.N = self.obs_np.shape
        return self

    def set_loadings(self, loadings, obsvar=None):
        assert obsvar is None
        self.load_np = np.asarray(loadings, float)
        self.K = self.load_np.shape[2]
        return self

    def _dev(self, a, shape=None, name="array"):
        a = torch.as_tensor(np.asarray(a, float) if not isinstance(a, torch.Tensor) else a, dtype=torch.float64)
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError("%s must be %s" % (name, tuple(shape)))
        return a

    def has_adjoint(self):
        return self._adjoint

    def record_stride(self):
        return self.n * (self.n + 1)

    def subset(self, index):
        idx = np.asarray(index)
        return OracleEngine(self.obs_np[idx], self.load_np[idx], self._adjoint, self.log)

    def _records(self, B):
        assert B % self.R == 0
        return np.arange(B) % self.R                    # instance s*R + r reads record r

    def params_from_alpha(self, alpha, dt=1.0):
        a = self._dev(alpha).numpy()
        phi, q = phi_q_from_alpha(a, self.load_np[self._records(a.shape[0])], dt)
        return torch.from_numpy(phi.copy()), torch.from_numpy(q.copy())

    def loglik(self, phi, q, warmup=1):
        phi, q = self._dev(phi), self._dev(q)
        rec = self._records(phi.shape[0])
        self.log.append(("loglik", int(phi.shape[0])))
        res = oracle.dfm_batch(self.obs_np[rec], phi.numpy(), q.numpy(), self.load_np[rec], warmup=warmup, smooth=False,
                               outputs="mle")
        return torch.from_numpy(res["mle"].copy())

    def _grad(self, alpha, dt, warmup):
        a = self._dev(alpha).numpy()
        rec = self._records(a.shape[0])
        f, g = np.empty(a.shape[0]), np.empty_like(a)
        for b, r in enumerate(rec):
            G = self.load_np[r]
            phi, q = phi_q_from_alpha(a[b], G, dt)
            f[b], gphi, gq = adjoint_ref.gradient(self.obs_np[r], phi, q, G, warmup=warmup)
            scale = np.concatenate([1.0 - np.sum(G * G, axis=1), np.ones(self.K)])
            dphi = phi * dt / a[b] ** 2                  # phi = exp(-dt / alpha), q = (1 - phi^2) * scale
            g[b] = (gphi - 2.0 * phi * scale * gq) * dphi
        return torch.from_numpy(f), torch.from_numpy(g)

    def loglik_grad_alpha(self, alpha, dt=1.0, warmup=1):
        alpha = self._dev(alpha)
        self.log.append(("forward+backward", int(alpha.shape[0])))
        return self._grad(alpha, dt, warmup)

    def loglik_forward_alpha(self, alpha, dt=1.0, warmup=1):
        alpha = self._dev(alpha)
        self.log.append(("forward", int(alpha.shape[0])))
        phi, q = self.params_from_alpha(alpha, dt)
        self._pending = (alpha.clone(), dt, warmup)
        rec = self._records(alpha.shape[0])
        res = oracle.dfm_batch(self.obs_np[rec], phi.numpy(), q.numpy(), self.load_np[rec], warmup=warmup, smooth=False,
                               outputs="mle")
        return torch.from_numpy(res["mle"].copy())

    def loglik_backward_alpha(self):
        assert self._pending is not None, "backward without a forward pass"
        alpha, dt, warmup = self._pending
        self._pending = None                             # one backward pass per forward pass, like the engine
        self.log.append(("backward", int(alpha.shape[0])))
        return self._grad(alpha, dt, warmup)[1]
