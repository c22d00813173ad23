"""Seeded synthetic dynamic-factor-model batches.

For model ``b`` of a batch with seed ``s`` the generator is
``np.random.default_rng([s, b])`` so a model's data does not depend on the batch
size or on which rank generates it (needed by the sharding-invariance tests).

* loadings ~ U(0.3, 0.6) / sqrt(K)  -> communality < 0.36
* alpha ~ U(5, 40) per state, phi = exp(-1/alpha), q as in ``params.phi_q_from_alpha``
* x_t = phi * x_{t-1} + sqrt(q) * eps_t,  x_{-1} = 0;  y_t = x_t[:N] + loadings @ x_t[N:]
* missing: each (t, j) dropped i.i.d. with probability ``missing``; ``first_step``
  forces step 0 fully observed ("observed"), fully missing ("empty") or leaves it
  random ("random").

The series are fed to the filter as they are (no re-standardisation), with
observation variance R = 0, x0 = 0, P0 = I.
"""
import numpy as np

from .params import phi_q_from_alpha

__all__ = ["make_dfm", "make_dfm_batch", "make_dfm_batch_torch"]


def make_dfm(N, K, T, seed, b=0, missing=0.0, first_step="observed"):
    rng = np.random.default_rng([int(seed), int(b)])
    n = N + K
    loadings = rng.uniform(0.3, 0.6, size=(N, K)) / np.sqrt(K)
    alpha = rng.uniform(5.0, 40.0, size=n)
    phi, q = phi_q_from_alpha(alpha, loadings)
    eps = rng.standard_normal((T, n)) * np.sqrt(q)
    x = np.zeros(n)
    y = np.empty((T, N))
    for t in range(T):
        x = phi * x + eps[t]
        y[t] = x[:N] + loadings @ x[N:]
    if missing > 0.0:
        drop = rng.random((T, N)) < missing
        if first_step == "observed":
            drop[0] = False
        elif first_step == "empty":
            drop[0] = True
        y[drop] = np.nan
    elif first_step == "empty":
        y[0] = np.nan
    return y, alpha, loadings, phi, q


def make_dfm_batch(B, N, K, T, seed, missing=0.0, first_step="observed", start=0):
    """Models ``start .. start+B-1`` of the batch with the given seed.

    Returns dict with obs [B,T,N] (NaN = missing), alpha/phi/q [B,N+K], loadings [B,N,K].
    """
    n = N + K
    out = dict(
        obs=np.empty((B, T, N)),
        alpha=np.empty((B, n)),
        loadings=np.empty((B, N, K)),
        phi=np.empty((B, n)),
        q=np.empty((B, n)),
    )
    for i in range(B):
        y, a, g, p, qq = make_dfm(N, K, T, seed, start + i, missing, first_step)
        out["obs"][i], out["alpha"][i], out["loadings"][i] = y, a, g
        out["phi"][i], out["q"][i] = p, qq
    return out


def make_dfm_batch_torch(B, N, K, T, seed, device, missing=0.0):
    """Device-side generator for large benchmark batches (same distributions, torch RNG;
    not reproducible against ``make_dfm_batch``).  Returns float64 torch tensors."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    n = N + K
    f64 = dict(dtype=torch.float64, device=device)
    loadings = (0.3 + 0.3 * torch.rand((B, N, K), generator=g, **f64)) / float(np.sqrt(K))
    alpha = 5.0 + 35.0 * torch.rand((B, n), generator=g, **f64)
    phi = torch.exp(-1.0 / alpha)
    q = 1.0 - phi * phi
    q[:, :N] = q[:, :N] * (1.0 - (loadings * loadings).sum(-1))
    sq = torch.sqrt(q)
    x = torch.zeros((B, n), **f64)
    obs = torch.empty((B, T, N), **f64)
    for t in range(T):
        x = phi * x + sq * torch.randn((B, n), generator=g, **f64)
        obs[:, t, :] = x[:, :N] + torch.einsum("bnk,bk->bn", loadings, x[:, N:])
    if missing > 0.0:
        drop = torch.rand((B, T, N), generator=g, device=device) < missing
        drop[:, 0, :] = False
        obs[drop] = float("nan")
    return dict(obs=obs, alpha=alpha, loadings=loadings, phi=phi, q=q)
