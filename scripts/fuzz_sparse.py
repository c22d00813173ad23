#!/usr/bin/env python
"""Fuzz of the single-record engine route (loglik_sparse_kernel<.., REC> + fill_gaps_kernel) against the step-by-step batched
filter on the same device: random shapes of the ahead-of-time list with n <= 16, random record lengths (up to six LDS tiles
of observed steps), gap patterns, parameter sets, initial moments, observation variances.  Prints the worst differences."""
import sys, os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
worst = {}
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    N, K = [(8, 2), (5, 1), (2, 1), (3, 1), (4, 1), (6, 2)][rng.integers(6)]
    T = int(rng.integers(1, 1600))
    keep = int(rng.choice([1, 1, 2, 3, 7, 18, 40]))
    d = make_dfm_batch(1, N, K, T, seed=int(rng.integers(1 << 30)), missing=float(rng.choice([0.0, 0.2, 0.6])), first_step="random")
    y = d["obs"][0].copy()
    mask = np.ones(T, bool)
    mask[int(rng.integers(keep))::keep] = False
    y[mask] = np.nan
    if rng.random() < 0.3:
        y[-int(rng.integers(1, 4)):] = np.nan
    n = N + K
    S = int(rng.integers(1, 17))
    phi = np.clip(d["phi"][0][None] * (1.0 + 0.05 * rng.standard_normal((S, n))), 0.0, 1.0 - 1e-9)
    if rng.random() < 0.3:
        phi[0, int(rng.integers(n))] = 0.0
    q = np.abs(d["q"][0][None] * (1.0 + 0.05 * rng.standard_normal((S, n)))) + 1e-12
    R = rng.uniform(0.0, 0.2, N) * (rng.random(N) < 0.4) if rng.random() < 0.5 else None
    x0 = rng.normal(size=(S, n)) if rng.random() < 0.5 else None
    P0 = None
    if rng.random() < 0.5:
        A = rng.normal(size=(S, n, n))
        P0 = A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n)
    kf = BatchedKalman(layout="time_major" if rng.random() < 0.5 else "model_major")
    kf.set_observations(y[None]).set_loadings(d["loadings"], None if R is None else R[None])
    a = kf.filter(phi, q, x0=x0, P0=P0)
    kf.set_variant("single_record", "stepwise")
    b = kf.filter(phi, q, x0=x0, P0=P0)
    for k in ("mle", "F", "Pf", "Xp", "Pp", "sigmas", "detfs"):
        va, vb = a[k].cpu().numpy(), b[k].cpu().numpy()
        sc = max(1.0, float(np.abs(vb).max()))
        e = float(np.abs(va - vb).max()) / sc
        if e > worst.get(k, (0.0,))[0]:
            worst[k] = (e, (N, K, T, keep, S))
    assert int(a["sigmacount"].cpu().numpy()[0]) == int(b["sigmacount"].cpu().numpy()[0])
    assert np.array_equal(a["status"].cpu().numpy(), b["status"].cpu().numpy())
    kf.close()
print({k: ("%.2e" % v[0], v[1]) for k, v in worst.items()})
