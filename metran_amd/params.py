"""Parameter -> state-space map of Metran's dynamic-factor model, batched.

Restates (vectorised over a leading batch axis, no pandas look-ups):

* ``Metran._phi``                      /root/reference/metran/metran.py:246-263
* ``Metran.get_transition_matrix``     metran/metran.py:265-290   (diagonal, returned as a vector)
* ``Metran.get_transition_covariance`` metran/metran.py:292-322   (diagonal, returned as a vector)
* ``Metran.get_observation_matrix``    metran/metran.py:347-370   (``Z = [I_N | loadings]``)
* ``Metran.get_observation_variance``  metran/metran.py:372-384   (zeros)

Parameter order follows ``Metran.parameters`` after ``solve()``: the N specific
(sdf) alphas first, then the K common (cdf) alphas.
"""
import numpy as np

__all__ = ["phi_from_alpha", "phi_q_from_alpha", "observation_matrix", "dt_days"]


def dt_days(freq="D"):
    """``Timedelta(to_offset(freq)) / Timedelta(1, "D")`` (metran/metran.py:262)."""
    from pandas import Timedelta
    from pandas.tseries.frequencies import to_offset

    return Timedelta(to_offset(freq)) / Timedelta(1, "D")


def phi_from_alpha(alpha, dt=1.0):
    """``phi = exp(-dt / alpha)``  (metran/metran.py:262-263)."""
    return np.exp(-float(dt) / np.asarray(alpha, dtype=np.float64))


def phi_q_from_alpha(alpha, loadings, dt=1.0):
    """Diagonals of the transition matrix and transition covariance.

    Parameters
    ----------
    alpha : array [..., N+K]   sdf alphas then cdf alphas
    loadings : array [..., N, K] factor loadings (``Metran.factors``)
    dt : float  time step in days

    Returns
    -------
    phi, q : arrays [..., N+K]
        ``q_i = (1 - phi_i**2) * (1 - sum_k loadings[i,k]**2)`` for i < N
        (metran/metran.py:311-316) and ``1 - phi_i**2`` for the common factors
        (metran/metran.py:317-321).
    """
    alpha = np.asarray(alpha, dtype=np.float64)
    loadings = np.asarray(loadings, dtype=np.float64)
    N, K = loadings.shape[-2], loadings.shape[-1]
    if alpha.shape[-1] != N + K:
        raise ValueError("alpha must have N+K=%d entries, got %d" % (N + K, alpha.shape[-1]))
    phi = phi_from_alpha(alpha, dt)
    communality = np.sum(np.square(loadings), axis=-1)  # metran/metran.py:311
    q = 1.0 - phi ** 2
    q = np.array(np.broadcast_to(q, np.broadcast_shapes(q.shape, communality.shape[:-1] + (N + K,))))
    q[..., :N] = q[..., :N] * (1.0 - communality)
    phi = np.broadcast_to(phi, q.shape)
    return np.ascontiguousarray(phi), np.ascontiguousarray(q)


def observation_matrix(loadings):
    """``Z = [I_N | loadings]`` (metran/metran.py:365-370), batched."""
    loadings = np.asarray(loadings, dtype=np.float64)
    N, K = loadings.shape[-2], loadings.shape[-1]
    Z = np.zeros(loadings.shape[:-2] + (N, N + K))
    Z[..., :, :N] = np.eye(N)
    Z[..., :, N:] = loadings
    return Z
