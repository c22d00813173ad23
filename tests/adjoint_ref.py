"""numpy restatement (test infrastructure) of the adjoint gradient of Metran's objective: the forward
recursion is the reference's sequential-processing filter (metran/kalmanfilter.py:236-400, get_mle
:550-567 incl. the compressed warm-up index); the backward pass is the reverse-mode derivative written
out in mk_kernels.hip::adjoint_kernel.  Checked against central differences in tests/test_adjoint.py."""
import numpy as np


def forward(y, phi, q, G, warmup=1, x0=None, P0=None, R=None):
    T, N = y.shape
    n = N + G.shape[1]
    x = np.zeros(n) if x0 is None else np.array(x0, float)
    P = np.eye(n) if P0 is None else np.array(P0, float)
    R = np.zeros(N) if R is None else R
    F, Pf = np.zeros((T, n)), np.zeros((T, n, n))
    mle, sc, nobs = 0.0, 0, 0
    for t in range(T):
        x = phi * x
        P = np.outer(phi, phi) * P + np.diag(q)
        obs = [j for j in range(N) if np.isfinite(y[t, j])]
        sig = det = 0.0
        for j in obs:
            z = np.zeros(n)
            z[j] = 1.0
            z[N:] = G[j]
            v = y[t, j] - z @ x
            d = P @ z
            f = z @ d + R[j]
            x = x + d * v / f
            P = P - np.outer(d, d) / f
            sig += v * v / f
            det += np.log(f)
        if obs:
            if sc >= warmup:
                mle += sig + det
            sc += 1
        if t >= warmup:
            nobs += len(obs)
        F[t], Pf[t] = x, P
    return mle + nobs * np.log(2 * np.pi), F, Pf, sc


def gradient(y, phi, q, G, warmup=1, x0=None, P0=None, R=None):
    """-> (mle, d mle/d phi, d mle/d q)"""
    mle, F, Pf, sctot = forward(y, phi, q, G, warmup, x0, P0, R)
    R = np.zeros(y.shape[1]) if R is None else R
    T, N = y.shape
    n = N + G.shape[1]
    xb, Pb = np.zeros(n), np.zeros((n, n))
    gphi, gq = np.zeros(n), np.zeros(n)
    rem = 0
    for t in range(T - 1, -1, -1):
        xprev = F[t - 1] if t > 0 else (np.zeros(n) if x0 is None else np.array(x0, float))
        Pprev = Pf[t - 1] if t > 0 else (np.eye(n) if P0 is None else np.array(P0, float))
        x = phi * xprev
        P = np.outer(phi, phi) * Pprev + np.diag(q)
        obs = [j for j in range(N) if np.isfinite(y[t, j])]
        if obs:
            w = 1.0 if sctot - rem - 1 >= warmup else 0.0
            rem += 1
            st = []
            for j in obs:
                z = np.zeros(n)
                z[j] = 1.0
                z[N:] = G[j]
                v = y[t, j] - z @ x
                d = P @ z
                f = z @ d + R[j]
                st.append((z, v, d, f))
                x = x + d * v / f
                P = P - np.outer(d, d) / f
            for z, v, d, f in reversed(st):
                rf = 1.0 / f
                a, b = xb @ d, Pb @ d
                c = d @ b
                vbar = (w * 2 * v + a) * rf
                fbar = (w * (1 - v * v * rf) - a * v * rf + c * rf) * rf
                dbar = xb * v * rf - 2 * b * rf + fbar * z
                xb = xb - vbar * z
                Pb = Pb + 0.5 * (np.outer(dbar, z) + np.outer(z, dbar))
        gq += np.diag(Pb)
        gphi += xb * xprev + 2 * (Pb * Pprev) @ phi
        Pb = np.outer(phi, phi) * Pb
        xb = phi * xb
    return mle, gphi, gq
