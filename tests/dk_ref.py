"""numpy restatement of the TAPE path (test infrastructure): the forward filter of the wide models writing, per
(step, series), one entry of the backward tape, and the inverse-free backward recursion (Durbin-Koopman r / N form)
that turns the tape into the smoothed projection.  What it restates is metran_amd/csrc/mk_split.hip (OUT = 4) and
metran_amd/csrc/mk_dk.hip; what it is checked against is the oracle (reference kalmansmoother + simulate,
/root/reference/metran/kalmanfilter.py:403-476, 569-603) in tests/test_dk_ref.py.

Observable basis.  Metran's state is x = [sdf_1..sdf_N | cdf_1..cdf_K] and series j observes z_j x = x_j + sum_k g_jk x_{N+k}
(metran.py:365-370).  With T = [[I, G], [0, I]] the state xt = T x = [y_1..y_N | cdf] has observation rows e_j, so a scalar
update touches ONE row / column of the backward information matrix; the price is a transition Pht = T Phi T^-1 =
[[Phi_s, G Phi_f - Phi_s G], [0, Phi_f]] that is no longer diagonal (paid once per step, not once per observation).

Tape entry (ES = n + 4 doubles) of series j at step t, written by the FILTER:
  observed    [ kt (n) | v/f | 1/f | y_j | 0 ]    kt = T k, k = P z_j^T / f the gain of that scalar update (kalmanfilter.py:349-366)
  unobserved  [ pt (n) | yhat_u | Ptt_uu | NaN | 0 ]   pt = T Pf z_u^T, yhat_u = z_u x_f, Ptt_uu = z_u Pf z_u^T  (END of the step)
Backward, per step (r, N in the observable basis; both zero behind the last step):
  unobserved u:  mean = yhat_u + pt.r,  var = Ptt_uu - pt' N pt                      (x_s = x_f + Pf r, V = Pf - Pf N Pf)
  observed j, descending:  w = N kt, beta = kt.r, alpha = kt.w;
        r_j += v/f - beta;   N[:, j] = N[j, :] = N[:, j] - w;   N[j, j] = (old) - 2 w_j + alpha + 1/f
        (r <- L'r + z v/f, N <- L'NL + z z'/f with L = I - kt e_j'; for R = 0: mean = y_j, var = 0)
  transition:  r <- Pht' r,  N <- Pht' N Pht
"""
import numpy as np


def entry_stride(N, K):
    return N + K + 4


def unpack_block(block, N, K):
    """Device layout of one (model, step) tape block (mk_internal.h: entry j = series part at j*XS, side row at SO + j*SS,
    side row = [ factor part (K) | s0 | s1 | s2 | 0 ]) to entries [N, n+4]."""
    SW = K + 4
    XS = SS = N + SW
    SO = N
    block = np.asarray(block).reshape(-1)
    X = np.stack([block[j * XS:j * XS + N] for j in range(N)])
    S = np.stack([block[SO + j * SS:SO + j * SS + SW] for j in range(N)])
    return np.concatenate([X, S], axis=1)


def transform(loadings):
    N, K = loadings.shape
    T = np.eye(N + K)
    T[:N, N:] = loadings
    return T


def filter_tape(obs, phi, q, loadings, obsvar=None, x0=None, P0=None, state=False):
    """One model.  obs [T,N] (NaN = missing).  Returns tape [T, N, n+4]; with ``state`` (the STATE tape, MK_OUT_TAPE |
    MK_OUT_VAR_ONLY) [T, N + K, n+4]: K more entries per step, the factor columns of the filtered covariance in the observable
    basis,  entry N+k = [ T Pf e_{N+k} (n) | x_f[N+k] | Pf[N+k][N+k] | NaN | 0 ]."""
    Tn, N = obs.shape
    K = loadings.shape[1]
    n = N + K
    Z = np.concatenate([np.eye(N), loadings], axis=1)
    Tm = transform(loadings)
    R = np.zeros(N) if obsvar is None else np.asarray(obsvar, float)
    x = np.zeros(n) if x0 is None else np.array(x0, float)
    P = np.eye(n) if P0 is None else np.array(P0, float)
    tape = np.zeros((Tn, N + K if state else N, n + 4))
    for t in range(Tn):
        x = phi * x
        P = P * np.outer(phi, phi) + np.diag(q)
        seen = np.isfinite(obs[t])
        for j in np.nonzero(seen)[0]:
            z = Z[j]
            v = obs[t, j] - z @ x
            d = P @ z
            f = z @ d + R[j]
            k = d / f
            x = x + k * v
            P = P - np.outer(k, k) * f
            e = tape[t, j]
            e[:n] = Tm @ k
            e[n:] = (v / f, 1.0 / f, obs[t, j], 0.0)
        for u in np.nonzero(~seen)[0]:
            z = Z[u]
            d = P @ z
            e = tape[t, u]
            e[:n] = Tm @ d
            e[n:] = (z @ x, z @ d, np.nan, 0.0)
        if state:
            for k in range(K):
                e = tape[t, N + k]
                e[:n] = Tm @ P[:, N + k]
                e[n:] = (x[N + k], P[N + k, N + k], np.nan, 0.0)
    return tape


def transition(phi, loadings):
    """Pht = T Phi T^-1 = [[Phi_s, C], [0, Phi_f]], C[a][k] = g_ak (phi_{N+k} - phi_a)."""
    N, K = loadings.shape
    n = N + K
    Pht = np.diag(phi).astype(float)
    Pht[:N, N:] = loadings * (phi[None, N:] - phi[:N, None])
    assert Pht.shape == (n, n)
    return Pht


def dk_smooth(tape, phi, loadings, obsvar=None):
    """One model.  Returns (means [T,N], variances [T,N]) of the smoothed observables z_j x_t (unscaled).
    ``obsvar`` R_j != 0: right after its update the filter's moments of an observed series are z x = y - v R/f and
    P z' = k R, so  mean = y - R (v/f - beta),  var = R (1 - R/f) - R^2 alpha  with the update's own beta and alpha."""
    Tn, N, ES = tape.shape
    n = ES - 4
    R = np.zeros(N) if obsvar is None else np.asarray(obsvar, float)
    Pht = transition(phi, loadings)
    r = np.zeros(n)
    Nm = np.zeros((n, n))
    means, variances = np.empty((Tn, N)), np.empty((Tn, N))
    for t in range(Tn - 1, -1, -1):
        seen = ~np.isnan(tape[t, :, n + 2])
        for u in np.nonzero(~seen)[0]:
            e = tape[t, u]
            p = e[:n]
            means[t, u] = e[n] + p @ r
            variances[t, u] = e[n + 1] - p @ Nm @ p
        for j in np.nonzero(seen)[0][::-1]:
            e = tape[t, j]
            k = e[:n]
            w = Nm @ k
            beta = k @ r
            alpha = k @ w
            r[j] += e[n] - beta
            col = Nm[:, j] - w
            col[j] = Nm[j, j] - 2.0 * w[j] + alpha + e[n + 1]
            Nm[:, j] = col
            Nm[j, :] = col
            means[t, j] = e[n + 2] - R[j] * (e[n] - beta)
            variances[t, j] = R[j] * (1.0 - R[j] * e[n + 1]) - R[j] * R[j] * alpha
        r = Pht.T @ r
        Nm = Pht.T @ Nm @ Pht
    return means, variances


def dk_smooth_state(tape, phi, loadings):
    """The STATE outputs from the state tape (filter_tape(..., state=True); R = 0): smoothed state means [T,n] and
    variances [T,n] in Metran's own basis, next to the projected (means [T,N], variances [T,N]) of dk_smooth -- what
    smoother_dk_kernel<N,K,false,true> computes (mk_dk.hip).  At the END of a step, with (r, N) before the step's updates,
        xt_s = xt_f + Pt r,   Vt = Pt - Pt N Pt          (observable basis, Pt = T Pf T')
    and Pt has zero rows / columns at the series observed at the step (R = 0: the observable IS the observation).  What the
    state variances need of Vt: its diagonal over the unobserved series (dk_smooth's variances), the K factor columns
    Vt[., N+k] = Pt[., N+k] - Pt N Pt[., N+k] -- K more products w_k = N pt_{N+k} and then, for every entry a of the step
    (unobserved series and factors), the K dot products pt_a . w_k -- and nothing else:
        x_a = xt_a - g_a . xt_F,   V_aa = Vt_aa - 2 g_a . Vt[a, F] + g_a' Vt_FF g_a,   x_{N+k} = xt_{N+k},  V = Vt_FF[k][k]."""
    Tn, NE, ES = tape.shape
    n = ES - 4
    N, K = loadings.shape
    assert NE == N + K
    Pht = transition(phi, loadings)
    r = np.zeros(n)
    Nm = np.zeros((n, n))
    S, var = np.empty((Tn, n)), np.empty((Tn, n))
    means, variances = np.empty((Tn, N)), np.empty((Tn, N))
    for t in range(Tn - 1, -1, -1):
        seen = ~np.isnan(tape[t, :N, n + 2])
        W = np.stack([Nm @ tape[t, N + k, :n] for k in range(K)], axis=1)          # [n, K]  w_k = N pt_{N+k}
        xf = np.array([tape[t, N + k, n] + tape[t, N + k, :n] @ r for k in range(K)])   # smoothed factor means
        Vff = np.array([[tape[t, N + k, N + l] - tape[t, N + k, :n] @ W[:, l] for l in range(K)] for k in range(K)])
        Vff = 0.5 * (Vff + Vff.T)
        for a in range(N):
            g = loadings[a]
            e = tape[t, a]
            if seen[a]:
                xt, vaa, vaf = e[n + 2], 0.0, np.zeros(K)
            else:
                p = e[:n]
                xt = e[n] + p @ r
                vaa = e[n + 1] - p @ Nm @ p
                vaf = e[N:n] - p @ W                                               # Vt[a, N+k]
            means[t, a], variances[t, a] = xt, vaa
            S[t, a] = xt - g @ xf
            var[t, a] = vaa - 2.0 * g @ vaf + g @ Vff @ g
        S[t, N:] = xf
        var[t, N:] = np.diag(Vff)
        for j in np.nonzero(seen)[0][::-1]:
            e = tape[t, j]
            k = e[:n]
            w = Nm @ k
            beta, alpha = k @ r, k @ w
            r[j] += e[n] - beta
            col = Nm[:, j] - w
            col[j] = Nm[j, j] - 2.0 * w[j] + alpha + e[n + 1]
            Nm[:, j] = col
            Nm[j, :] = col
        r = Pht.T @ r
        Nm = Pht.T @ Nm @ Pht
    return S, var, means, variances


def dk_state_moments(obs, phi, q, loadings):
    """Groundwork for the state outputs (``filter_smooth`` / ``MK_OUT_VAR_ONLY`` of wide models, today the RTS kernel): the
    smoothed STATE means and covariances from the same backward recursion, ``x_s = x_f + Pf r``, ``V = Pf - Pf N Pf`` with the
    filtered moments of the step and (r, N) before the step's updates, taken back to Metran's own basis (r = T'rt, N = T'Nt T).
    One model; returns (S [T,n], Ps [T,n,n]).  Checked against the oracle in tests/test_dk_tape.py."""
    Tn, N = obs.shape
    K = loadings.shape[1]
    n = N + K
    Z = np.concatenate([np.eye(N), loadings], axis=1)
    Tm = transform(loadings)
    x, P = np.zeros(n), np.eye(n)
    F, Pf = np.empty((Tn, n)), np.empty((Tn, n, n))
    for t in range(Tn):  # the filter once more, keeping the filtered moments (kalmanfilter.py:318-390)
        x = phi * x
        P = P * np.outer(phi, phi) + np.diag(q)
        for j in np.nonzero(np.isfinite(obs[t]))[0]:
            z = Z[j]
            d = P @ z
            f = z @ d
            k = d / f
            x = x + k * (obs[t, j] - z @ x)
            P = P - np.outer(k, k) * f
        F[t], Pf[t] = x, P
    tape = filter_tape(obs, phi, q, loadings)
    Pht = transition(phi, loadings)
    r, Nm = np.zeros(n), np.zeros((n, n))
    S, Ps = np.empty((Tn, n)), np.empty((Tn, n, n))
    for t in range(Tn - 1, -1, -1):
        ro, No = Tm.T @ r, Tm.T @ Nm @ Tm
        S[t] = F[t] + Pf[t] @ ro
        Ps[t] = Pf[t] - Pf[t] @ No @ Pf[t]
        seen = ~np.isnan(tape[t, :, n + 2])
        for j in np.nonzero(seen)[0][::-1]:
            e = tape[t, j]
            k = e[:n]
            w = Nm @ k
            beta, alpha = k @ r, k @ w
            r[j] += e[n] - beta
            col = Nm[:, j] - w
            col[j] = Nm[j, j] - 2.0 * w[j] + alpha + e[n + 1]
            Nm[:, j] = col
            Nm[j, :] = col
        r = Pht.T @ r
        Nm = Pht.T @ Nm @ Pht
    return S, Ps


def project(means, variances, scale=None, offset=None):
    """simulate() with the series' standard deviations and means folded in (metran.py:944-961, kalmanfilter.py:597-602)."""
    if scale is not None:
        means = means * scale
        variances = variances * scale * scale
    if offset is not None:
        means = means + offset
    return means, np.maximum(variances, 0.0)
