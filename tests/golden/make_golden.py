#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the *reference itself*.

Run ONCE in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

The reference package is imported unmodified through ``_refshim`` (pastas stub,
un-jitted numba source).  For every case both reference engines are run --
``seqkalmanfilter`` (the numba source, metran/kalmanfilter.py:236-400) and
``seqkalmanfilter_np`` (metran/kalmanfilter.py:122-233) -- followed by
``kalmansmoother`` (metran/kalmanfilter.py:403-476), ``SPKalmanFilter.get_mle``
(:550-567), ``simulate`` (:569-603) and ``decompose`` (:605-644).

The fixtures are what pins ``oracle/`` (and through it the HIP kernels) to the
reference; the GPU box has no /root/reference, so nothing at test time reads it.
"""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refshim  # noqa: E402

metran = _refshim.install()
import metran.kalmanfilter as kfm  # noqa: E402

from metran_amd.params import observation_matrix, phi_q_from_alpha  # noqa: E402
from metran_amd.synthetic import make_dfm  # noqa: E402
from analytic_objective import solver_objective  # noqa: E402


def run_reference(obs, phi, q, loadings, r=None, warmup=1, engine="numba-source"):
    """obs [T,N] NaN=missing -> dict of every array the reference path produces."""
    T, N = obs.shape
    n = phi.shape[0]
    kf = kfm.SPKalmanFilter(engine="numpy")
    kf.filtermethod = kfm.seqkalmanfilter if engine == "numba-source" else kfm.seqkalmanfilter_np
    kf.set_observations(pd.DataFrame(obs))
    Z = observation_matrix(loadings)
    R = np.zeros(N) if r is None else np.asarray(r, float)
    kf.set_matrices(np.diag(phi), np.diag(q), Z, R)
    kf.run_smoother()
    out = dict(
        sigmas=np.asarray(kf.sigmas, float),
        detfs=np.asarray(kf.detfs, float),
        sigmacount=np.int64(len(kf.sigmas)),
        count=kf.observation_count.copy(),
        mle=np.float64(kf.get_mle(warmup=warmup)),
        F=kf.filtered_state_means.copy(),
        Pf=kf.filtered_state_covariances.copy(),
        Xp=kf.predicted_state_means.copy(),
        Pp=kf.predicted_state_covariances.copy(),
        S=kf.smoothed_state_means.copy(),
        Ps=kf.smoothed_state_covariances.copy(),
    )
    return kf, out


def synthetic_case(fname, N, K, T, seed, models, missing=0.0, first_steps=None, cov_every=1,
                   extra_nan=None):
    data = {}
    for i, b in enumerate(models):
        fs = "observed" if first_steps is None else first_steps[i]
        y, alpha, loadings, phi, q = make_dfm(N, K, T, seed, b, missing, fs)
        if extra_nan is not None:
            extra_nan(i, y)
        _, a3 = run_reference(y, phi, q, loadings, engine="numba-source")
        _, a4 = run_reference(y, phi, q, loadings, engine="numpy")
        # both reference engines agree: record how closely
        data[f"m{i}_mle_np_engine"] = a4["mle"]
        assert abs(a3["mle"] - a4["mle"]) <= 1e-12 * abs(a3["mle"]), (a3["mle"], a4["mle"])
        tsel = np.unique(np.r_[np.arange(0, T, cov_every), T - 1])
        data[f"m{i}_obs"] = y
        data[f"m{i}_alpha"] = alpha
        data[f"m{i}_loadings"] = loadings
        data[f"m{i}_phi"] = phi
        data[f"m{i}_q"] = q
        data[f"m{i}_tsel"] = tsel
        for k in ("sigmas", "detfs", "sigmacount", "count", "mle", "F", "Xp", "S"):
            data[f"m{i}_{k}"] = a3[k]
        for k in ("Pf", "Pp", "Ps"):
            data[f"m{i}_{k}"] = a3[k][tsel]
    data["nmodels"] = np.int64(len(models))
    np.savez_compressed(os.path.join(HERE, fname), **data)
    print(fname, {k: v.shape for k, v in data.items() if k.startswith("m0_")})


def g1_real():
    """examples/data 5-series / 1-factor model."""
    d = os.path.join(_refshim.REFERENCE_ROOT, "examples", "data")
    series = []
    for i in range(1, 6):
        s = pd.read_csv(f"{d}/B21B021400{i}_res.csv", index_col=0, parse_dates=True).squeeze()
        s.name = f"B21B021400{i}"
        series.append(s)
    mt = metran.Metran(series, name="B21B0214")
    mt.get_factors(mt.oseries)
    mt._init_kalmanfilter(mt.oseries, engine="numpy")
    mt.set_init_parameters()
    mt.kf.filtermethod = kfm.seqkalmanfilter  # numba source, un-jitted
    astar = np.array([5.501017, 13.560042, 4.682870, 11.381674, 13.140605, 22.980925])
    a10 = np.full(6, 10.0)
    mle_star = mt.get_mle(astar)
    mle_10 = mt.get_mle(a10)
    mt.kf.filtermethod = kfm.seqkalmanfilter_np
    mle_star_np = mt.get_mle(astar)
    mt.kf.filtermethod = kfm.seqkalmanfilter
    p = pd.Series(astar, index=mt.parameters.index)
    mt.parameters["optimal"] = astar
    Phi, Q, Z, R = mt._get_matrices(p)
    mt._run_kalman("smoother", p=p)
    kf = mt.kf
    T = kf.observations.shape[0]
    tsel = np.unique(np.r_[np.arange(0, 25), np.arange(25, T, 40), np.arange(T - 25, T)])
    Zs = mt.get_scaled_observation_matrix(p=p)
    sim_m, sim_v = kf.simulate(Zs, method="smoother")
    simf_m, simf_v = kf.simulate(Zs, method="filter")
    sdf_m, cdf_m = kf.decompose(Zs, method="smoother")
    sim5 = mt.get_simulation("B21B0214005", alpha=0.05)
    dec1 = mt.decompose_simulation("B21B0214001")
    smeans = mt.get_state_means()
    data = dict(
        obs=mt.oseries.values.astype(float),  # standardised, NaN = missing
        index_ns=mt.oseries.index.values.astype("datetime64[ns]").astype(np.int64),
        oseries_std=mt.oseries_std,
        oseries_mean=mt.oseries_mean,
        loadings=mt.factors,
        alpha_star=astar,
        alpha_10=a10,
        phi=np.diag(Phi).copy(),
        q=np.diag(Q).copy(),
        Z=Z,
        Z_scaled=Zs,
        mle_star=np.float64(mle_star),
        mle_star_np_engine=np.float64(mle_star_np),
        mle_10=np.float64(mle_10),
        sigmas=np.asarray(kf.sigmas),
        detfs=np.asarray(kf.detfs),
        count=kf.observation_count,
        tsel=tsel,
        F=kf.filtered_state_means,
        Xp=kf.predicted_state_means,
        S=kf.smoothed_state_means,
        Pf=kf.filtered_state_covariances[tsel],
        Pp=kf.predicted_state_covariances[tsel],
        Ps=kf.smoothed_state_covariances[tsel],
        sim_means=np.asarray(sim_m),
        sim_vars=np.asarray(sim_v),
        simf_means=np.asarray(simf_m)[tsel],
        simf_vars=np.asarray(simf_v)[tsel],
        sdf_means=np.asarray(sdf_m)[tsel],
        cdf_means=np.asarray(cdf_m)[:, tsel],
        get_simulation_005=sim5.values[:50],
        decompose_001=dec1.values[:50],
        state_means_head=smeans.values[:5],
        state_means_tail=smeans.values[-5:],
    )
    # masked re-smooth (examples/metran_practical_example.ipynb cell 31-33, tests/test_metran.py:32-40)
    oseries = mt.get_observations()
    mask = (0 * oseries).astype(bool)
    mask.loc["1997-8-28", "B21B0214005"] = True
    mt.mask_observations(mask)
    simm = mt.get_simulation("B21B0214005", alpha=None)
    # NB (0 * oseries).astype(bool) is True wherever oseries is NaN; the one *new* mask is the date
    data["mask_t"] = np.int64(oseries.index.get_loc(pd.Timestamp("1997-08-28")))
    data["masked_sim_005"] = simm.values
    data["masked_mle_star"] = np.float64(mt.get_mle(astar))
    mt.unmask_observations()
    np.savez_compressed(os.path.join(HERE, "g1_real.npz"), **data)
    print("g1_real", mle_star, mle_star_np, mle_10, data["masked_mle_star"])


def g1_solve():
    """Full Metran.solve() on examples/data (stored obj 2332.33, nfev 77, AIC 2344.33,
    examples/metran_practical_example.ipynb:142-155): the reference's ScipySolve (L-BFGS-B, 2-point finite
    differences, metran/solver.py:222-288) with the numpy engine."""
    d = os.path.join(_refshim.REFERENCE_ROOT, "examples", "data")
    series = []
    for i in range(1, 6):
        s = pd.read_csv(f"{d}/B21B021400{i}_res.csv", index_col=0, parse_dates=True).squeeze()
        s.name = f"B21B021400{i}"
        series.append(s)
    mt = metran.Metran(series, name="B21B0214")
    mt.solve(report=False, engine="numpy")
    np.savez_compressed(
        os.path.join(HERE, "g1_solve.npz"),
        optimal=mt.parameters["optimal"].values.astype(float),
        stderr=mt.parameters["stderr"].values.astype(float),
        obj=np.float64(mt.fit.obj_func), nfev=np.int64(mt.fit.nfev), aic=np.float64(mt.fit.aic),
        pcov=mt.fit.pcov.values.astype(float), names=np.array(list(mt.parameters.index)),
        initial=mt.parameters["initial"].values.astype(float), pmin=mt.parameters["pmin"].values.astype(float))
    print("g1_solve", mt.fit.obj_func, mt.fit.nfev)


def solver_covariance():
    """The reference's covariance helpers (metran/solver.py:65-192) run as they are on inputs that need no model:
    ``BaseSolver._get_covariance`` (nested forward differences, the positive-diagonal test and its repair step) on an analytic
    objective -- once at a minimum-like point, once with a fixed (non-varying) parameter in the callback, once on an objective
    whose Hessian is indefinite so that the repair runs --, ``_nearPSD`` on symmetric matrices (two indefinite ones among
    them) and ``_get_correlations``."""
    from pandas import DataFrame
    from metran.solver import BaseSolver

    out = {}
    sol = BaseSolver(mt=None)
    full = lambda p, callback: float(solver_objective(callback(p))[0])
    ident = lambda p: p
    x0 = np.array([5.2, 11.5, 7.3, 19.0])
    out["x0"] = x0.copy()
    out["cov_full"] = sol._get_covariance(x0.copy(), full, ident)
    # parameter 1 fixed at 12.5: the callback scatters the three varying values into the full vector (_array_todict, :290-305)
    vary = np.array([True, False, True, True])
    initial = np.array([0.0, 12.5, 0.0, 0.0])

    def todict(p):
        par = initial.copy()
        par[vary] = p
        return par

    out["vary"], out["initial"] = vary, initial
    out["cov_fixed"] = sol._get_covariance(x0[vary].copy(), full, todict)
    # _nearPSD / _get_correlations
    rng = np.random.default_rng(5)
    mats = []
    for k in range(4):
        M = rng.standard_normal((4, 4))
        M = 0.5 * (M + M.T) + (3.0, 3.0, 0.0, 1.2)[k] * np.eye(4)   # two positive definite, two indefinite
        mats.append(M)
    out["psd_in"] = np.stack(mats)
    out["psd_out"] = np.stack([BaseSolver._nearPSD(M.copy()) for M in mats])
    names = ["a", "b", "c", "d"]
    out["pcor"] = BaseSolver._get_correlations(DataFrame(out["cov_full"], index=names, columns=names)).values.astype(float)
    np.savez_compressed(os.path.join(HERE, "solver_covariance.npz"), **out)
    print("solver_covariance: stderr", np.sqrt(np.diag(out["cov_full"])), "| nearPSD of an indefinite matrix:\n", out["psd_out"][2])


def g2_seeded():
    """examples/dynamic_factor_model.ipynb cell 7: seeded 2-series synthetic."""
    np.random.seed(20210505)
    mean = np.zeros(3)
    scale = [1, 0.6, 2]
    noise = np.random.multivariate_normal(mean, np.diag(np.square(scale)), 2001)
    phi = np.array([0.80, 0.95, 0.90])
    a = np.zeros_like(noise)
    for i in range(1, noise.shape[0]):
        a[i] = noise[i] + np.multiply(a[i - 1], phi)
    s1 = np.add(a[1:, 0], a[1:, 2])
    s2 = np.add(a[1:, 1], a[1:, 2])
    s = pd.DataFrame(
        data=np.array([s1, s2]).T,
        index=pd.date_range(start="1-1-2000", periods=2000),
        columns=["series 1", "series 2"],
    )
    mt = metran.Metran(s)
    mt.get_factors(mt.oseries)
    mt._init_kalmanfilter(mt.oseries, engine="numpy")
    mt.set_init_parameters()
    mt.kf.filtermethod = kfm.seqkalmanfilter
    a10 = np.full(3, 10.0)
    mle = mt.get_mle(a10)
    p = pd.Series(a10, index=mt.parameters.index)
    Phi, Q, Z, R = mt._get_matrices(p)
    mt._run_kalman("smoother", p=p)
    kf = mt.kf
    tsel = np.arange(0, 2000, 50)
    np.savez_compressed(
        os.path.join(HERE, "g2_seeded.npz"),
        obs=mt.oseries.values.astype(float),
        loadings=mt.factors,
        alpha=a10,
        phi=np.diag(Phi).copy(),
        q=np.diag(Q).copy(),
        mle=np.float64(mle),
        sigmas=np.asarray(kf.sigmas),
        detfs=np.asarray(kf.detfs),
        S=kf.smoothed_state_means,
        F=kf.filtered_state_means,
        tsel=tsel,
        Ps=kf.smoothed_state_covariances[tsel],
        Pf=kf.filtered_state_covariances[tsel],
    )
    print("g2_seeded mle", mle, "loadings", mt.factors.ravel())


def exact_filter_smoother(y, phi, q, loadings, P0=None, dead=(), digits=60):
    """The same recursions (kalmanfilter.py:236-400, 403-476) in ``digits``-digit arithmetic (mpmath, a
    build-container-only dependency of THIS script): what the reference would return without rounding.
    ``dead`` lists states whose variance is exactly zero throughout (their rows/columns of Pp are exactly
    zero; the pseudo-inverse of :455 is then the inverse on the remaining states)."""
    import mpmath as mp

    mp.mp.dps = digits
    T, N = y.shape
    n = phi.shape[0]
    ph = [mp.mpf(float(v)) for v in phi]
    qq = [mp.mpf(float(v)) for v in q]
    Z = observation_matrix(loadings)
    Zm = [[mp.mpf(float(Z[j, c])) for c in range(n)] for j in range(N)]
    x = [mp.mpf(0)] * n
    P = [[mp.mpf(float((np.eye(n) if P0 is None else P0)[r, c])) for c in range(n)] for r in range(n)]
    F, Pf = [], []
    for t in range(T):
        x = [ph[r] * x[r] for r in range(n)]
        P = [[ph[r] * ph[c] * P[r][c] + (qq[r] if r == c else 0) for c in range(n)] for r in range(n)]
        for j in range(N):
            if not np.isfinite(y[t, j]):
                continue
            z = Zm[j]
            v = mp.mpf(float(y[t, j])) - sum(z[c] * x[c] for c in range(n))
            d = [sum(P[r][c] * z[c] for c in range(n)) for r in range(n)]
            f = sum(z[c] * d[c] for c in range(n))
            k = [d[r] / f for r in range(n)]
            x = [x[r] + k[r] * v for r in range(n)]
            P = [[P[r][c] - k[r] * d[c] for c in range(n)] for r in range(n)]
        F.append(list(x))
        Pf.append([row[:] for row in P])
    live = [i for i in range(n) if i not in dead]
    m = len(live)
    S = [None] * T
    Ps = [None] * T
    S[T - 1] = mp.matrix(F[T - 1])
    Ps[T - 1] = mp.matrix(Pf[T - 1])
    Phi = mp.diag(ph)
    Q = mp.diag(qq)
    for t in range(T - 2, -1, -1):
        Pft, Ft = mp.matrix(Pf[t]), mp.matrix(F[t])
        A = Phi * Pft * Phi + Q
        Al = mp.matrix(m, m)
        for a_, i in enumerate(live):
            for b_, j in enumerate(live):
                Al[a_, b_] = A[i, j]
        Ali = mp.inverse(Al)
        Ainv = mp.zeros(n, n)
        for a_, i in enumerate(live):
            for b_, j in enumerate(live):
                Ainv[i, j] = Ali[a_, b_]
        J = Pft * Phi * Ainv
        S[t] = Ft + J * (S[t + 1] - Phi * Ft)
        Ps[t] = Pft + J * (Ps[t + 1] - A) * J.T
    f64 = lambda M, r, c: np.array([[float(M[i, j]) for j in range(c)] for i in range(r)])  # noqa: E731
    return (np.array([[float(v) for v in row] for row in F]),
            np.array([[[float(v) for v in row] for row in Pm] for Pm in Pf]),
            np.array([[float(S[t][i]) for i in range(n)] for t in range(T)]),
            np.array([f64(Ps[t], n, n) for t in range(T)]))


def heywood():
    """(Near-)singular predicted covariances -- the ``pinv`` branch of the reference smoother (:455).

    m0  one series with communality exactly 1 => q = 0 for its specific state (metran.py:314-316), T = 800,
        every step observed: min eig(Pp) decays from 2e-2 to ~1e-17 (cond 1e17) over the record;
    m1  a common factor at alpha = 1e8 (phi = 1 - 1e-8, q = 2e-8): small but regular;
    m2  as m0 with P0[2,2] = 0 as well: that state is exactly zero throughout, Pp has an exactly zero
        row/column -- an LDL^T pivot of exactly 0.
    Stored: the reference's outputs AND the same recursion in 60-digit arithmetic (``*_exact``).  On m0 the
    reference is 7e-8 away from the exact smoothed means: numpy's pinv truncates singular values below
    1e-15 * s_max, and which side of that threshold a rounding-level singular value falls on is decided by
    LAPACK's rounding (so is the position of the oracle's Jacobi eigenvalues: the oracle differs from the
    reference by up to 1e-8 here, both from the exact answer by ~7e-8)."""
    N, K, T = 8, 2, 800
    data = {}
    cases = []
    y, alpha, load, phi, q = make_dfm(N, K, T, 5150, 0, 0.0, "observed")
    load0 = load.copy()
    load0[2] = [0.6, 0.8]
    phi0, q0 = phi_q_from_alpha(alpha, load0)
    assert q0[2] == 0.0
    cases.append((y, phi0, q0, load0, None, ()))
    a1 = alpha.copy()
    a1[N] = 1e8
    phi1, q1 = phi_q_from_alpha(a1, load)
    cases.append((y, phi1, q1, load, None, ()))
    P0 = np.eye(N + K)
    P0[2, 2] = 0.0
    cases.append((y[:200], phi0, q0, load0, P0, (2,)))
    for i, (yy, ph, qq, ld, p0, dead) in enumerate(cases):
        kf = kfm.SPKalmanFilter(engine="numpy")
        kf.filtermethod = kfm.seqkalmanfilter
        kf.set_observations(pd.DataFrame(yy))
        Phi = np.diag(ph)
        kf.set_matrices(Phi, np.diag(qq), observation_matrix(ld), np.zeros(N))
        kf.run_filter(initial_state_covariance=None if p0 is None else p0.copy())
        S, Ps = kfm.kalmansmoother(kf.filtered_state_means, kf.filtered_state_covariances,
                                   kf.predicted_state_means, kf.predicted_state_covariances, Phi)
        Fe, Pfe, Se, Pse = exact_filter_smoother(yy, ph, qq, ld, P0=p0, dead=dead)
        Tn = yy.shape[0]
        tsel = np.unique(np.r_[np.arange(0, Tn, 40), Tn - 1, Tn - 2])
        mineig = np.array([np.linalg.eigvalsh(kf.predicted_state_covariances[t])[0] for t in range(Tn)])
        d = dict(obs=yy, phi=ph, q=qq, loadings=ld, P0=np.eye(N + K) if p0 is None else p0, tsel=tsel,
                 mle=np.float64(kf.get_mle()), F=kf.filtered_state_means, Pf=kf.filtered_state_covariances[tsel],
                 S=S, Ps=Ps[tsel], F_exact=Fe, S_exact=Se, Ps_exact=Pse[tsel], Pf_exact=Pfe[tsel], mineig_Pp=mineig)
        for k, v in d.items():
            data[f"m{i}_{k}"] = v
        print("heywood m%d: min eig(Pp) %.1e..%.1e | reference vs exact: F %.1e S %.1e Ps %.1e" % (
            i, mineig[1:].min(), mineig[1:].max(), np.abs(kf.filtered_state_means - Fe).max(),
            np.abs(S - Se).max(), np.abs(Ps - Pse).max()))
    data["nmodels"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "heywood.npz"), **data)


def factor_analysis():
    """Row f4: ``FactorAnalysis.solve`` of the reference (metran/factoranalysis.py:42-119) and its pieces on
    examples/data (BASELINE G1e: loadings 0.857982 ...), the seeded notebook model (G2: 0.93540765), the 2 x 2
    matrix of the reference's own unit tests (tests/test_factoranalysis.py:10-17) and seeded synthetic models with
    one, two and three common factors, missing observations included (pairwise-complete correlations).  What
    scipy's L-BFGS-B did inside ``_minres`` is recorded as well (message, nit, nfev, x): it returns its start vector
    in every case (see metran_amd/factoranalysis.py)."""
    import scipy.optimize as scopt

    from metran.factoranalysis import FactorAnalysis

    cases = []
    d = os.path.join(_refshim.REFERENCE_ROOT, "examples", "data")
    series = []
    for i in range(1, 6):
        s = pd.read_csv(f"{d}/B21B021400{i}_res.csv", index_col=0, parse_dates=True).squeeze()
        s.name = f"B21B021400{i}"
        series.append(s)
    cases.append(("g1", metran.Metran(series, name="B21B0214").oseries.values.astype(float)))
    np.random.seed(20210505)
    noise = np.random.multivariate_normal(np.zeros(3), np.diag(np.square([1, 0.6, 2])), 2001)
    a = np.zeros_like(noise)
    for i in range(1, noise.shape[0]):
        a[i] = noise[i] + np.multiply(a[i - 1], np.array([0.80, 0.95, 0.90]))
    cases.append(("g2", np.array([a[1:, 0] + a[1:, 2], a[1:, 1] + a[1:, 2]]).T))
    rng = np.random.default_rng(424242)

    def synth(N, K, T, strength, miss):
        load = np.zeros((N, K))
        for j in range(N):  # simple structure + small cross-loadings
            load[j, j % K] = strength * rng.uniform(0.8, 1.0)
            load[j] += rng.uniform(-0.08, 0.08, size=K)
        f = np.zeros((T, K))
        e = np.zeros((T, N))
        for t in range(1, T):
            f[t] = 0.9 * f[t - 1] + rng.standard_normal(K) * np.sqrt(1 - 0.81)
            e[t] = 0.7 * e[t - 1] + rng.standard_normal(N) * np.sqrt(1 - 0.49)
        y = f @ load.T + e * np.sqrt(np.maximum(1 - (load ** 2).sum(1), 0.05))
        y[rng.random((T, N)) < miss] = np.nan
        return y

    cases.append(("s8k2", synth(8, 2, 1500, 0.85, 0.2)))
    cases.append(("s12k3", synth(12, 3, 2500, 0.8, 0.1)))
    cases.append(("s6k1", synth(6, 1, 800, 0.7, 0.3)))
    cases.append(("s20k4", synth(20, 4, 3000, 0.85, 0.0)))
    cases.append(("weak", rng.standard_normal((400, 5))))  # no structure: Kaiser fallback / whatever the reference does
    data = {"names": np.array([c[0] for c in cases])}
    orig = scopt.minimize
    for name, y in cases:
        rec = {}

        def spy(fun, x0, *a_, **k_):
            r = orig(fun, x0, *a_, **k_)
            rec.update(x0=np.array(x0, float), x=np.array(r.x, float), nit=int(r.nit), nfev=int(r.nfev),
                       message=str(r.message), fun0=float(fun(np.array(x0, float), *k_["args"])),
                       grad0=np.array(k_["jac"](np.array(x0, float), *k_["args"]), float))
            return r

        scopt.minimize = spy
        try:
            fa = FactorAnalysis()
            df = pd.DataFrame(y)
            factors = fa.solve(df)
            corr = fa._get_correlations(df)
            ev, evec = fa._get_eigval(corr)
            nfm, nfm4 = fa._maptest(corr, evec, ev)
        finally:
            scopt.minimize = orig
        nf = 0 if factors is None else factors.shape[1]
        out = dict(obs=y, corr=corr, eigval=fa.eigval, nfactors_map=np.int64(nfm), nfactors_map4=np.int64(nfm4),
                   nfactors=np.int64(nf), factors=np.zeros((y.shape[1], 0)) if factors is None else factors,
                   fep=np.float64(getattr(fa, "fep", np.nan) if factors is not None else np.nan))
        if rec:
            out.update(psi0=rec["x0"], psi=rec["x"], nit=np.int64(rec["nit"]), nfev=np.int64(rec["nfev"]),
                       fun0=np.float64(rec["fun0"]), grad0=rec["grad0"], message=np.array(rec["message"]),
                       loadings_unrotated=fa._get_loadings(rec["x"], corr, max(nf, 1)))
        for k, v in out.items():
            data[f"{name}_{k}"] = v
        print("factor_analysis %-6s N=%d nf(map,map4,used)=(%d,%d,%d) lbfgsb: %s nit=%s |x-x0|=%.1e" % (
            name, y.shape[1], nfm, nfm4, nf, rec.get("message"), rec.get("nit"),
            np.abs(rec["x"] - rec["x0"]).max() if rec else float("nan")))
    c2 = np.array([[1.0, 0.8], [0.8, 1.0]])
    ev, evec = FactorAnalysis()._get_eigval(c2)
    data["unit_corr"] = c2
    data["unit_eigval"] = ev
    data["unit_maptest"] = np.array(FactorAnalysis()._maptest(c2, evec, ev), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "factor_analysis.npz"), **data)


def factor_multi():
    """Row f4, multi-factor models (round-2 verdict, item 1): the reference's ``FactorAnalysis.solve`` on seeded random
    20- and 32-series models with 4 or 8 true factors in block structure (loadings 0.7-0.9, T = 1000) -- the MAP test
    as the reference's code behaves then selects 2 (or 3) factors -- searched until the fixture holds, per series count,
    at least 6 models for which ``np.linalg.eig`` inside ``_get_loadings`` (factoranalysis.py:396-398) does NOT return
    the nf largest eigenvalues first, 4 for which it returns them in another order, and 10 ordinary ones; plus a few
    with three factors.  Stored per model: the reference's correlation matrix, eigenvalues, factor counts, the start
    and the final vector of scipy's L-BFGS-B inside ``_minres``, its iteration count, the unrotated loadings, the
    final (rotated, sign-fixed) loadings and the ranks of the eigenvalues ``eig`` returned first (information)."""
    import logging

    import scipy.optimize as scopt

    from metran.factoranalysis import FactorAnalysis

    logging.disable(logging.CRITICAL)
    orig = scopt.minimize
    data = {}
    names = []

    def block_model(N, K, T, rng):
        load = np.zeros((N, K))
        for j in range(N):
            load[j, j * K // N] = rng.uniform(0.7, 0.9)
        return rng.standard_normal((T, K)) @ load.T + rng.standard_normal((T, N)) * np.sqrt(1 - (load ** 2).sum(1))

    def run(y):
        rec = {}

        def spy(fun, x0, *a_, **k_):
            r = orig(fun, x0, *a_, **k_)
            rec.update(x0=np.array(x0, float), x=np.array(r.x, float), nit=int(r.nit), message=str(r.message))
            return r

        scopt.minimize = spy
        try:
            fa = FactorAnalysis()
            df = pd.DataFrame(y)
            factors = fa.solve(df)
            corr = fa._get_correlations(df)
            ev, evec = fa._get_eigval(corr)
            nfm, nfm4 = fa._maptest(corr, evec, ev)
        finally:
            scopt.minimize = orig
        if factors is None or not rec:
            return None
        nf = factors.shape[1]
        psi = rec["x"]
        sc = np.diag(1 / np.sqrt(psi))
        w = np.linalg.eig(np.dot(sc, np.dot(corr, sc)))[0].real
        pos = np.argsort(-w, kind="stable")
        rank = np.empty(len(w), dtype=np.int64)
        rank[pos] = np.arange(len(w))
        return dict(corr=corr, eigval=fa.eigval, nfactors_map=np.int64(nfm), nfactors_map4=np.int64(nfm4),
                    nfactors=np.int64(nf), psi0=np.clip(rec["x0"], 0.005, 1), psi=psi, nit=np.int64(rec["nit"]),
                    loadings_unrotated=fa._get_loadings(psi, corr, nf), factors=factors, fep=np.float64(fa.fep),
                    eig_rank=rank[:nf])

    for N, K, seed, want in ((20, 4, 20001, dict(nondominant=6, permuted=4, sorted=10)),
                             (32, 4, 32001, dict(nondominant=6, permuted=4, sorted=10)),
                             (32, 8, 32801, dict(nondominant=2, permuted=1, sorted=3))):
        rng = np.random.default_rng(seed)
        have = dict.fromkeys(want, 0)
        tries = 0
        while any(have[k] < want[k] for k in want) and tries < 2000:
            tries += 1
            out = run(block_model(N, K, 1000, rng))
            if out is None:
                continue
            nf = int(out["nfactors"])
            rk = list(out["eig_rank"])
            kind = "sorted" if rk == list(range(nf)) else ("permuted" if sorted(rk) == list(range(nf)) else "nondominant")
            if nf < 2 or have[kind] >= want[kind]:
                continue
            have[kind] += 1
            name = "n%dk%d_%s%d" % (N, K, kind[0], have[kind])
            names.append(name)
            for k, v in out.items():
                data[f"{name}_{k}"] = v
            print("factor_multi %-14s nf=%d eig ranks %s nit=%d moved=%s" % (name, nf, rk, int(out["nit"]),
                                                                             bool(np.abs(out["psi"] - out["psi0"]).max() > 0)))
        print("factor_multi N=%d K=%d: %s after %d models" % (N, K, have, tries))
    # multi-factor models for which scipy's L-BFGS-B LEAVES its start vector (its path then runs through _minresgrad,
    # i.e. through eig's order, at every iterate)
    # (rare: about one multi-factor model in forty; varied structures are searched until three are found)
    rng = np.random.default_rng(777)
    have = tries = 0
    while have < 3 and tries < 20000:
        tries += 1
        N, K = int(rng.integers(4, 25)), int(rng.integers(2, 6))
        load = np.zeros((N, K))
        for j in range(N):
            load[j, j * K // N] = rng.uniform(0.5, 0.95)
            if tries % 3 == 1:
                load[j] += rng.uniform(-0.2, 0.2, K)
        if tries % 3 == 2:
            load = rng.uniform(-0.6, 0.6, (N, K))
        u = 1 - (load ** 2).sum(1)
        if (u <= 0.02).any():
            continue
        out = run(rng.standard_normal((400, K)) @ load.T + rng.standard_normal((400, N)) * np.sqrt(u))
        if out is None or int(out["nfactors"]) < 2 or not np.abs(out["psi"] - out["psi0"]).max() > 0:
            continue
        have += 1
        name = "mv%d" % have
        names.append(name)
        for k, v in out.items():
            data[f"{name}_{k}"] = v
        print("factor_multi %-14s N=%d nf=%d eig ranks %s nit=%d moved=True (model %d of the search)" % (
            name, N, int(out["nfactors"]), list(out["eig_rank"]), int(out["nit"]), tries))
    data["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "factor_multi.npz"), **data)
    logging.disable(logging.NOTSET)


def edge_nan(i, y):
    # model 0: series 1 never observed; model 1: a run of fully-empty steps and an inf
    if i == 0:
        y[:, 1] = np.nan
    elif i == 1:
        y[3:7, :] = np.nan
        y[9, 0] = np.inf
    elif i == 2:
        y[:, :] = np.nan
        y[5, 2] = 0.25  # single observation in the whole record


def heywood_wide():
    """A WIDE model (17 series, 1 factor: a shape of the tape path, 16 < n, N <= 32) with a series of communality exactly 1
    (q = 0 for its specific state, metran.py:314-316) AND 35 % missing observations: the predicted covariance turns singular
    along the record while the smoother still has something to do (heywood.npz is fully observed: its smoothed observables are
    the observations).  Stored: the reference's projected smoothed means / variances (kalmansmoother + simulate) and the same
    recursion in 60-digit arithmetic -- what the inverse-free backward pass is compared with (tests/test_dk_tape.py)."""
    N, K, T = 17, 1, 240
    y, alpha, load, phi, q = make_dfm(N, K, T, 6160, 0, 0.35, "observed")
    load = load.copy()
    load[4] = [1.0]
    alpha = alpha.copy()
    alpha[4] = 5.0   # phi = 0.82: the variance of that state (q = 0) has decayed to rounding level after ~100 steps
    phi, q = phi_q_from_alpha(alpha, load)
    assert q[4] == 0.0
    kf = kfm.SPKalmanFilter(engine="numpy")
    kf.filtermethod = kfm.seqkalmanfilter
    kf.set_observations(pd.DataFrame(y))
    Phi, Z = np.diag(phi), observation_matrix(load)
    kf.set_matrices(Phi, np.diag(q), Z, np.zeros(N))
    kf.run_filter()
    S, Ps = kfm.kalmansmoother(kf.filtered_state_means, kf.filtered_state_covariances, kf.predicted_state_means,
                               kf.predicted_state_covariances, Phi)
    Fe, Pfe, Se, Pse = exact_filter_smoother(y, phi, q, load)
    mineig = np.array([np.linalg.eigvalsh(kf.predicted_state_covariances[t])[0] for t in range(T)])
    proj = lambda m, P: (m @ Z.T, np.einsum("jn,tnm,jm->tj", Z, P, Z))  # noqa: E731
    m_ref, v_ref = proj(S, Ps)
    m_ex, v_ex = proj(Se, Pse)
    print("heywood_wide: min eig(Pp) %.1e..%.1e | reference vs exact: projected means %.1e variances %.1e" % (
        mineig[1:].min(), mineig[1:].max(), np.abs(m_ref - m_ex).max(), np.abs(v_ref - v_ex).max()))
    np.savez_compressed(os.path.join(HERE, "heywood_wide.npz"), obs=y, phi=phi, q=q, loadings=load, mle=np.float64(kf.get_mle()),
                        sim_means=m_ref, sim_vars=v_ref, sim_means_exact=m_ex, sim_vars_exact=v_ex, mineig_Pp=mineig)


if __name__ == "__main__":
    if len(sys.argv) > 1:  # regenerate selected fixtures only: python make_golden.py heywood ...
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    heywood()
    heywood_wide()
    factor_analysis()
    factor_multi()
    g1_real()
    g1_solve()
    solver_covariance()
    g2_seeded()
    # C2 shape (8 series / 2 factors), small T: every array, every step
    synthetic_case("c2_small.npz", 8, 2, 48, seed=2000, models=[0, 1, 2])
    # C2 shape at full T=1000: means everywhere, covariances every 50 steps
    synthetic_case("c2_T1000.npz", 8, 2, 1000, seed=2000, models=[0, 1], cov_every=50)
    # C4 shape (32 series / 4 factors) with 30 % missing; model 1 has an EMPTY first step
    synthetic_case("c4_missing.npz", 32, 4, 36, seed=4000, models=[0, 1], missing=0.3,
                   first_steps=["observed", "empty"], cov_every=5)
    # C4 shape over a longer horizon (wide kernels: one model per wavefront), covariances every 50 steps
    synthetic_case("c4_T400.npz", 32, 4, 400, seed=4001, models=[0], missing=0.3, cov_every=50)
    # small odd shapes / degenerate missingness
    synthetic_case("edge_cases.npz", 3, 1, 14, seed=77, models=[0, 1, 2], missing=0.25,
                   first_steps=["observed", "empty", "random"], extra_nan=edge_nan)
    synthetic_case("n17_k3.npz", 14, 3, 30, seed=1703, models=[0], missing=0.1)
