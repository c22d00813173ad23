"""GPU: the four lock-step L-BFGS kernels of the batched calibration (metran_amd/csrc/mk_lbfgs.hip) against the torch code they
replaced (tests/oracle_engine.py::TorchLbfgs, the restatement the CPU tier drives ``calibrate_batch`` over): random flights with
active bounds, converged and NaN models, every model's own history ring empty, partial, full and wrapped anywhere, rejected /
accepted / non-finite trial values, both forms of the line search (lock-step; every model its own, with a back-tracking budget).  Tolerance 1e-11 relative on directions (the kernels sum in index order and contract multiply-adds, torch does neither); masks
and counts exactly -- the later stages are fed the restatement's outputs, so a rounding-level difference cannot flip a test."""
import numpy as np
import pytest
import torch

from oracle_engine import TorchLbfgs

pytestmark = pytest.mark.gpu


def _flight(R, n, H, seed):
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    x = rng.uniform(0.5, 20.0, (R, n))
    lo = np.full((R, n), 1e-5)
    on = rng.random((R, n)) < 0.15
    x[on] = lo[on]                                             # parameters ON their bound
    g = rng.normal(size=(R, n)) * 10.0 ** rng.uniform(-7, 1, (R, 1))   # some models below gtol
    g[3 % R] = np.nan
    Sh = rng.normal(size=(H, R, n)) * 0.1
    Yh = Sh * rng.uniform(0.5, 2.0, (H, R, 1)) + 0.01 * rng.normal(size=(H, R, n))
    rho = 1.0 / np.maximum((Sh * Yh).sum(2), 1e-300)
    hlen = rng.integers(0, H + 1, R).astype(np.int32)          # every model its own ring: empty, partial, full
    hlen[: min(R, 3)] = [0, H, 1][: min(R, 3)]
    hpos = rng.integers(0, H, R).astype(np.int32)              # ... wrapped anywhere
    active = rng.random(R) < 0.9
    return dict(x=t(x), lo=t(lo), g=t(g), Sh=t(Sh), Yh=t(Yh), rho=t(rho), hlen=t(hlen), hpos=t(hpos), active=t(active),
                f=t(rng.normal(size=R) * 100 + 2000))


def _dev(d):
    return {k: v.cuda() for k, v in d.items()}


@pytest.mark.parametrize("own", [False, True])
@pytest.mark.parametrize("R,n,history", [(257, 10, 10), (5, 6, 10), (1000, 36, 10), (33, 3, 4)])
def test_lbfgs_kernels_equal_the_torch_restatement(R, n, history, own):
    """``own``: the own-line-search form of the adjoint mode (phase / step / nback per model, accepted mask, masked update)."""
    from metran_amd.engine import BatchedKalman

    kf = BatchedKalman(0)
    c = _flight(R, n, history, seed=R + 11 * own)
    gpu = _dev(c)
    rng = np.random.default_rng(R)
    # ---- direction
    pg_c, d_c = torch.from_numpy(rng.normal(size=(R, n))), torch.from_numpy(rng.normal(size=(R, n)))   # what a model in mid-search keeps
    pg_g, d_g = pg_c.cuda(), d_c.cuda()
    phase = torch.from_numpy((rng.random(R) < 0.3).astype(np.uint8)) if own else None
    step = torch.from_numpy(10.0 ** rng.uniform(-2, 0, R))
    nback = torch.from_numpy(rng.integers(0, 5, R).astype(np.int32)) if own else None
    ph_g, st_g, nb_g = (None if v is None else v.cuda() for v in (phase, step if own else None, nback))
    na_c = TorchLbfgs.lbfgs_direction(c["x"], c["g"], c["lo"], c["active"], c["Sh"], c["Yh"], c["rho"], c["hlen"], c["hpos"], 1e-5, pg_c, d_c,
                                      phase, step if own else None, nback)
    na_g = kf.lbfgs_direction(gpu["x"], gpu["g"], gpu["lo"], gpu["active"], gpu["Sh"], gpu["Yh"], gpu["rho"], gpu["hlen"], gpu["hpos"], 1e-5,
                              pg_g, d_g, ph_g, st_g, nb_g)
    assert na_g == na_c and torch.equal(gpu["active"].cpu(), c["active"])
    torch.testing.assert_close(pg_g.cpu(), pg_c, rtol=0, atol=0, equal_nan=True)
    act = c["active"]
    torch.testing.assert_close(d_g.cpu()[act], d_c[act], rtol=1e-11, atol=1e-13)
    if own:
        assert torch.equal(ph_g.cpu(), phase) and torch.equal(nb_g.cpu(), nback)
        torch.testing.assert_close(st_g.cpu(), step, rtol=0, atol=0)
    # ---- trial point + Armijo test
    searching = act.clone()
    x_new, f_new = c["x"].clone(), c["f"].clone()
    xt_c, xe_c = torch.empty_like(c["x"]), torch.empty_like(c["x"])
    TorchLbfgs.lbfgs_trial(c["x"], d_c, step, c["lo"], searching, x_new, xt_c, xe_c)
    g_step, g_search, g_xnew, g_fnew = step.cuda(), searching.cuda(), x_new.cuda(), f_new.cuda()
    xt_g, xe_g = torch.empty_like(gpu["x"]), torch.empty_like(gpu["x"])
    kf.lbfgs_trial(gpu["x"], d_c.cuda(), g_step, gpu["lo"], g_search, g_xnew, xt_g, xe_g)
    torch.testing.assert_close(xt_g.cpu(), xt_c, rtol=1e-12, atol=1e-300, equal_nan=True)   # (the kernel contracts x + step d)
    torch.testing.assert_close(xe_g.cpu(), xe_c, rtol=1e-12, atol=1e-300, equal_nan=True)
    gd = (pg_c * (xt_c - c["x"])).sum(1)
    ft = c["f"] + gd * torch.from_numpy(rng.uniform(-0.5, 1.5, R)) + 1e-9   # some pass, some fail the sufficient-decrease test
    ft[1 % R] = float("nan")
    ft[2 % R] = float("inf")
    if own:
        nback.copy_(torch.from_numpy(rng.integers(0, 4, R).astype(np.int32)))   # some models on their last trial point (budget 4)
        nb_g = nback.cuda()
        acc_c, acc_g = torch.zeros(R, dtype=torch.bool), torch.ones(R, dtype=torch.bool).cuda()
        out_c = TorchLbfgs.lbfgs_armijo(ft, c["f"], pg_c, xt_c, c["x"], searching, step, x_new, f_new, nback, 4, acc_c)
        out_g = kf.lbfgs_armijo(ft.cuda(), gpu["f"], pg_c.cuda(), xt_c.cuda(), gpu["x"], g_search, g_step, g_xnew, g_fnew, nb_g, 4, acc_g)
        assert out_g == out_c and torch.equal(acc_g.cpu(), acc_c) and torch.equal(nb_g.cpu(), nback)
    else:
        out_c = TorchLbfgs.lbfgs_armijo(ft, c["f"], pg_c, xt_c, c["x"], searching, step, x_new, f_new)
        out_g = kf.lbfgs_armijo(ft.cuda(), gpu["f"], pg_c.cuda(), xt_c.cuda(), gpu["x"], g_search, g_step, g_xnew, g_fnew)
        assert out_g == out_c
    assert torch.equal(g_search.cpu(), searching)
    torch.testing.assert_close(g_xnew.cpu(), x_new, rtol=0, atol=0, equal_nan=True)
    torch.testing.assert_close(g_fnew.cpu(), f_new, rtol=0, atol=0, equal_nan=True)
    torch.testing.assert_close(g_step.cpu(), step, rtol=1e-12, atol=0)
    # ---- history update: every model's own ring
    for keep_old in ((False,) if own else (True, False)):
        cc = {k: v.clone() for k, v in c.items()}
        gg = {k: v.clone() for k, v in gpu.items()}
        g_new = torch.from_numpy(rng.normal(size=(R, n)))
        ph_c = torch.ones(R, dtype=torch.uint8) if own else None
        ph_g = None if ph_c is None else ph_c.cuda()
        ng_c = TorchLbfgs.lbfgs_update(cc["x"], cc["f"], cc["g"], x_new, f_new, g_new, keep_old, None if own else searching, cc["active"], 2.2e-9,
                                       cc["Sh"], cc["Yh"], cc["rho"], cc["hlen"], cc["hpos"], acc_c if own else None, ph_c)
        ng_g = kf.lbfgs_update(gg["x"], gg["f"], gg["g"], x_new.cuda(), f_new.cuda(), g_new.cuda(), keep_old, None if own else searching.cuda(),
                               gg["active"], 2.2e-9, gg["Sh"], gg["Yh"], gg["rho"], gg["hlen"], gg["hpos"], acc_c.cuda() if own else None, ph_g)
        assert ng_g == ng_c and torch.equal(gg["active"].cpu(), cc["active"])
        assert torch.equal(gg["hlen"].cpu(), cc["hlen"]) and torch.equal(gg["hpos"].cpu(), cc["hpos"])
        if own:
            assert torch.equal(ph_g.cpu(), ph_c)
        for k in ("x", "f", "g"):
            torch.testing.assert_close(gg[k].cpu(), cc[k], rtol=0, atol=0, equal_nan=True)
        for k in ("Sh", "Yh", "rho"):
            torch.testing.assert_close(gg[k].cpu(), cc[k], rtol=1e-12, atol=0, equal_nan=True)
    kf.close()


def test_calibration_over_the_kernels_reaches_scipys_optimum():
    """calibrate_batch on the device (the four kernels in its loop) against scipy's L-BFGS-B per model on the oracle's objective."""
    from scipy.optimize import minimize

    import oracle
    from metran_amd.calibrate import calibrate_batch
    from metran_amd.engine import BatchedKalman
    from metran_amd.params import phi_q_from_alpha
    from metran_amd.synthetic import make_dfm_batch

    R, N, K, T = 6, 5, 1, 300
    d = make_dfm_batch(R, N, K, T, seed=21, missing=0.2)
    kf = BatchedKalman(0)
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    res = calibrate_batch(kf, maxiter=150)
    assert bool(res.converged.all())
    for r in range(R):
        def fun(a):
            phi, q = phi_q_from_alpha(a, d["loadings"][r])
            return float(oracle.dfm_batch(d["obs"][r:r + 1], phi[None], q[None], d["loadings"][r:r + 1], smooth=False, outputs="mle")["mle"][0])
        ref = minimize(fun, np.full(N + K, 10.0), method="L-BFGS-B", bounds=[(1e-5, None)] * (N + K))
        assert float(res.obj[r]) <= ref.fun + 1e-6 * abs(ref.fun)
    kf.close()
