"""Test infrastructure: the slice of ``metran_amd.engine.BatchedKalman`` that the HOST-side drivers touch
(``calibrate_batch``, ``HipSolve`` / ``BatchObjective``), on the CPU: values from the C oracle (``oracle_dfm_batch``),
gradients from the numpy adjoint restatement (tests/adjoint_ref.py).  It lets the CPU tier run those drivers unchanged
(tests/test_calibrate_host.py, tests/test_solver_host.py); the product never sees it -- ``BatchedKalman`` has no CPU path."""
import numpy as np
import torch

import adjoint_ref
import oracle
from metran_amd.params import phi_q_from_alpha


class TorchLbfgs:
    """The L-BFGS steps of ``calibrate_batch`` as plain torch operations -- the restatement of ``metran_amd/csrc/mk_lbfgs.hip``
    (same arguments and in-place conventions as ``BatchedKalman.lbfgs_*``; every model has its own history ring): the CPU tier
    runs the driver over it, tests/test_lbfgs_gpu.py compares the kernels with it."""

    @staticmethod
    def _pair(Sh, Yh, rho, hlen, hpos, i):
        """Pair i of every model counted from the NEWEST (i = 0), and whether the model has it."""
        H, R = Sh.shape[0], Sh.shape[1]
        slot = ((hpos + hlen - 1 - i) % H).long().clamp_min(0)
        ar = torch.arange(R)
        return Sh[slot, ar], Yh[slot, ar], rho[slot, ar], (i < hlen)

    @staticmethod
    def lbfgs_direction(x, g, lo, active, Sh, Yh, rho, hlen, hpos, gtol, pg, d, phase=None, step=None, nback=None):
        H = Sh.shape[0]
        keep = (phase.bool() & active) if phase is not None else torch.zeros_like(active)   # in the middle of a line search
        bound = (x <= lo) & (g > 0)
        pg_new = torch.where(bound, torch.zeros_like(g), g)
        act_new = active & (pg_new.abs().amax(1) > gtol)
        qv = pg_new.clone()
        al = []
        for i in range(H):
            S_, Y_, r_, has = TorchLbfgs._pair(Sh, Yh, rho, hlen, hpos, i)
            a_ = torch.where(has, r_ * (S_ * qv).sum(1), torch.zeros_like(r_))
            al.append(a_)
            qv = qv - a_[:, None] * Y_
        S_, Y_, _, has0 = TorchLbfgs._pair(Sh, Yh, rho, hlen, hpos, 0)
        gamma = (S_ * Y_).sum(1) / (Y_ * Y_).sum(1).clamp_min(1e-300)
        qv = torch.where(has0[:, None], qv * gamma[:, None], qv)
        for i in range(H - 1, -1, -1):
            S_, Y_, r_, has = TorchLbfgs._pair(Sh, Yh, rho, hlen, hpos, i)
            b_ = r_ * (Y_ * qv).sum(1)
            qv = torch.where(has[:, None], qv + (al[i] - b_)[:, None] * S_, qv)
        dd = -qv
        bad = (dd * pg_new).sum(1) >= 0
        dd = torch.where(bad[:, None], -pg_new, dd)
        first = hlen == 0
        dd = torch.where(first[:, None], dd / pg_new.abs().amax(1, keepdim=True).clamp_min(1e-300), dd)
        dd = torch.where(bound, torch.zeros_like(dd), dd)
        dd = torch.where(act_new[:, None], dd, torch.zeros_like(dd))
        pg.copy_(torch.where(keep[:, None], pg, pg_new))
        d.copy_(torch.where(keep[:, None], d, dd))
        active.copy_(torch.where(keep, active, act_new))
        started = active & ~keep
        if step is not None:
            step.copy_(torch.where(started, torch.ones_like(step), step))
        if nback is not None:
            nback.copy_(torch.where(started, torch.zeros_like(nback), nback))
        if phase is not None:
            phase.copy_(torch.where(started, torch.ones_like(phase), phase))
        return int(active.sum())

    @staticmethod
    def lbfgs_trial(x, d, step, lo, searching, x_new, xt, xe):
        xt.copy_(torch.maximum(x + step[:, None] * d, lo))
        xe.copy_(torch.where(searching[:, None], xt, x_new))

    @staticmethod
    def lbfgs_armijo(ft, f, pg, xt, x, searching, step, x_new, f_new, nback=None, max_backtracks=0, accepted=None):
        gd = (pg * (xt - x)).sum(1)
        ok = searching & (ft <= f + 1e-4 * gd) & torch.isfinite(ft)
        x_new.copy_(torch.where(ok[:, None], xt, x_new))
        f_new.copy_(torch.where(ok, ft, f_new))
        curv = ft - f - gd
        theta = torch.where(torch.isfinite(ft) & (curv > 0), -gd / (2.0 * curv), torch.full_like(ft, 0.1))
        if nback is None:                                 # lock-step form
            searching &= ~ok
            step.copy_(torch.where(searching, step * theta.clamp(0.1, 0.5), step))
            return int(searching.sum())
        accepted.copy_(ok)                                # own line search per model
        rejected = searching & ~ok
        nback.copy_(torch.where(rejected, nback + 1, nback))
        out = rejected & (nback >= max_backtracks)        # out of trial points: done (at numerical precision)
        searching &= ~out
        cont = rejected & ~out
        step.copy_(torch.where(cont, step * theta.clamp(0.1, 0.5), step))
        return int(cont.sum()), int(ok.sum())

    @staticmethod
    def lbfgs_update(x, f, g, x_new, f_new, g_new, keep_old, searching, active, ftol, Sh, Yh, rho, hlen, hpos, mask=None, phase=None,
                     nit=None, maxiter=0):
        H, R = Sh.shape[0], Sh.shape[1]
        m = torch.ones_like(active) if mask is None else mask.bool()
        srch = searching.bool() if (searching is not None and mask is None) else torch.zeros_like(active)
        gn = torch.where(srch[:, None], g, g_new) if keep_old else g_new
        s_ = x_new - x
        y_ = gn - g
        sy = (s_ * y_).sum(1)
        good = m & (sy > 1e-10 * (y_ * y_).sum(1).clamp_min(1e-300))
        slot = ((hpos + hlen) % H).long()
        idx = good.nonzero().squeeze(1)
        Sh[slot[idx], idx] = s_[idx]
        Yh[slot[idx], idx] = y_[idx]
        rho[slot[idx], idx] = 1.0 / sy[idx].clamp_min(1e-300)
        full = hlen >= H
        hpos.copy_(torch.where(good & full, (hpos + 1) % H, hpos))
        hlen.copy_(torch.where(good & ~full, hlen + 1, hlen))
        rel = (f - f_new) / torch.maximum(torch.maximum(f.abs(), f_new.abs()), torch.ones_like(f))
        was_active = active.clone()
        active.copy_(torch.where(m, active & ~srch & (rel > ftol), active))
        if nit is not None:   # per-model iteration count; a model at maxiter leaves the flight (scipy's maxiter, per model)
            nit.copy_(torch.where(m & was_active, nit + 1, nit))
            if maxiter > 0:
                active.copy_(active & ~(m & was_active & (nit >= maxiter)))
        x.copy_(torch.where(m[:, None], x_new, x))
        g.copy_(torch.where(m[:, None], gn, g))
        f.copy_(torch.where(m, f_new, f))
        if phase is not None:
            phase.copy_(torch.where(m, torch.zeros_like(phase), phase))
        return int(good.sum())


class OracleEngine(TorchLbfgs):
    """The slice of ``BatchedKalman`` that ``calibrate_batch`` touches, on the CPU, values from the C oracle
    (``oracle_dfm_batch``) and gradients from the numpy adjoint restatement (tests/adjoint_ref.py); the L-BFGS steps from the
    torch restatement above."""

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


class OracleFilterEngine:
    """The slice of ``BatchedKalman`` that the drop-in layer touches (``metran_amd.kalmanfilter``: ``seqkalmanfilter_hip``,
    ``kalmansmoother_hip``, the ``SPKalmanFilter`` mirror), for ONE record: ``set_observations`` / ``set_loadings`` / ``filter``
    / ``smooth`` / ``filter_smooth`` / ``simulate`` / ``decompose`` answered by the C oracle's per-model entry points.
    ``uploads`` counts ``set_observations`` calls with host data (the upload cache's business)."""

    def __init__(self):
        self.obs = None
        self.uploads = 0
        self.calls = []

    def set_observations(self, obs):
        if isinstance(obs, torch.Tensor):                # "already resident": the mirror hands back what it got from us
            self.obs = obs
            return self
        self.uploads += 1
        self.obs = torch.from_numpy(np.array(obs, dtype=np.float64))
        self.R, self.T, self.N = self.obs.shape
        assert self.R == 1
        return self

    def set_loadings(self, loadings, obsvar=None):
        self.loadings = np.asarray(loadings, float)[0]
        self.obsvar = np.zeros(self.N) if obsvar is None else np.asarray(obsvar, float)[0]
        self.K = self.loadings.shape[1]
        return self

    def _Z(self):
        return np.concatenate([np.eye(self.N), self.loadings], axis=1)

    def filter(self, phi, q, warmup=1, x0=None, P0=None):
        self.calls.append("filter")
        n = self.N + self.K
        o, oi, oc = oracle.set_observations(self.obs[0].numpy())
        x0 = np.zeros(n) if x0 is None else np.asarray(x0, float)[0]
        P0 = np.eye(n) if P0 is None else np.asarray(P0, float)[0]
        sg, df, sc, F, Pf, Xp, Pp = oracle.seqkalmanfilter(o, np.diag(phi[0]), np.diag(q[0]), self._Z(), self.obsvar, oi, oc, x0, P0)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None]  # noqa: E731
        mle = oracle.get_mle(sg[:sc], df[:sc], oc, warmup)
        bad = not np.all(np.isfinite(df[:sc]))
        return {"sigmas": t(sg), "detfs": t(df), "sigmacount": torch.tensor([sc]), "F": t(F), "Pf": t(Pf), "Xp": t(Xp), "Pp": t(Pp),
                "mle": torch.tensor([mle], dtype=torch.float64), "status": torch.tensor([1 if bad else 0], dtype=torch.int32)}

    def smooth(self, phi, q, F, Pf):
        self.calls.append("smooth")
        F = (F[0].numpy() if isinstance(F, torch.Tensor) else np.asarray(F)[0])
        Pf = (Pf[0].numpy() if isinstance(Pf, torch.Tensor) else np.asarray(Pf)[0])
        ph, qq = np.asarray(phi, float)[0], np.asarray(q, float)[0]
        Xp = np.vstack([np.zeros((1, F.shape[1])), F[:-1] * ph])                       # predicted moments from the filtered ones
        Pp = np.concatenate([np.eye(F.shape[1])[None], Pf[:-1] * np.outer(ph, ph) + np.diag(qq)])
        S, Ps = oracle.kalmansmoother(F, Pf, Xp, Pp, np.diag(ph))
        return {"S": torch.from_numpy(S)[None], "Ps": torch.from_numpy(Ps)[None], "status": torch.zeros(1, dtype=torch.int32)}

    def smooth_dense(self, phi, F, Pf, Xp, Pp):
        self.calls.append("smooth_dense")
        a = lambda v: (v[0].numpy() if isinstance(v, torch.Tensor) else np.asarray(v)[0])  # noqa: E731
        S, Ps = oracle.kalmansmoother(a(F), a(Pf), a(Xp), a(Pp), np.diag(np.asarray(phi, float)[0]))
        return {"S": torch.from_numpy(S)[None], "Ps": torch.from_numpy(Ps)[None], "status": torch.zeros(1, dtype=torch.int32)}

    def filter_smooth(self, phi, q, warmup=1):
        r = self.filter(phi, q, warmup)
        S, Ps = oracle.kalmansmoother(r["F"][0].numpy(), r["Pf"][0].numpy(), r["Xp"][0].numpy(), r["Pp"][0].numpy(), np.diag(phi[0]))
        r["S"], r["Ps"] = torch.from_numpy(S)[None], torch.from_numpy(Ps)[None]
        return r

    def simulate(self, Z, means, covariances):
        sm, sv = oracle.simulate(Z, means[0], covariances[0])
        return torch.from_numpy(sm)[None], torch.from_numpy(sv)[None]

    def decompose(self, Z, means):
        sdf, cdf = oracle.decompose(Z, means[0])
        return torch.from_numpy(sdf)[None], torch.from_numpy(cdf)[None]
