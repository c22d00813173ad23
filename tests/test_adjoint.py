"""Adjoint gradient of -2 log L ("analytic/adjoint gradient").  The reference has no
gradient; the bar is agreement with central differences of the ORACLE's objective (restating
metran/kalmanfilter.py:236-400, 550-567) and with the numpy restatement of the backward pass."""
import numpy as np
import pytest

import adjoint_ref
import oracle
from metran_amd.params import phi_q_from_alpha
from metran_amd.synthetic import make_dfm, make_dfm_batch


def _oracle_mle(y, phi, q, G):
    return float(oracle.dfm_batch(y[None], phi[None], q[None], G[None], smooth=False, outputs="mle")["mle"][0])


def test_numpy_adjoint_matches_central_differences_of_the_oracle():
    for seed, (N, K, T, miss, first) in enumerate([(5, 2, 60, 0.25, "random"), (3, 1, 40, 0.0, "full"), (4, 1, 50, 0.5, "empty")]):
        y, alpha, G, phi, q = make_dfm(N, K, T, 40 + seed, 0, miss, first)
        mle, gphi, gq = adjoint_ref.gradient(y, phi, q, G)
        assert abs(mle - _oracle_mle(y, phi, q, G)) <= 1e-10 * abs(mle)
        for i in range(N + K):
            for vec, g in ((phi, gphi), (q, gq)):
                h = 1e-6
                p = vec.copy(); p[i] += h
                a = _oracle_mle(y, p if vec is phi else phi, p if vec is q else q, G)
                p = vec.copy(); p[i] -= h
                b = _oracle_mle(y, p if vec is phi else phi, p if vec is q else q, G)
                fd = (a - b) / (2 * h)
                assert abs(g[i] - fd) <= 2e-6 * max(1.0, abs(fd)), (seed, i, g[i], fd)


def test_numpy_adjoint_property():
    """Property test (hypothesis, derandomised) of the backward pass's restatement -- the thing the GPU kernels are compared
    with: any small shape and missingness pattern (whole steps empty, the first among them), warm-up 0..3 (the compressed
    warm-up index of get_mle, kalmanfilter.py:550-567), a given initial state and covariance, observation variances: the gradient
    w.r.t. phi and q equals central differences of the forward recursion, whose value is the oracle's where the oracle takes
    the same arguments."""
    from hypothesis import given, settings, strategies as st

    @st.composite
    def case(draw):
        N, K, T = draw(st.integers(1, 4)), draw(st.integers(1, 2)), draw(st.integers(1, 10))
        rng = np.random.default_rng(draw(st.integers(0, 2 ** 31 - 1)))
        n = N + K
        G = rng.uniform(0.2, 0.6, (N, K)) / np.sqrt(K)
        phi = np.exp(-1.0 / rng.uniform(0.7, 40.0, n))
        q = (1.0 - phi ** 2) * np.r_[1.0 - (G ** 2).sum(1), np.ones(K)]
        y = rng.standard_normal((T, N))
        y[rng.random((T, N)) < draw(st.sampled_from([0.0, 0.3, 0.8]))] = np.nan
        if draw(st.booleans()):
            y[rng.random(T) < 0.4] = np.nan
        if draw(st.booleans()):
            y[0] = np.nan
        x0 = P0 = R = None
        if draw(st.booleans()):
            x0 = rng.standard_normal(n)
            A = rng.standard_normal((n, n))
            P0 = A @ A.T / n + 0.1 * np.eye(n)
        if draw(st.booleans()):
            R = rng.uniform(0.0, 0.4, N) * (rng.random(N) < 0.6)
        return y, phi, q, G, draw(st.integers(0, 3)), x0, P0, R

    @settings(max_examples=150, deadline=None, derandomize=True)
    @given(case())
    def check(c):
        y, phi, q, G, warmup, x0, P0, R = c
        mle, gphi, gq = adjoint_ref.gradient(y, phi, q, G, warmup=warmup, x0=x0, P0=P0, R=R)
        if x0 is None:
            ref = oracle.dfm_batch(y[None], phi[None], q[None], G[None], obsvar=None if R is None else R[None], warmup=warmup,
                                   smooth=False, outputs="mle")["mle"][0]
            assert abs(mle - ref) <= 1e-10 * max(1.0, abs(ref))
        f = lambda ph, qq: adjoint_ref.forward(y, ph, qq, G, warmup, x0, P0, R)[0]  # noqa: E731
        for i in range(len(phi)):
            for vec, g in ((phi, gphi), (q, gq)):
                h = 1e-5 * max(abs(vec[i]), 1e-2)
                up, dn = vec.copy(), vec.copy()
                up[i] += h
                dn[i] -= h
                fd = ((f(up, q) - f(dn, q)) if vec is phi else (f(phi, up) - f(phi, dn))) / (2 * h)
                assert abs(g[i] - fd) <= 1e-5 * max(1.0, abs(fd)), (i, vec is phi, g[i], fd)

    check()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["model_major", "time_major"])
@pytest.mark.parametrize("N,K,T,B,missing,first", [(8, 2, 120, 37, 0.0, "full"), (8, 2, 90, 21, 0.3, "random"),
                                                   (5, 1, 70, 19, 0.2, "empty"), (2, 1, 60, 5, 0.1, "random"),
                                                   (6, 2, 50, 8, 0.4, "random"),
                                                   # wide models (n > 16): adjoint_wide_kernel, one model
                                                   # per wavefront; forward pass = the split-layout filter
                                                   (32, 4, 40, 5, 0.3, "random"), (14, 3, 50, 6, 0.2, "empty"),
                                                   (32, 4, 30, 3, 0.0, "full")])
def test_hip_adjoint_gradient(layout, N, K, T, B, missing, first):
    from metran_amd.engine import BatchedKalman

    d = make_dfm_batch(B, N, K, T, seed=500 + N, missing=missing, first_step=first)
    kf = BatchedKalman(0, layout=layout)
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    mle, gphi, gq = (t.cpu().numpy() for t in kf.loglik_grad(d["phi"], d["q"]))
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"], smooth=False, outputs="mle")["mle"]
    np.testing.assert_allclose(mle, ref, rtol=1e-11)
    for b in range(B):
        _, rp, rq = adjoint_ref.gradient(d["obs"][b], d["phi"][b], d["q"][b], d["loadings"][b])
        np.testing.assert_allclose(gphi[b], rp, rtol=1e-9, atol=1e-9 * np.abs(rp).max())
        np.testing.assert_allclose(gq[b], rq, rtol=1e-9, atol=1e-9 * np.abs(rq).max())
    # chain rule to alpha vs central differences of the oracle through the reference parametrisation
    mle2, galpha = (t.cpu().numpy() for t in kf.loglik_grad_alpha(d["alpha"]))
    np.testing.assert_allclose(mle2, ref, rtol=1e-11)
    for b in (0, B - 1):
        for i in range(N + K):
            h = 1e-5 * d["alpha"][b, i]
            vals = []
            for sgn in (+1, -1):
                al = d["alpha"][b].copy()
                al[i] += sgn * h
                ph, qq = phi_q_from_alpha(al, d["loadings"][b])
                vals.append(_oracle_mle(d["obs"][b], ph, qq, d["loadings"][b]))
            fd = (vals[0] - vals[1]) / (2 * h)
            assert abs(galpha[b, i] - fd) <= 1e-5 * max(1e-3, abs(fd)), (b, i, galpha[b, i], fd)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,T,B,missing,first,extra", [(32, 4, 60, 7, 0.3, "random", False), (14, 3, 50, 6, 0.2, "empty", True),
                                                         (32, 4, 30, 3, 0.0, "full", True)])
def test_wide_adjoint_update_tape(N, K, T, B, missing, first, extra):
    """Round 6: with an update tape on the context (``mk_set_adjoint_updates``; ``BatchedKalman.adjoint_updates``, the default)
    the recording forward pass keeps (d, 1/f, v) of every scalar update and ``adjoint_wide_kernel<.., UPD>`` reads them instead of
    recomputing each step from the filtered record of the step before.  Same objective bit for bit, the same gradient to rounding
    as the recomputing walk AND the numpy adjoint; observation variances, initial moments, the two-phase form; a tape that is too
    small for the call is not used; the 16-lane kernel has none."""
    import ctypes

    import torch

    from metran_amd.engine import BatchedKalman

    d = make_dfm_batch(B, N, K, T, seed=900 + N, missing=missing, first_step=first)
    n = N + K
    rng = np.random.default_rng(N)
    kw = {}
    R = None
    if extra:
        R = rng.uniform(0.0, 0.3, (B, N)) * (rng.random((B, N)) < 0.5)
        A = rng.normal(size=(B, n, n))
        kw = dict(x0=rng.normal(size=(B, n)), P0=A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n))
    res = {}
    for upd in (False, True):
        # (the tape is written by the one-model-per-wavefront filter: the same forward kernel on both sides of the comparison)
        kf = BatchedKalman(0, layout="time_major").set_variant("wide_filter", "lane_per_state")
        kf.adjoint_updates = upd
        kf.set_observations(d["obs"]).set_loadings(d["loadings"], R)
        assert int(kf._L.mk_adjoint_update_stride(N, K)) == N * (n + (n & 1) + 2)
        res[upd] = tuple(t.cpu().numpy() for t in kf.loglik_grad(d["phi"], d["q"], **kw))
        assert (getattr(kf, "_grad_upd", None) is not None) == upd
        if upd:   # the two phases separately: the backward launch reads the tape the forward launch wrote
            m2 = kf.loglik_forward(d["phi"], d["q"], **kw).cpu().numpy()
            g2 = tuple(t.cpu().numpy() for t in kf.loglik_backward())
            assert np.array_equal(m2, res[True][0]) and np.array_equal(g2[0], res[True][1]) and np.array_equal(g2[1], res[True][2])
            # a tape too small for the call is simply not used: the recomputing walk's numbers, bit for bit
            small = torch.empty(64, dtype=torch.float64, device="cuda")
            assert kf._L.mk_set_adjoint_updates(kf._ctx, ctypes.c_void_p(small.data_ptr()), 64) == 0
            kf._ensure_grad_updates = lambda B_: None          # (keep the engine from re-attaching its own, large enough, tape)
            r3 = tuple(t.cpu().numpy() for t in kf.loglik_grad(d["phi"], d["q"], **kw))
            assert all(np.array_equal(a, b) for a, b in zip(r3, res[False]))
        kf.close()
    assert np.array_equal(res[False][0], res[True][0])                       # the objective does not know about the tape
    for a, b in zip(res[False][1:], res[True][1:]):
        np.testing.assert_allclose(b, a, rtol=0, atol=1e-12 * np.abs(a).max())
    if not extra:
        for b in range(B):
            _, rp, rq = adjoint_ref.gradient(d["obs"][b], d["phi"][b], d["q"][b], d["loadings"][b])
            np.testing.assert_allclose(res[True][1][b], rp, rtol=1e-9, atol=1e-9 * np.abs(rp).max())
            np.testing.assert_allclose(res[True][2][b], rq, rtol=1e-9, atol=1e-9 * np.abs(rq).max())
    assert int(BatchedKalman(0)._L.mk_adjoint_update_stride(8, 2)) == 0      # n <= 16: the 16-lane kernel recomputes


@pytest.mark.gpu
def test_wide_adjoint_at_the_reference_tolerance():
    """(32, 4) -- configs[3]'s shape -- against central differences of the ORACLE's objective at 1e-6 (the verdict's bar),
    with x0 / P0 / R given, and the same gradient whichever wide filter wrote the records."""
    from metran_amd.engine import BatchedKalman

    B, N, K, T = 3, 32, 4, 25
    d = make_dfm_batch(B, N, K, T, seed=3204, missing=0.3, first_step="random")
    rng = np.random.default_rng(8)
    x0 = 0.3 * rng.normal(size=(B, N + K))
    A = rng.normal(size=(B, N + K, N + K)) * 0.1
    P0 = np.eye(N + K)[None] + A @ A.transpose(0, 2, 1)
    R = rng.uniform(0.01, 0.1, size=(B, N))
    kf = BatchedKalman(0).set_observations(d["obs"]).set_loadings(d["loadings"], obsvar=R)
    assert kf.has_adjoint()
    mle, gphi, gq = (t.cpu().numpy() for t in kf.loglik_grad(d["phi"], d["q"], x0=x0, P0=P0))
    kf.set_variant("wide_filter", "lane_per_state")
    mle2, gphi2, gq2 = (t.cpu().numpy() for t in kf.loglik_grad(d["phi"], d["q"], x0=x0, P0=P0))
    np.testing.assert_allclose(gphi2, gphi, rtol=1e-9, atol=1e-9 * np.abs(gphi).max())
    np.testing.assert_allclose(gq2, gq, rtol=1e-9, atol=1e-9 * np.abs(gq).max())

    def obj(b, phi, q):  # the numpy restatement of the reference's filter + get_mle (pinned to the oracle above)
        return adjoint_ref.forward(d["obs"][b], phi, q, d["loadings"][b], x0=x0[b], P0=P0[b], R=R[b])[0]

    for b in range(B):
        m, rp, rq = adjoint_ref.gradient(d["obs"][b], d["phi"][b], d["q"][b], d["loadings"][b], x0=x0[b], P0=P0[b], R=R[b])
        assert abs(mle[b] - m) <= 1e-11 * abs(m)
        np.testing.assert_allclose(gphi[b], rp, rtol=1e-9, atol=1e-9 * np.abs(rp).max())
        np.testing.assert_allclose(gq[b], rq, rtol=1e-9, atol=1e-9 * np.abs(rq).max())
    b = 1
    for i in (0, 7, 31, 32, 35):
        for which, g in (("phi", gphi[b]), ("q", gq[b])):
            h = 1e-6
            up, dn = d[which][b].copy(), d[which][b].copy()
            up[i] += h
            dn[i] -= h
            if which == "phi":
                fd = (obj(b, up, d["q"][b]) - obj(b, dn, d["q"][b])) / (2 * h)
            else:
                fd = (obj(b, d["phi"][b], up) - obj(b, d["phi"][b], dn)) / (2 * h)
            assert abs(g[i] - fd) <= 1e-6 * max(1.0, abs(fd)), (which, i, g[i], fd)


@pytest.mark.gpu
def test_adjoint_at_the_lower_bound():
    """alpha at Metran's lower bound 1e-5 (metran/metran.py:446-462) underflows phi to 0: the gradient stays
    finite and the components of those parameters vanish with d phi / d alpha."""
    from metran_amd.engine import BatchedKalman

    d = make_dfm_batch(6, 8, 2, 80, seed=77, missing=0.2)
    alpha = d["alpha"].copy()
    alpha[:, 0] = 1e-5
    alpha[2, 9] = 1e-5
    kf = BatchedKalman(0).set_observations(d["obs"]).set_loadings(d["loadings"])
    mle, g = (t.cpu().numpy() for t in kf.loglik_grad_alpha(alpha))
    assert np.isfinite(mle).all() and np.isfinite(g).all()
    assert not g[:, 0].any() and g[2, 9] == 0.0 and np.abs(g[:, 1:9]).min() > 0
    for b in (0, 2):
        ph, qq = phi_q_from_alpha(alpha[b], d["loadings"][b])
        assert abs(mle[b] - _oracle_mle(d["obs"][b], ph, qq, d["loadings"][b])) <= 1e-10 * abs(mle[b])


@pytest.mark.gpu
def test_adjoint_with_initial_state_and_observation_variance():
    """Caller-supplied x0 / P0 (run_filter arguments, kalmanfilter.py:696-750) and a non-zero observation
    variance R (set_matrices) flow through the backward pass as constants."""
    from metran_amd.engine import BatchedKalman

    B, N, K, T = 7, 5, 1, 60
    d = make_dfm_batch(B, N, K, T, seed=91, missing=0.2)
    rng = np.random.default_rng(3)
    x0 = rng.normal(size=(B, N + K))
    A = rng.normal(size=(B, N + K, N + K)) * 0.2
    P0 = np.eye(N + K)[None] + A @ A.transpose(0, 2, 1)
    R = rng.uniform(0.01, 0.2, size=(B, N))
    kf = BatchedKalman(0).set_observations(d["obs"]).set_loadings(d["loadings"], obsvar=R)
    mle, gphi, gq = (t.cpu().numpy() for t in kf.loglik_grad(d["phi"], d["q"], x0=x0, P0=P0))
    for b in range(B):
        m, rp, rq = adjoint_ref.gradient(d["obs"][b], d["phi"][b], d["q"][b], d["loadings"][b], x0=x0[b], P0=P0[b], R=R[b])
        assert abs(mle[b] - m) <= 1e-11 * abs(m)
        np.testing.assert_allclose(gphi[b], rp, rtol=1e-9, atol=1e-9 * np.abs(rp).max())
        np.testing.assert_allclose(gq[b], rq, rtol=1e-9, atol=1e-9 * np.abs(rq).max())


@pytest.mark.gpu
def test_adjoint_on_a_runtime_specialised_shape(tmp_path_factory, monkeypatch):
    """(7,2) is not in the ahead-of-time list: the shape module built at run time carries the adjoint
    kernel too (mkmod_launch_adjoint)."""
    import os

    from metran_amd.engine import BatchedKalman

    monkeypatch.setenv("METRAN_HIP_CACHE", os.environ.get("METRAN_HIP_CACHE", str(tmp_path_factory.getbasetemp() / "mkjit")))
    d = make_dfm_batch(5, 7, 2, 50, seed=72, missing=0.2)
    kf = BatchedKalman(0).set_observations(d["obs"]).set_loadings(d["loadings"])
    mle, gphi, gq = (t.cpu().numpy() for t in kf.loglik_grad(d["phi"], d["q"]))
    for b in range(5):
        m, rp, rq = adjoint_ref.gradient(d["obs"][b], d["phi"][b], d["q"][b], d["loadings"][b])
        assert abs(mle[b] - m) <= 1e-11 * abs(m)
        np.testing.assert_allclose(gphi[b], rp, rtol=1e-9, atol=1e-9 * np.abs(rp).max())
        np.testing.assert_allclose(gq[b], rq, rtol=1e-9, atol=1e-9 * np.abs(rq).max())


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(8, 2), (32, 4)])
def test_forward_and_backward_phases_separately(N, K):
    """``mk_loglik_grad_phases``: several recording forward passes at different points, then ONE backward walk = the
    gradient of the LAST point, bit for bit what ``mk_loglik_grad`` returns there (the line search of calibrate_batch)."""
    import torch

    from metran_amd.engine import BatchedKalman

    B, T = 11, 60
    d = make_dfm_batch(B, N, K, T, seed=900 + N, missing=0.2, first_step="random")
    kf = BatchedKalman(0, layout="time_major")
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    alpha = torch.from_numpy(d["alpha"]).cuda()
    ref_mle, ref_g = kf.loglik_grad_alpha(alpha)
    other_mle, _ = kf.loglik_grad_alpha(alpha * 1.7)
    m1 = kf.loglik_forward_alpha(alpha * 1.7)          # a rejected trial point ...
    m2 = kf.loglik_forward_alpha(alpha)                # ... then the accepted one
    g = kf.loglik_backward_alpha()
    assert torch.equal(m1, other_mle) and torch.equal(m2, ref_mle)
    assert torch.equal(g, ref_g)
    kf2 = BatchedKalman(0, layout="time_major")
    kf2.set_observations(d["obs"]).set_loadings(d["loadings"])
    with pytest.raises(Exception):
        kf2.loglik_backward()                          # no forward pass yet
    kf.close(), kf2.close()
