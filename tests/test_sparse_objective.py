"""mk_loglik on ONE shared record (the solver's finite-difference instances) walks only the observed steps
and applies runs of empty steps in closed form (loglik_sparse_kernel).  Same objective as the reference's
step-by-step recursion (oracle restating kalmanfilter.py:236-400, 550-567) to 1e-10 relative."""
import numpy as np
import pytest

import oracle
from metran_amd.params import phi_q_from_alpha
from metran_amd.synthetic import make_dfm, make_dfm_batch

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _sparse_record(N, K, T, seed, keep_every, first):
    y, alpha, G, phi, q = make_dfm(N, K, T, seed, 0, 0.2, "observed")
    rng = np.random.default_rng(seed)
    keep = np.zeros(T, bool)
    keep[rng.choice(T, size=max(2, T // keep_every), replace=False)] = True
    keep[0] = first
    y[~keep] = np.nan
    return y, alpha, G


@pytest.mark.parametrize("N,K,T,keep_every,first", [(5, 1, 900, 18, True), (5, 1, 400, 9, False), (8, 2, 600, 25, False),
                                                    (2, 1, 300, 3, True), (8, 2, 200, 1, True), (3, 1, 50, 50, False),
                                                    (5, 1, 1400, 2, True), (8, 2, 600, 1, False)])   # 700 / 600 observed steps: several LDS tiles of 256
def test_sparse_objective_equals_the_oracle(N, K, T, keep_every, first):
    from metran_amd.engine import BatchedKalman

    y, alpha, G = _sparse_record(N, K, T, 800 + N + T, keep_every, first)
    S = 13                                                  # parameter sets sharing the record
    rng = np.random.default_rng(T)
    alphas = alpha[None] * rng.uniform(0.5, 2.0, size=(S, N + K))
    alphas[3, 0] = 1e-5                                     # phi underflows to 0 (Metran's lower bound)
    kf = BatchedKalman(0).set_observations(y[None]).set_loadings(G[None])
    phi, q = kf.params_from_alpha(alphas)
    mle = kf.loglik(phi, q).cpu().numpy()
    ph, qq = phi.cpu().numpy(), q.cpu().numpy()
    ref = oracle.dfm_batch(np.repeat(y[None], S, 0), ph, qq, np.repeat(G[None], S, 0), smooth=False, outputs="mle")["mle"]
    np.testing.assert_allclose(mle, ref, rtol=1e-10, atol=1e-10)
    # warm-up variants of get_mle and caller-supplied initial state / observation variance
    x0 = rng.normal(size=(S, N + K))
    A = rng.normal(size=(S, N + K, N + K)) * 0.3
    P0 = np.eye(N + K)[None] + A @ A.transpose(0, 2, 1)
    R = rng.uniform(0.01, 0.3, size=(1, N))
    kf.set_loadings(G[None], obsvar=R)
    import adjoint_ref  # numpy restatement of the same recursion with x0 / P0 / R arguments

    for warmup in (0, 1, 3):
        mle = kf.loglik(phi, q, warmup=warmup, x0=x0, P0=P0).cpu().numpy()
        for s_ in (0, 3, S - 1):
            ref = adjoint_ref.forward(y, ph[s_], qq[s_], G, warmup=warmup, x0=x0[s_], P0=P0[s_], R=R[0])[0]
            assert abs(mle[s_] - ref) <= 1e-10 * max(1.0, abs(ref)), (warmup, s_, mle[s_], ref)


def test_sparse_objective_on_examples_data(g1):
    """examples/data: 343 observed of 6255 daily steps; reference values of BASELINE.md (G1)."""
    from metran_amd.engine import BatchedKalman

    kf = BatchedKalman(0).set_observations(g1["obs"][None]).set_loadings(g1["loadings"][None])
    kf.enable_timing(True)
    phi, q = kf.params_from_alpha(np.stack([g1["alpha_star"], g1["alpha_10"]]))
    mle = kf.loglik(phi, q).cpu().numpy()
    assert abs(mle[0] - 2332.327069381027) < 1e-8 and abs(mle[1] - 2384.792799342231) < 1e-8
    ms, _ = kf.last_kernel_ms()
    assert ms < 3.0, ms     # the step-by-step filter needs ~9 ms for the 6255 steps of this record


def test_observed_step_list_is_cached_per_record_and_invalidated(g1):
    """The list of observed steps is built once per uploaded record (the solver evaluates the objective ~80 times
    on it) and rebuilt when the record changes: a new buffer, or the same buffer changed in place followed by
    mk_observations_changed (which every mutating engine call issues)."""
    from metran_amd._lib import check
    from metran_amd.engine import BatchedKalman

    kf = BatchedKalman(0).set_observations(g1["obs"][None]).set_loadings(g1["loadings"][None])
    phi, q = kf.params_from_alpha(g1["alpha_star"][None])
    first = float(kf.loglik(phi, q)[0])
    assert float(kf.loglik(phi, q)[0]) == first == float(kf.loglik(phi, q)[0])     # cached list: same value
    # mask one observation through the engine (new buffer + notification): the reference's masked value
    import torch

    mask = torch.zeros((1,) + g1["obs"].shape, dtype=torch.uint8)
    mask[0, int(g1["mask_t"]), 4] = 1
    kf.mask_observations(mask)
    assert abs(float(kf.loglik(phi, q)[0]) - float(g1["masked_mle_star"])) < 1e-8
    kf.unmask_observations()
    assert float(kf.loglik(phi, q)[0]) == first
    # in-place change of the SAME buffer (an empty step receives an observation): stale until the context is told
    t_new = int(np.nonzero(~np.isfinite(g1["obs"]).any(1))[0][100])
    kf.obs[0, t_new, 2] = 0.4
    stale = float(kf.loglik(phi, q)[0])
    check(kf._L.mk_observations_changed(kf._ctx))
    fresh = float(kf.loglik(phi, q)[0])
    y = g1["obs"].copy()
    y[t_new, 2] = 0.4
    ph, qq = phi.cpu().numpy(), q.cpu().numpy()
    ref = oracle.dfm_batch(y[None], ph, qq, g1["loadings"][None], smooth=False, outputs="mle")["mle"][0]
    assert abs(fresh - ref) < 1e-8 and stale == first and abs(first - ref) > 1e-3


@pytest.mark.parametrize("N,K,T,keep_every,first", [(5, 1, 400, 17, "observed"), (8, 2, 300, 5, "empty"), (3, 1, 120, 1, "observed"), (14, 2, 90, 7, "empty"),
                                                    (5, 1, 1300, 2, "observed"), (8, 2, 513, 1, "empty")])   # > 256 observed steps: several LDS tiles
def test_single_record_filter_walks_the_observed_steps(N, K, T, keep_every, first):
    """The record-writing filter of ONE record (the engine route of Metran.solve: mk_filter, <= 16 instances, all four state
    arrays): observed steps walked one after the other, the records of the empty steps written in closed form by a second
    kernel (kalmanfilter.py:335).  Against the oracle, and against the step-by-step batched kernel
    (``set_variant("single_record", "stepwise")``); three parameter sets on the one record, initial moments, observation
    variances, a persistence of exactly zero."""
    from metran_amd.engine import BatchedKalman
    from metran_amd.params import observation_matrix

    d = make_dfm_batch(1, N, K, T, seed=60 + N, missing=0.2, first_step=first)
    y = d["obs"][0].copy()
    mask = np.ones(T, bool)
    mask[::keep_every] = False
    y[mask] = np.nan                                   # runs of keep_every - 1 empty steps
    if first == "empty":
        y[:3] = np.nan
    y[-2:] = np.nan                                    # and an empty tail
    n = N + K
    rng = np.random.default_rng(N)
    S = 3
    phi = np.clip(d["phi"][0][None] * (1.0 + 0.02 * rng.standard_normal((S, n))), 0.0, 0.999999)
    phi[1, 0] = 0.0                                    # alpha at its lower bound: phi underflows to 0
    q = d["q"][0][None] * (1.0 + 0.02 * rng.standard_normal((S, n)))
    R = rng.uniform(0.0, 0.2, N) * (rng.random(N) < 0.5)
    x0 = rng.normal(size=(S, n))
    A = rng.normal(size=(S, n, n))
    P0 = A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n)
    kf = BatchedKalman()
    kf.set_observations(y[None]).set_loadings(d["loadings"], R[None])
    assert kf.get_variant("single_record") == "sparse"
    r = kf.filter(phi, q, x0=x0, P0=P0)
    kf.set_variant("single_record", "stepwise")
    r2 = kf.filter(phi, q, x0=x0, P0=P0)
    o, oi, oc = oracle.set_observations(y)
    for s in range(S):
        sg, df, sc, F, Pf, Xp, Pp = oracle.seqkalmanfilter(o, np.diag(phi[s]), np.diag(q[s]), observation_matrix(d["loadings"][0]), R, oi, oc,
                                                           x0[s], P0[s])
        for res in (r, r2):
            assert int(_np(res["sigmacount"])[s]) == sc
            np.testing.assert_allclose(_np(res["F"])[s], F, atol=1e-10)
            np.testing.assert_allclose(_np(res["Pf"])[s], Pf, atol=1e-10)
            np.testing.assert_allclose(_np(res["Xp"])[s], Xp, atol=1e-10)
            np.testing.assert_allclose(_np(res["Pp"])[s], Pp, atol=1e-10)
            np.testing.assert_allclose(_np(res["sigmas"])[s, :sc], sg[:sc], rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(_np(res["detfs"])[s, :sc], df[:sc], atol=1e-10)
            assert not _np(res["sigmas"])[s, sc:].any() and not _np(res["detfs"])[s, sc:].any()
            ref = oracle.get_mle(sg[:sc], df[:sc], oc)
            assert abs(_np(res["mle"])[s] - ref) <= 1e-9 * abs(ref)
    kf.close()
