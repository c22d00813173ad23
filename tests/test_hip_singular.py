"""GPU: (near-)singular predicted covariances and the status bits.

Contract: the smoother's LDL^T inverts every POSITIVE pivot, however small, and drops
the direction of a pivot <= 0 (1/d := 0, like the pseudo-inverse of kalmanfilter.py:455 drops a null
direction; flag MK_FLAG_RANK_DEFICIENT, informational); a pivot < -1e-8 (indefinite) or an innovation
variance <= 0 sets an ERROR bit and every adapter raises.  Measured against the same recursion in 60-digit
arithmetic (tests/golden/heywood.npz, ``*_exact``) this is closer to the exact answer than the reference
itself, whose pinv truncation costs 9e-8 on the smoothed means of m0."""
import numpy as np
import pytest

import oracle
from conftest import golden_models
from metran_amd.params import observation_matrix

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("layout", ["model_major", "time_major"])
def test_heywood_fixture(layout):
    from metran_amd.engine import FLAG_NONPOSITIVE_F, FLAG_NOT_SPD, FLAG_RANK_DEFICIENT, BatchedKalman

    kf = BatchedKalman(layout=layout)
    for i, m in golden_models("heywood.npz"):
        kf.set_observations(m["obs"][None]).set_loadings(m["loadings"][None])
        r = kf.filter_smooth(m["phi"][None], m["q"][None], P0=m["P0"][None])
        ts = m["tsel"]
        st = int(_np(r["status"])[0])
        assert not st & (FLAG_NONPOSITIVE_F | FLAG_NOT_SPD)
        if i == 2:
            assert st & FLAG_RANK_DEFICIENT          # an exactly-zero pivot was met and dropped
        if i == 1:
            assert st == 0
        S, Ps, F = _np(r["S"])[0], _np(r["Ps"])[0], _np(r["F"])[0]
        assert abs(float(_np(r["mle"])[0]) - float(m["mle"])) <= 1e-9 * abs(float(m["mle"]))
        np.testing.assert_allclose(F, m["F"], atol=1e-10)
        # against exact arithmetic: the repo's usual 1e-9 bar holds on all three
        np.testing.assert_allclose(S, m["S_exact"], atol=1e-9)
        np.testing.assert_allclose(Ps[ts], m["Ps_exact"], atol=1e-9)
        # against the reference: 1e-9 where it is well defined; on m0 the reference's own truncation
        # error (9e-8 vs exact, asserted in tests/test_oracle_golden.py) is the tolerance: 2e-7
        np.testing.assert_allclose(S, m["S"], atol=2e-7 if i == 0 else 1e-9)
        np.testing.assert_allclose(Ps[ts], m["Ps"], atol=1e-9)
        # projection epilogue on the same records: finite, variances >= 0
        p = kf.simulate_smoothed(m["phi"][None], m["q"][None], P0=m["P0"][None])
        assert np.isfinite(_np(p["sim_means"])).all() and (_np(p["sim_vars"]) >= 0).all()


def test_adapter_raises_on_nonpositive_innovation_variance(g2):
    """f <= 0: the reference would silently return NaN/inf (division, log); the adapters raise."""
    from metran_amd.kalmanfilter import MetranHipError, SPKalmanFilter, seqkalmanfilter_hip
    import pandas as pd

    y = g2["obs"][:50]
    o, oi, oc = oracle.set_observations(y)
    Phi, Q, Z = np.diag(g2["phi"]), np.diag(g2["q"]), observation_matrix(g2["loadings"])
    R_bad = np.array([-10.0, 0.0])  # f = R_0 + Z_0 P Z_0^T < 0
    with pytest.raises(MetranHipError, match="innovation variance"):
        seqkalmanfilter_hip(o, Phi, Q, Z, R_bad, oi, oc, np.zeros(3), np.eye(3))
    kf = SPKalmanFilter(engine="hip")
    kf.set_observations(pd.DataFrame(y))
    kf.set_matrices(Phi, Q, Z, R_bad)
    with pytest.raises(Exception, match="innovation variance"):
        kf.run_filter()
    with pytest.raises(Exception, match="innovation variance"):
        kf.run_smoother()
    kf.set_matrices(Phi, Q, Z, np.zeros(2))
    kf.run_smoother()  # and works again with valid matrices
    assert np.isfinite(kf.smoothed_state_means).all()


def test_adapter_raises_on_indefinite_covariance():
    from metran_amd.kalmanfilter import MetranHipError, kalmansmoother_hip

    T, n = 6, 3
    phi = np.array([0.9, 0.8, 0.7])
    F = np.zeros((T, n))
    Pf = np.tile(-np.eye(n), (T, 1, 1))                  # not a covariance
    Pp = np.tile(np.diag(-phi * phi + 0.01), (T, 1, 1))  # Phi Pf Phi + Q with q = 0.01: negative pivots
    with pytest.raises(MetranHipError, match="indefinite"):
        kalmansmoother_hip(F, Pf, F.copy(), Pp, np.diag(phi))


def test_five_arg_smoother_without_a_preceding_filter_call():
    """Arrays of unknown origin (copies): q is recovered from the step with the least cancellation; checked
    on the model with q = 2e-8 for the common factor (heywood m1), where q = diag(Pp[1]) - phi^2 diag(Pf[0])
    at step 0 (P0 = I) would lose 8 of 16 digits."""
    from metran_amd.kalmanfilter import kalmansmoother_hip, seqkalmanfilter_hip

    m = dict(golden_models("heywood.npz"))[1]
    y = m["obs"][:300]
    n = m["phi"].shape[0]
    o, oi, oc = oracle.set_observations(y)
    Phi, Q, Z = np.diag(m["phi"]), np.diag(m["q"]), observation_matrix(m["loadings"])
    ref = oracle.seqkalmanfilter(o, Phi, Q, Z, np.zeros(8), oi, oc, np.zeros(n), np.eye(n))
    S_ref, Ps_ref = oracle.kalmansmoother(ref[3], ref[4], ref[5], ref[6], Phi)
    S, Ps = kalmansmoother_hip(ref[3].copy(), ref[4].copy(), ref[5].copy(), ref[6].copy(), Phi)
    np.testing.assert_allclose(S, S_ref, atol=1e-8)
    np.testing.assert_allclose(Ps, Ps_ref, atol=1e-8)
    # the reference's own call sequence (run_smoother, :676-694): q remembered from the engine call, no cancellation
    res = seqkalmanfilter_hip(o, Phi, Q, Z, np.zeros(8), oi, oc, np.zeros(n), np.eye(n))
    S2, Ps2 = kalmansmoother_hip(res[3], res[4], res[5], res[6], Phi)
    np.testing.assert_allclose(S2, S_ref, atol=1e-9)
    np.testing.assert_allclose(Ps2, Ps_ref, atol=1e-9)
