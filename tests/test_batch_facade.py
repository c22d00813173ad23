"""MetranBatch: the accessors of metran.Metran (metran/metran.py:605-989) for several models at once,
against the goldens produced by the reference itself on examples/data (tests/golden/make_golden.py)."""
import numpy as np
import pandas as pd
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _g1_series(g1):
    idx = pd.DatetimeIndex(g1["index_ns"].astype("datetime64[ns]"))
    raw = g1["obs"] * g1["oseries_std"] + g1["oseries_mean"]
    return [pd.Series(raw[:, j], index=idx, name="B21B021400%d" % (j + 1)).dropna() for j in range(raw.shape[1])]


def test_facade_on_examples_data(g1):
    from metran_amd.batch import MetranBatch

    gs = load_golden("g1_solve.npz")
    series = _g1_series(g1)
    short = [s.iloc[: len(s) // 2] for s in series]           # a second, shorter model: padded record
    mb = MetranBatch([series, short], factors=g1["loadings"])
    assert (mb.R, mb.T, mb.N, mb.K) == (2, 6255, 5, 1)
    astar = np.stack([g1["alpha_star"], g1["alpha_star"]])
    mle = mb.get_mle(astar).cpu().numpy()
    assert abs(mle[0] - 2332.327069381027) < 1e-6            # BASELINE.md G1 (standardisation done on the device)
    # get_simulation == the reference's (first 50 rows, mean/lower/upper), both methods of projection
    sim = mb.get_simulation(0, "B21B0214005", alpha=astar, ci=0.05)
    np.testing.assert_allclose(sim.values[:50, 0], g1["get_simulation_005"][:, 0], rtol=0, atol=1e-8)
    # the bounds carry sqrt(variance): at observed dates the variance is 0 up to ~1e-15 of rounding, whose
    # square root is ~5e-8
    np.testing.assert_allclose(sim.values[:50, 1:], g1["get_simulation_005"][:, 1:], rtol=0, atol=5e-7)
    assert list(sim.columns) == ["mean", "lower", "upper"] and sim.shape[0] == 6255
    m = mb.get_simulated_means(astar)[0].cpu().numpy()
    np.testing.assert_allclose(m, g1["sim_means"] + g1["oseries_mean"], rtol=0, atol=1e-8)
    v = mb.get_simulated_variances(astar)[0].cpu().numpy()
    np.testing.assert_allclose(v, g1["sim_vars"], rtol=0, atol=1e-8)
    mf = mb.get_simulated_means(astar, method="filter")[0].cpu().numpy()
    np.testing.assert_allclose(mf[g1["tsel"]], g1["simf_means"] + g1["oseries_mean"], rtol=0, atol=1e-8)
    vf = mb.get_simulated_variances(astar, method="filter")[0].cpu().numpy()
    np.testing.assert_allclose(vf[g1["tsel"]], g1["simf_vars"], rtol=0, atol=1e-8)
    # smoothed states with the reference's column names
    st = mb.get_state_means(0, astar)
    assert list(st.columns) == ["B21B021400%d_sdf" % i for i in range(1, 6)] + ["cdf1"]
    np.testing.assert_allclose(st.values[:5], g1["state_means_head"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(st.values[-5:], g1["state_means_tail"], rtol=0, atol=1e-8)
    # decompose_simulation / get_state_variances / get_state (metran.py:682-756, 885-942) vs the reference's
    dec = mb.decompose_simulation(0, "B21B0214001", alpha=astar)
    assert list(dec.columns) == ["sdf", "cdf1"]
    np.testing.assert_allclose(dec.values[:50], g1["decompose_001"], rtol=0, atol=1e-8)
    sv = mb.get_state_variances(0, astar)
    assert list(sv.columns) == list(st.columns)
    np.testing.assert_allclose(sv.values[g1["tsel"]], np.diagonal(g1["Ps"], axis1=1, axis2=2), rtol=0, atol=1e-8)
    s0 = mb.get_state(0, 0, astar)
    from scipy.stats import norm

    assert list(s0.columns) == ["mean", "lower", "upper"]
    np.testing.assert_allclose(s0["mean"].values, g1["S"][:, 0], rtol=0, atol=1e-8)
    np.testing.assert_allclose((s0["upper"] - s0["mean"]).values[g1["tsel"]], norm.ppf(0.975) * np.sqrt(g1["Ps"][:, 0, 0]),
                               rtol=0, atol=1e-7)
    ff = mb.get_state_means(0, astar, method="filter")
    np.testing.assert_allclose(ff.values, g1["F"], rtol=0, atol=1e-8)
    # the result cache (metran.py:978-989): the accessors above launched one run per kind for this parameter set
    assert set(mb._cache) == {"project", "smoother", "filter"}
    before = {k: id(v[1]) for k, v in mb._cache.items()}
    mb.get_simulation(0, "B21B0214002", alpha=astar), mb.get_state(0, 3, astar), mb.decompose_simulation(1, "B21B0214003", astar)
    assert {k: id(v[1]) for k, v in mb._cache.items()} == before
    mb.get_state_means(0, astar * 1.01)   # another parameter set: recomputed
    assert id(mb._cache["smoother"][1]) != before["smoother"]
    # masking (metran.py:464-506): the projection at the masked date changes and the cache is dropped
    import torch

    mask = torch.zeros((2, 6255, 5), dtype=torch.uint8)
    mask[0, int(g1["mask_t"]), 4] = 1
    mb.mask_observations(mask)
    msim = mb.get_simulation(0, "B21B0214005", alpha=astar, ci=None)
    np.testing.assert_allclose(msim.values, g1["masked_sim_005"].ravel(), rtol=0, atol=1e-7)
    mb.unmask_observations()
    np.testing.assert_allclose(mb.get_simulation(0, "B21B0214005", alpha=astar, ci=None).values[:50],
                               g1["get_simulation_005"][:, 0], rtol=0, atol=1e-8)
    # the padded model returns frames of its own length
    assert mb.get_simulation(1, "B21B0214001", alpha=astar).shape[0] == int(mb.batch.lengths[1]) < 6255
    with pytest.raises(KeyError, match="Unknown name"):
        mb.get_simulation(0, "nope", alpha=astar)
    # solve() reaches the reference optimum for model 0 (Metran.solve on the same data: obj 2332.327, nfev 77)
    fit = mb.solve(stderr=True)
    assert bool(fit.converged.all())
    assert abs(float(fit.obj[0]) - float(gs["obj"])) < 1e-4
    np.testing.assert_allclose(fit.alpha[0].cpu().numpy(), gs["optimal"], rtol=1e-2)
    sim2 = mb.get_simulation(0, "B21B0214005")                # default parameters = the optimum just found
    np.testing.assert_allclose(sim2.values[:50], g1["get_simulation_005"], rtol=0, atol=2e-3)
