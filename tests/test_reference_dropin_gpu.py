"""The REAL reference class on the GPU through the drop-in boundary.

The unmodified ``metran`` package (the mounted reference in the build container, its verbatim staging copy
``oracle/_ref`` on the GPU box: ``oracle/make_ref.sh``) is imported through the pastas stub of
``tests/golden/_refshim.py``, ``metran_amd.kalmanfilter.install`` replaces the three module globals of
INTEGRATION.md section 2, and the bodies of the reference's own integration tests
(/root/reference/tests/test_metran.py:4-40, which only assert "it runs") are replayed WITH values asserted
against fixtures generated from the reference's CPU engines (tests/golden/g1_real.npz, g1_solve.npz).
Test infrastructure only: ``metran_amd`` never imports anything under ``oracle/`` or ``tests/``.
"""
import glob
import os
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import load_golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import _refshim  # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _refshim.reference_available(),
                                 reason="no reference: run oracle/make_ref.sh where /root/reference is mounted")]

ASTAR = [5.501017, 13.560042, 4.682870, 11.381674, 13.140605, 22.980925]  # notebook optimum, g1_real's point


@pytest.fixture(scope="module")
def metran():
    import metran_amd.kalmanfilter as hip

    m = _refshim.install()
    hip.install(m)
    yield m
    hip.uninstall(m)


def _series_list():
    """tests/conftest.py:13-24 of the reference (its ``series_list`` fixture), sorted for determinism."""
    files = sorted(glob.glob(os.path.join(_refshim.REFERENCE_ROOT, "examples", "data", "*_res.csv")))
    assert len(files) == 5
    out = []
    for f in files:
        s = pd.read_csv(f, header=0, index_col=0, parse_dates=True).squeeze()
        s.name = os.path.basename(f).split("_")[0]
        out.append(s)
    return out


@pytest.fixture
def mt_init(metran):
    return metran.Metran(_series_list(), name="B21B0214")


@pytest.fixture(scope="module")
def mt(metran):
    """Solved model (the reference's ``mt`` fixture): ScipySolve driving the HIP engine."""
    m = metran.Metran(_series_list(), name="B21B0214")
    m.solve(report=False)
    return m


def _p(mt):
    return pd.Series(ASTAR, index=mt.parameters.index)


def test_engine_is_the_hip_callable(metran, mt):
    import metran_amd.kalmanfilter as hip

    assert mt.kf.filtermethod is hip.seqkalmanfilter_hip
    assert metran.kalmanfilter.kalmansmoother is hip.kalmansmoother_hip


def test_metran_solve_scipy(mt):
    """test_metran.py:4-5 -- every get_mle() of scipy's L-BFGS-B ran filter_kernel (B = 1)."""
    gs = load_golden("g1_solve.npz")
    assert abs(mt.fit.obj_func - 2332.3270693771483) < 1e-5
    assert abs(mt.fit.obj_func - float(gs["obj"])) < 1e-5
    assert abs(int(mt.fit.nfev) - 77) <= 14            # the reference's own count (notebook :143), +- 2 iterations
    np.testing.assert_allclose(mt.parameters["optimal"].values.astype(float), gs["optimal"], rtol=2e-3)
    assert round(mt.fit.aic, 2) == 2344.33
    rep = mt.fit_report()
    assert "ScipySolve" in rep and "2332.33" in rep and "nfev" in rep


def test_metran_solve_hipsolve(metran, mt_init):
    """Plug point A with the real class: Metran.solve(solver=HipSolve) -> fit_report renders, accessors work."""
    from metran_amd.solver import HipSolve, HipSolveAdjoint

    gs = load_golden("g1_solve.npz")
    mt_init.solve(solver=HipSolve, report=False)
    assert mt_init.settings["solver"] == "HipSolve"
    assert abs(mt_init.fit.obj_func - float(gs["obj"])) < 1e-5
    np.testing.assert_allclose(mt_init.parameters["optimal"].values.astype(float), gs["optimal"], rtol=2e-3)
    rep = mt_init.fit_report()
    assert "HipSolve" in rep and "2332.3" in rep and "nfev" in rep
    assert isinstance(mt_init.metran_report(), str)
    # default-p accessors right after the solve (mt.kf must hold the optimum's matrices)
    sm = mt_init.get_state_means()
    assert sm.shape == (6255, 6) and np.isfinite(sm.values).all()
    sim = mt_init.get_simulation("B21B0214005")
    assert list(sim.columns) == ["mean", "lower", "upper"]
    # ... and they are the values at the optimum
    p = pd.Series(mt_init.parameters["optimal"].values.astype(float), index=mt_init.parameters.index)
    np.testing.assert_allclose(sm.values, mt_init.get_state_means(p=p).values, atol=1e-9)
    # the adjoint-gradient class through the same issubclass() gate (metran.py:1033), on a fresh model: the
    # reference cannot solve() one instance twice under pandas 2 (set_init_parameters assigns 5-tuples to rows
    # of a parameter table that has gained the 'optimal' and 'stderr' columns, metran.py:448)
    mt2 = metran.Metran(_series_list(), name="B21B0214")
    mt2.solve(solver=HipSolveAdjoint, report=False)
    assert mt2.settings["solver"] == "HipSolveAdjoint"
    assert abs(mt2.fit.obj_func - float(gs["obj"])) < 1e-5
    assert mt2.fit.nfev < 40
    assert np.isfinite(mt2.get_state_means().values).all()


def test_metran_state_means(mt, g1):
    """test_metran.py:12-13 at the fixture's parameter point."""
    sm = mt.get_state_means(p=_p(mt))
    assert list(sm.columns) == ["B21B021400%d_sdf" % i for i in range(1, 6)] + ["cdf1"]
    np.testing.assert_allclose(sm.values[:5], g1["state_means_head"], atol=1e-9)
    np.testing.assert_allclose(sm.values[-5:], g1["state_means_tail"], atol=1e-9)
    np.testing.assert_allclose(sm.values, g1["S"], atol=1e-9)
    # stored notebook rows, examples/metran_practical_example.ipynb:395-427
    np.testing.assert_allclose(sm.values[0], [0.226549, 0.021665, 0.028548, 0.026005, 0.153683, 0.809228], atol=6e-7)
    filt = mt.get_state_means(p=_p(mt), method="filter")
    np.testing.assert_allclose(filt.values, g1["F"], atol=1e-10)


def test_metran_simulated_means(mt, g1):
    """test_metran.py:16-17"""
    sim = mt.get_simulated_means(p=_p(mt))
    np.testing.assert_allclose(sim.values, g1["sim_means"] + g1["oseries_mean"], atol=1e-9)
    var = mt.get_simulated_variances(p=_p(mt))
    np.testing.assert_allclose(var.values, g1["sim_vars"], atol=1e-9)


def test_metran_get_simulation(mt, g1):
    """test_metran.py:20-21"""
    sim = mt.get_simulation("B21B0214005", p=_p(mt), alpha=0.05)
    np.testing.assert_allclose(sim.values[:50], g1["get_simulation_005"], atol=1e-9)


def test_metran_decompose_simulation(mt, g1):
    """test_metran.py:24-25"""
    dec = mt.decompose_simulation("B21B0214001", p=_p(mt))
    assert list(dec.columns) == ["sdf", "cdf1"]
    np.testing.assert_allclose(dec.values[:50], g1["decompose_001"], atol=1e-9)


def test_metran_get_state(mt, g1):
    """test_metran.py:28-29: state 0 with its confidence band from the smoothed variances."""
    from scipy.stats import norm

    st = mt.get_state(0, p=_p(mt))
    assert list(st.columns) == ["mean", "lower", "upper"]
    np.testing.assert_allclose(st["mean"].values, g1["S"][:, 0], atol=1e-9)
    half = norm.ppf(0.975) * np.sqrt(g1["Ps"][:, 0, 0])
    np.testing.assert_allclose((st["upper"] - st["mean"]).values[g1["tsel"]], half, atol=1e-9)


def test_metran_masked_oseries(mt, g1):
    """test_metran.py:32-40, with the reference's masked projection asserted."""
    p = _p(mt)
    proj1 = mt.get_simulation("B21B0214005", p=p)
    oseries = mt.get_observations()
    mask = (0 * oseries).astype(bool)
    mask.loc["1997-8-28", "B21B0214005"] = True
    mt.mask_observations(mask)
    try:
        proj2 = mt.get_simulation("B21B0214005", p=p)
        masked = mt.get_simulation("B21B0214005", p=p, alpha=None)
        np.testing.assert_allclose(masked.values.ravel(), g1["masked_sim_005"].ravel(), atol=1e-8)
        assert abs(mt.get_mle(p) - float(g1["masked_mle_star"])) < 1e-9 * 2332
    finally:
        mt.unmask_observations()
    assert (proj1 != proj2).any().any()
    assert abs(mt.get_mle(p) - 2332.327069381027) < 1e-9 * 2332


def test_metran_get_factors_on_the_device(metran, mt_init):
    """Row f4 through the reference class: Metran.get_factors (metran.py:199-226) constructs FactorAnalysis by its
    module-global name; with metran_amd.factoranalysis.install it is the MI355X one.  BASELINE.md G1e."""
    import metran_amd.factoranalysis as hfa

    hfa.install(metran)
    try:
        f = mt_init.get_factors()
        assert isinstance(metran.metran.FactorAnalysis(), hfa.FactorAnalysis)
    finally:
        hfa.uninstall(metran)
    assert f.shape == (5, 1) and mt_init.nfactors == 1
    np.testing.assert_allclose(f.ravel(), [0.85798172, 0.93587365, 0.96619738, 0.95779419, 0.90085667], atol=1e-8)
    np.testing.assert_allclose(mt_init.eigval, [4.41619129, 0.30437694, 0.13506397, 0.08427842, 0.06008939], atol=1e-8)
    assert abs(mt_init.fep - 88.32382575015878) < 1e-8
    assert not isinstance(metran.metran.FactorAnalysis(), hfa.FactorAnalysis)
