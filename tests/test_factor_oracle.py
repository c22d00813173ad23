"""Pins oracle/factor_oracle.py (numpy restatement of metran/factoranalysis.py) to the reference through
tests/golden/factor_analysis.npz, which ``make_golden.py factor_analysis`` generated from the reference itself.
Also asserts the known answers the reference publishes: tests/test_factoranalysis.py:10,17,24 (eigenvalues
[1.8, 0.2], MAP test -> 1 factor, loadings shape (5, 1)), BASELINE.md G1e (loadings 0.857982 ...) and G2
(0.93540765, eigenvalues [1.87212635, 0.12787365])."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import factor_oracle as fo

CASES = ["g1", "g2", "s8k2", "s12k3", "s6k1", "s20k4", "weak"]


@pytest.fixture(scope="module")
def fa():
    return load_golden("factor_analysis.npz")


def match_columns(f, ref):
    """Loadings agree up to the order of the columns (varimax does not define one); signs are fixed by the
    reference's dominant-sign convention."""
    assert f.shape == ref.shape
    used, out = set(), np.empty_like(ref)
    for j in range(ref.shape[1]):
        k = min((k for k in range(f.shape[1]) if k not in used), key=lambda k: np.abs(f[:, k] - ref[:, j]).max())
        used.add(k)
        out[:, j] = f[:, k]
    return out


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference(fa, name):
    y = fa[name + "_obs"]
    r = fo.solve(y)
    np.testing.assert_allclose(r["corr"], fa[name + "_corr"], atol=1e-13)
    np.testing.assert_allclose(r["eigval"], fa[name + "_eigval"], atol=1e-12)
    assert (r["nfactors_map"], r["nfactors_map4"]) == (int(fa[name + "_nfactors_map"]), int(fa[name + "_nfactors_map4"]))
    assert r["nfactors"] == int(fa[name + "_nfactors"])
    np.testing.assert_allclose(r["psi0"], fa[name + "_psi0"], atol=1e-12)
    np.testing.assert_allclose(fo.minresfun(fa[name + "_psi0"], r["corr"], r["nfactors"]), float(fa[name + "_fun0"]), rtol=1e-10)
    np.testing.assert_allclose(fo.minresgrad(fa[name + "_psi0"], r["corr"], r["nfactors"]), fa[name + "_grad0"], rtol=1e-8,
                               atol=1e-10)
    # where L-BFGS-B moved (g2, s6k1) its path is reproduced because it IS the same scipy routine on the same f, g
    np.testing.assert_allclose(r["psi"], fa[name + "_psi"], atol=1e-8)
    np.testing.assert_allclose(match_columns(r["factors"], fa[name + "_factors"]), fa[name + "_factors"], atol=1e-8)
    assert abs(r["fep"] - float(fa[name + "_fep"])) < 1e-9


def test_lbfgsb_returns_its_start_vector_except_twice(fa):
    moved = {n: float(np.abs(fa[n + "_psi"] - fa[n + "_psi0"]).max()) for n in CASES}
    assert {n for n, m in moved.items() if m > 0} == {"g2", "s6k1"}
    assert all(str(fa[n + "_message"]).startswith("ABNORMAL") and int(fa[n + "_nit"]) == 0 for n in CASES
               if moved[n] == 0)


def test_published_known_answers(fa):
    w, v = fo.get_eigval(fa["unit_corr"])
    np.testing.assert_allclose(w, [1.8, 0.2])                              # tests/test_factoranalysis.py:10
    assert fo.maptest(fa["unit_corr"], v, w)[0] == 1 == int(fa["unit_maptest"][0])   # :17
    assert fa["g1_factors"].shape == (5, 1)                                # :24
    np.testing.assert_allclose(fa["g1_factors"].ravel(), [0.857982, 0.935874, 0.966197, 0.957794, 0.900857], atol=5e-7)
    np.testing.assert_allclose(fa["g2_factors"].ravel(), [0.93540765, 0.93540765], atol=5e-9)
    np.testing.assert_allclose(fa["g2_eigval"], [1.87212635, 0.12787365], atol=5e-9)
    assert abs(float(fa["g1_fep"]) - 88.32) < 5e-3                         # notebook: fep 88.32%


# ---- multi-factor models (tests/golden/factor_multi.npz): columns in the reference's OWN order, no matching ----
def multi_names():
    return [str(n) for n in load_golden("factor_multi.npz")["names"]]


@pytest.fixture(scope="module")
def fm():
    return load_golden("factor_multi.npz")


def test_multi_fixture_has_what_the_verdict_asked_for(fm):
    names = multi_names()
    assert len(names) >= 40
    nondominant = [n for n in names if sorted(fm[n + "_eig_rank"]) != list(range(int(fm[n + "_nfactors"])))]
    permuted = [n for n in names if n not in nondominant and list(fm[n + "_eig_rank"]) != list(range(int(fm[n + "_nfactors"])))]
    assert len(nondominant) >= 8 and len(permuted) >= 4
    assert all(int(fm[n + "_nfactors"]) >= 2 for n in names)
    assert {fm[n + "_corr"].shape[0] for n in names} >= {20, 32}
    assert sum(float(np.abs(fm[n + "_psi"] - fm[n + "_psi0"]).max()) > 0 for n in names) >= 3


@pytest.mark.parametrize("name", multi_names())
def test_oracle_reproduces_reference_multi(fm, name):
    r = fo.solve(corr=fm[name + "_corr"])
    nf = int(fm[name + "_nfactors"])
    assert r["nfactors"] == nf
    assert (r["nfactors_map"], r["nfactors_map4"]) == (int(fm[name + "_nfactors_map"]), int(fm[name + "_nfactors_map4"]))
    np.testing.assert_allclose(r["eigval"], fm[name + "_eigval"], atol=1e-12)
    np.testing.assert_allclose(r["psi0"], fm[name + "_psi0"], atol=1e-12)
    np.testing.assert_allclose(r["psi"], fm[name + "_psi"], atol=1e-9)
    ld = fo.get_loadings(fm[name + "_psi"], fm[name + "_corr"], nf)
    np.testing.assert_allclose(ld, fm[name + "_loadings_unrotated"], atol=1e-12)
    np.testing.assert_allclose(r["factors"], fm[name + "_factors"], atol=1e-9)      # same columns, same order, same signs
    sc = 1 / np.sqrt(fm[name + "_psi"])
    assert list(fo.eig_order(fm[name + "_corr"] * sc[:, None] * sc[None, :], nf)) == list(fm[name + "_eig_rank"])
    assert abs(r["fep"] - float(fm[name + "_fep"])) < 1e-9


def test_product_eig_order_is_the_oracles(fm):
    """Host logic of the product (no GPU): metran_amd.factoranalysis.eig_order, batched, instance b -> matrix b % R."""
    from metran_amd.factoranalysis import eig_order

    names = [n for n in multi_names() if fm[n + "_corr"].shape[0] == 32]
    corr = np.stack([fm[n + "_corr"] for n in names])
    psi = np.stack([fm[n + "_psi"] for n in names])
    got = eig_order(corr, np.concatenate([psi, psi]), 4)
    for b in range(2 * len(names)):
        n = names[b % len(names)]
        nf = int(fm[n + "_nfactors"])
        assert list(got[b, :nf]) == list(fm[n + "_eig_rank"])
