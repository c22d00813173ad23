"""Host logic of the batched factor analysis without a GPU: ``FactorAnalysisBatch._lockstep_minres`` (one public
``scipy.optimize.minimize(method="L-BFGS-B")`` per model on its own thread, every round of objective / jacobian requests
answered by ONE batched evaluation) must give, model by model, exactly what the reference's sequential call gives
(metran/factoranalysis.py:209-216).  The device evaluation is replaced by the numpy oracle's functions here; the GPU
tests (tests/test_factoranalysis_gpu.py) run the same code with ``mk_fa_minres`` behind it."""
import numpy as np
import pytest
import scipy.optimize as scopt
import torch

from conftest import load_golden
from metran_amd import factoranalysis as fa_mod
from oracle import factor_oracle as fo


class _CpuEngine:
    device = torch.device("cpu")


class _OracleBacked(fa_mod.FactorAnalysisBatch):
    """minres_eval served by the oracle (counts the batched launches)."""

    def __init__(self):
        self.maxfactors = None
        self.kf = _CpuEngine()
        self.launches = 0
        self.largest = 0

    def minres_eval(self, corr, nfactors, psi, kmax, want=("f", "g", "loadings"), order="lapack", corr_host=None):
        self.launches += 1
        psi = np.asarray(psi, dtype=np.float64)
        self.largest = max(self.largest, len(psi))
        c = corr.numpy()
        nf = nfactors.numpy()
        np.testing.assert_array_equal(c, corr_host)      # the host copy handed along is the device tensor's
        f = np.array([fo.minresfun(psi[b], c[b], int(nf[b])) for b in range(len(psi))])
        g = np.stack([fo.minresgrad(psi[b], c[b], int(nf[b])) for b in range(len(psi))])
        return torch.from_numpy(f), torch.from_numpy(g), None


def _cases():
    g = load_golden("factor_analysis.npz")
    m = load_golden("factor_multi.npz")
    out = [(g[n + "_corr"], int(g[n + "_nfactors"]), g[n + "_psi0"], g[n + "_psi"]) for n in ("g1", "s6k1", "s8k2", "weak")]
    out += [(m[n + "_corr"], int(m[n + "_nfactors"]), m[n + "_psi0"], m[n + "_psi"]) for n in ("mv1", "mv2", "mv3")]
    return out


def test_lockstep_equals_the_sequential_reference_call(monkeypatch):
    cases = [c for c in _cases() if c[0].shape[0] in (4, 5, 6)]
    by_n = {}
    for c in cases:
        by_n.setdefault(c[0].shape[0], []).append(c)
    for N, group in by_n.items():
        fb = _OracleBacked()
        monkeypatch.setattr(fa_mod, "_LOCKSTEP_THREADS", 2)          # several chunks as well
        corr = np.stack([c[0] for c in group])
        nf = torch.tensor([c[1] for c in group])
        start = np.stack([c[2] for c in group])
        x = fb._lockstep_minres(torch.from_numpy(corr), corr, nf, start, max(c[1] for c in group))
        for i, c in enumerate(group):
            ref = scopt.minimize(fo.minresfun, c[2], method="L-BFGS-B", jac=fo.minresgrad, bounds=[(0.005, 1)] * N,
                                 args=(c[0], c[1]))
            np.testing.assert_array_equal(x[i], ref.x)               # same routine, same numbers: bit for bit
            np.testing.assert_allclose(x[i], c[3], atol=1e-9)        # = what the reference recorded
        assert fb.largest <= 2


def test_one_launch_per_round_not_per_model():
    c = [c for c in _cases() if c[0].shape[0] == 4]
    assert len(c) >= 3
    M = 24
    fb = _OracleBacked()
    corr = np.stack([c[i % len(c)][0] for i in range(M)])
    nf = torch.tensor([c[i % len(c)][1] for i in range(M)])
    start = np.stack([c[i % len(c)][2] for i in range(M)])
    x = fb._lockstep_minres(torch.from_numpy(corr), corr, nf, start, 2)
    for i in range(M):
        np.testing.assert_allclose(x[i], c[i % len(c)][3], atol=1e-9)
    seq = 0
    for cc in c:
        seq = max(seq, scopt.minimize(fo.minresfun, cc[2], method="L-BFGS-B", jac=fo.minresgrad,
                                      bounds=[(0.005, 1)] * 4, args=(cc[0], cc[1])).nfev)
    assert fb.launches <= seq + 1 and fb.largest == M                # rounds = the longest model's evaluations


def test_a_failing_evaluation_is_raised_not_swallowed():
    class Boom(_OracleBacked):
        def minres_eval(self, *a, **k):
            raise RuntimeError("device evaluation failed")

    c = _cases()[0]
    fb = Boom()
    with pytest.raises(RuntimeError, match="device evaluation failed"):
        fb._lockstep_minres(torch.from_numpy(c[0][None]), c[0][None], torch.tensor([c[1]]), c[2][None], 1)
