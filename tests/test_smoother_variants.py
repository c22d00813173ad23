"""GPU: the alternative n <= 15 smoother, smoother_blk_kernel (the two n^3 products as 4x4x4 f64 MFMA blocks, the mean as
column n of the covariance tile; selected with ``mk_set_kernel_variant(ctx, MK_VARIANT_SMOOTHER16, 1)``), gives the oracle's
numbers in every mode the default smoother_record_kernel serves: full-square and packed-symmetric records, projection and
variance epilogues, a partial last workgroup, missing data, several state dimensions.  Likewise the round-1 wide smoother
(``MK_VARIANT_WIDE_SMOOTHER``).  And: no environment variable selects a kernel or changes a result any more (round-2
verdict, weak 2) -- ``MK_WIDE_TUNE=7`` used to make ``mk_smooth`` skip the factorisation with status 0."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def npy(t):
    return t.detach().cpu().numpy()


def _check_all_modes(variant):
    import oracle
    from metran_amd.engine import BatchedKalman
    from metran_amd.synthetic import make_dfm_batch

    for (N, K, T, B, missing) in [(8, 2, 60, 37, 0.2), (5, 1, 33, 9, 0.4), (2, 1, 17, 5, 0.0), (6, 2, 21, 8, 0.1), (4, 1, 2, 3, 0.0), (3, 1, 1, 2, 0.0)]:
        n = N + K
        d = make_dfm_batch(B, N, K, T, seed=4242 + N, missing=missing, first_step="random")
        ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
        Z = np.concatenate([np.broadcast_to(np.eye(N), (B, N, N)), d["loadings"]], axis=2)
        for layout in ("time_major", "model_major"):
            for sym in (False, True):
                kf = BatchedKalman(layout=layout, packed_sym=sym).set_variant("smoother16", variant)
                assert kf.get_variant("smoother16") == variant
                kf.set_observations(d["obs"]).set_loadings(d["loadings"])
                r = kf.filter_smooth(d["phi"], d["q"])
                Ps = npy(kf.unpack_sym(r["Ps"])) if sym else npy(r["Ps"])
                np.testing.assert_allclose(npy(r["S"]), ref["S"], atol=1e-9)
                np.testing.assert_allclose(Ps, ref["Ps"], atol=1e-9)
                assert int(npy(r["status"]).sum()) == 0
                p = kf.simulate_smoothed(d["phi"], d["q"])
                np.testing.assert_allclose(npy(p["sim_means"]), np.einsum("bjn,btn->btj", Z, ref["S"]), atol=1e-9)
                np.testing.assert_allclose(npy(p["sim_vars"]), np.maximum(np.einsum("bjn,btnm,bjm->btj", Z, ref["Ps"], Z), 0), atol=1e-9)
                v = kf.smooth_state_variances(d["phi"], d["q"])
                np.testing.assert_allclose(npy(v["S"]), ref["S"], atol=1e-9)
                np.testing.assert_allclose(npy(v["var"]), np.einsum("btnn->btn", ref["Ps"]), atol=1e-9)


@pytest.mark.parametrize("variant", ["blk", "record"])
def test_smoother_variant_matches_the_oracle(variant):
    _check_all_modes(variant)


def test_wide_smoother_variants_agree_with_the_oracle():
    import oracle
    from metran_amd.engine import BatchedKalman
    from metran_amd.synthetic import make_dfm_batch

    d = make_dfm_batch(5, 14, 3, 40, seed=1717, missing=0.2, first_step="random")
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    for variant in ("mfma", "v1", "mfma_unfolded"):
        kf = BatchedKalman().set_variant("wide_smoother", variant)
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        r = kf.filter_smooth(d["phi"], d["q"])
        np.testing.assert_allclose(npy(r["S"]), ref["S"], atol=1e-9)
        np.testing.assert_allclose(npy(r["Ps"]), ref["Ps"], atol=1e-9)


@pytest.mark.parametrize("N,K", [(32, 4), (14, 3)])
def test_folded_wide_smoother_is_bit_identical_to_the_unfolded_one(N, K):
    """The lane fold (rows of A in the lanes the model does not use, one multiply-add per (c, k) of the fused
    factorisation / forward sweep) reorders nothing within a row: every output of every mode -- records (full-square and
    packed-symmetric), projection, state variances -- equals the unfolded kernel's BIT FOR BIT, with missing data, an
    empty first step, several wavefronts and a batch that is not a multiple of anything; and both equal the oracle."""
    import torch

    import oracle
    from metran_amd.engine import BatchedKalman
    from metran_amd.synthetic import make_dfm_batch

    B, T = 7, 45
    d = make_dfm_batch(B, N, K, T, seed=3600 + N, missing=0.3, first_step="random")
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    for sym in (False, True):
        out = {}
        for variant in ("mfma", "mfma_unfolded"):
            kf = BatchedKalman(packed_sym=sym).set_variant("wide_smoother", variant)
            assert kf.get_variant("wide_smoother") == variant
            kf.set_observations(d["obs"]).set_loadings(d["loadings"])
            r = kf.filter_smooth(d["phi"], d["q"])
            p = kf.simulate_smoothed(d["phi"], d["q"])
            v = kf.smooth_state_variances(d["phi"], d["q"])
            out[variant] = [r["S"].clone(), r["Ps"].clone(), p["sim_means"].clone(), p["sim_vars"].clone(), v["S"].clone(),
                            v["var"].clone(), r["status"].clone()]
            Ps = npy(kf.unpack_sym(r["Ps"])) if sym else npy(r["Ps"])
            np.testing.assert_allclose(npy(r["S"]), ref["S"], atol=1e-9)
            np.testing.assert_allclose(Ps, ref["Ps"], atol=1e-9)
            np.testing.assert_allclose(npy(v["var"]), np.einsum("btnn->btn", ref["Ps"]), atol=1e-9)
            assert int(npy(r["status"]).sum()) == 0
        for x, y in zip(out["mfma"], out["mfma_unfolded"]):
            assert torch.equal(x, y)


ENV_SCRIPT = r'''
import numpy as np
import oracle
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch
for (N, K) in ((14, 3), (32, 4), (8, 2)):
    d = make_dfm_batch(3, N, K, 30, seed=99 + N, missing=0.3)
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    kf = BatchedKalman()
    assert kf.get_variant("smoother16") == "record" and kf.get_variant("wide_smoother") == "mfma"
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    r = kf.filter_smooth(d["phi"], d["q"])
    np.testing.assert_allclose(r["S"].cpu().numpy(), ref["S"], atol=1e-9)
    np.testing.assert_allclose(r["Ps"].cpu().numpy(), ref["Ps"], atol=1e-9)
    assert int(r["status"].abs().sum().item()) == 0
print("env ignored ok")
'''


def test_no_environment_variable_changes_a_result():
    """MK_WIDE_TUNE=7 (formerly: skip the factorisation and both products, status 0), MK_WIDE_SMOOTHER, MK_SMOOTHER16 are
    not read by the library: same oracle-equal output, default variants reported."""
    env = dict(os.environ, MK_WIDE_TUNE="7", MK_WIDE_SMOOTHER="v1", MK_SMOOTHER16="blk", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", ENV_SCRIPT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "env ignored ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
