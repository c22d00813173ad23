"""GPU: the alternative n <= 15 smoother, smoother_blk_kernel (the two n^3 products as 4x4x4 f64 MFMA blocks, the mean as
column n of the covariance tile; selected with MK_SMOOTHER16=blk), gives the oracle's numbers in every mode the default
smoother_record_kernel serves: full-square and packed-symmetric records, projection and variance epilogues, a partial
last workgroup, missing data, several state dimensions.  The selection is read once per process, hence the subprocess."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import numpy as np
import oracle
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch

def npy(t):
    return t.detach().cpu().numpy()

for (N, K, T, B, missing) in [(8, 2, 60, 37, 0.2), (5, 1, 33, 9, 0.4), (2, 1, 17, 5, 0.0), (6, 2, 21, 8, 0.1), (4, 1, 2, 3, 0.0), (3, 1, 1, 2, 0.0)]:
    n = N + K
    d = make_dfm_batch(B, N, K, T, seed=4242 + N, missing=missing, first_step="random")
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    Z = np.concatenate([np.broadcast_to(np.eye(N), (B, N, N)), d["loadings"]], axis=2)
    for layout in ("time_major", "model_major"):
        for sym in (False, True):
            kf = BatchedKalman(layout=layout, packed_sym=sym)
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
print("variants ok")
'''


@pytest.mark.parametrize("variant", ["blk", "record"])
def test_smoother_variant_matches_the_oracle(variant):
    env = dict(os.environ, MK_SMOOTHER16=variant, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "variants ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
