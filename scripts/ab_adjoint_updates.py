"""Same-box A/B of the wide adjoint gradient WITH and WITHOUT the update tape (BatchedKalman.adjoint_updates; C ABI
mk_set_adjoint_updates): kernel ms of the recording forward pass and of the backward walk at the flight sizes a wide calibration
runs at, and the largest difference between the two gradients.
  gpurun -- 'python scripts/ab_adjoint_updates.py'"""
import sys

import torch

sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

dev = torch.device("cuda", 0)
for (N, K, B, T) in ((32, 4, 1, 2000), (32, 4, 64, 500), (32, 4, 512, 500), (32, 4, 2048, 500), (14, 3, 512, 500)):
    d = make_dfm_batch_torch(B, N, K, T, seed=77, device=dev, missing=0.3)
    got = {}
    for rnd in range(2):
        for upd in (False, True):
            kf = BatchedKalman(layout="time_major")
            kf.adjoint_updates = upd
            kf.set_observations(d["obs"]).set_loadings(d["loadings"])
            kf.loglik_grad(d["phi"], d["q"])
            torch.cuda.synchronize()
            kf.enable_timing(True, accumulate=True)
            for _ in range(4):
                mle, gphi, gq = kf.loglik_grad(d["phi"], d["q"])
            torch.cuda.synchronize()
            f, fn, s, sn = kf.kernel_ms_totals()
            got[upd] = (mle.clone(), gphi.clone(), gq.clone())
            print("(%d,%d) B=%4d T=%4d  update tape %-5s round %d  forward %.2f ms  backward %.2f ms" % (N, K, B, T, upd, rnd, f / fn, s / sn), flush=True)
            kf.close()
    a, b = got[False], got[True]
    scale = max(float(a[1].abs().max()), float(a[2].abs().max()))
    print("   max |mle diff| %.1e   max |gradient diff| / max |gradient| %.1e" % (
        float((a[0] - b[0]).abs().max()), max(float((a[1] - b[1]).abs().max()), float((a[2] - b[2]).abs().max())) / scale), flush=True)
