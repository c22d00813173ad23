"""Latency of the wide objective for FEW instances (what Metran.solve(solver=HipSolve) launches for one 32-series / 4-factor model:
P + 1 = 37 instances of one record): the lane-per-state filter against the split-layout one, per launch, hipEvents.
  gpurun -- 'python scripts/probe_small_wide.py'"""
import sys

import torch

sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

dev = torch.device("cuda", 0)
for (N, K, T) in ((32, 4, 2000), (20, 2, 2000)):
    d = make_dfm_batch_torch(1, N, K, T, seed=11, device=dev, missing=0.3)
    for B in (1, 37, 74, 512, 1024):
        phi = d["phi"].repeat(B, 1).contiguous()
        q = d["q"].repeat(B, 1).contiguous()
        ref = None
        for variant in ("lane_per_state", "split"):
            kf = BatchedKalman(0, layout="time_major").set_variant("wide_filter", variant)
            kf.set_observations(d["obs"]).set_loadings(d["loadings"])
            kf.loglik(phi, q)
            torch.cuda.synchronize()
            kf.enable_timing(True, accumulate=True)
            for _ in range(5):
                m = kf.loglik(phi, q)
            torch.cuda.synchronize()
            f, fn, _, _ = kf.kernel_ms_totals()
            ref = m.clone() if ref is None else ref
            print("(%d,%d) T=%d  B=%4d  %-15s %8.3f ms per launch   max rel diff %.1e" % (
                N, K, T, B, variant, f / fn, float(((m - ref).abs() / ref.abs()).max())), flush=True)
            kf.close()
