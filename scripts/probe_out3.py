"""Dev probe: configs[1] shape, filter writing the filtered record only (OUT=3) + projecting / variance smoothers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch

for B in (4096, 8192):
    d = make_dfm_batch_torch(B, 8, 2, 1000, seed=2000, device=torch.device("cuda:0"), missing=0.0)
    kf = BatchedKalman(0, layout="time_major")
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    bufs = kf.alloc_projection(B)
    kf.enable_timing(True, accumulate=True)
    for name, fn in (("project", lambda: kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)),
                     ("var_only", lambda: kf.smooth_state_variances(d["phi"], d["q"]))):
        for _ in range(3):
            fn()
        kf.kernel_ms_totals()
        for _ in range(10):
            fn()
        f, nf, s, ns = kf.kernel_ms_totals()
        print("B %d %-9s filter(OUT=3) %.3f ms  smoother %.3f ms" % (B, name, f / nf, s / ns))
