import numpy as np, torch, sys, os
sys.path.insert(0, '.')
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
N, K, T = 32, 4, 60
dev = torch.device("cuda", 0)
print("library:", os.environ.get("METRAN_HIP_LIBRARY", "default"))
for B in (1024, 4096):
    d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=dev, missing=0.3)
    out = {}
    for variant in ("mfma", "mfma16"):
        kf = BatchedKalman(layout="time_major").set_variant("wide_smoother", variant)
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        r = kf.filter_smooth(d["phi"], d["q"]); torch.cuda.synchronize()
        kf.enable_timing(True, accumulate=True)
        for _ in range(3):
            r = kf.filter_smooth(d["phi"], d["q"])
        torch.cuda.synchronize()
        f_tot, f_n, s_tot, s_n = kf.kernel_ms_totals()
        out[variant] = r["Ps"].clone()
        print("  B %d %s smoother %.3f ms" % (B, variant, s_tot / s_n))
        del kf, r
    dP = (out["mfma"] - out["mfma16"]).abs()
    bad = (dP > 1e-10).any(3).any(2)
    print("B %5d: max dPs %.3e  bad (model,t) %d of %d" % (B, float(dP.max()), int(bad.sum()), bad.numel()))
    del out, d
    torch.cuda.empty_cache()
