import json, sys, torch
sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
T, B = 2000, 4096
d = make_dfm_batch_torch(B, 32, 4, T, seed=4000, device=torch.device("cuda", 0), missing=0.3)
for layout in ("time_major", "model_major", "time_major", "model_major"):
    for state in (False, True):
        kf = BatchedKalman(layout=layout)
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        if state:
            bufs = kf.alloc_state_variances(B); run = lambda: kf.smooth_state_variances(d["phi"], d["q"], buffers=bufs)
        else:
            bufs = kf.alloc_projection(B); run = lambda: kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
        run(); torch.cuda.synchronize()
        kf.enable_timing(True, accumulate=True)
        for _ in range(4): run()
        torch.cuda.synchronize()
        f, fn, s, sn = kf.kernel_ms_totals()
        print(layout, "state" if state else "projection", "filter %.2f smoother %.2f -> %.0f models/s" % (f / fn, s / sn, B / ((f / fn + s / sn) / 1e3)), float(bufs["mle"].sum()), flush=True)
        del kf, bufs; torch.cuda.empty_cache()
