"""Dev probe: C4 shape (32 series / 4 factors, 30 % missing) timing and parity on a sample."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
B, N, K, T = int(os.environ.get("B", 256)), 32, 4, int(os.environ.get("T", 500))
dev = torch.device("cuda", 0)
d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=dev, missing=0.3)
kf = BatchedKalman(0, layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
kf.enable_timing(True)
for outs in ([()] if os.environ.get("PROJ") else [(), ("F", "Pf", "Xp", "Pp", "S", "Ps")]):
    for i in range(3):
        if outs: r = kf.filter_smooth(d["phi"], d["q"], outputs=outs)
        else: r = {"mle": kf.loglik(d["phi"], d["q"])}
        f, s = kf.last_kernel_ms()
    print("outputs=%s  filter %.2f ms  smoother %.2f ms  -> %.0f models/s" % ("all" if outs else "loglik", f, max(s, 0), B / ((f + max(s, 0)) / 1e3)))
idx = [0, 1, B - 1]
if os.environ.get("PROJ"): r["S"] = r["Ps"] = None
ref = oracle.dfm_batch(*(d[k][idx].cpu().numpy() for k in ("obs", "phi", "q", "loadings")))
print("mle rel err", np.max(np.abs(r["mle"][idx].cpu().numpy() - ref["mle"]) / np.abs(ref["mle"])))
if r.get("S") is not None:
    print("S err", np.abs(r["S"][idx].cpu().numpy() - ref["S"]).max(), "Ps err", np.abs(r["Ps"][idx].cpu().numpy() - ref["Ps"]).max())
if os.environ.get("PROJ"):
    del r
    torch.cuda.empty_cache()
    for i in range(2):
        rr = kf.simulate_smoothed(d["phi"], d["q"])
        f, s = kf.last_kernel_ms()
        del rr
    print("projection path (filtered record only + fused simulate): filter %.2f ms  smoother %.2f ms -> %.0f models/s" % (f, s, B / ((f + s) / 1e3)))
