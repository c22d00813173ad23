"""Dev probe: kernel time of configs[1] as a function of how long the GPU has been busy (clock / power-state ramp)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch

B = int(os.environ.get("B", "4096"))
d = make_dfm_batch_torch(B, 8, 2, 1000, seed=2000, device=torch.device("cuda:0"), missing=0.0)
kf = BatchedKalman(0, layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
bufs = kf._alloc_outputs(B, ["F", "Pf", "Xp", "Pp", "S", "Ps"])
torch.cuda.synchronize()
time.sleep(float(os.environ.get("IDLE", "2")))
kf.enable_timing(True, accumulate=True)
t0 = time.perf_counter()
for blk in range(30):
    for _ in range(10):
        kf.filter_smooth(d["phi"], d["q"], buffers=bufs)
    f, nf, s, ns = kf.kernel_ms_totals()
    print("launches %3d-%3d  t=%.3f s  filter %.3f ms  smoother %.3f ms" % (10 * blk, 10 * blk + 9, time.perf_counter() - t0, f / nf, s / ns))
