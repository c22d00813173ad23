import torch, sys, os
sys.path.insert(0, '.')
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
B, N, K, T = 4096, 32, 4, 300
d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=torch.device("cuda", 0), missing=0.3)
kf = BatchedKalman(layout="time_major").set_variant("wide_smoother", os.environ.get("VARIANT", "mfma"))
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
bufs = kf.alloc_projection(B)
for _ in range(2):
    kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
torch.cuda.synchronize()
kf.enable_timing(True, accumulate=True)
for _ in range(4):
    kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
torch.cuda.synchronize()
f_tot, f_n, s_tot, s_n = kf.kernel_ms_totals()
print(os.environ.get("VARIANT", "mfma"), "%-40s filter %.2f ms smoother %.2f ms" % (os.environ.get("METRAN_HIP_LIBRARY", "default").split("/")[-1], f_tot / f_n, s_tot / s_n))
