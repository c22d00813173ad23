"""Dev probe: wide (32 series / 4 factors) filter and smoother kernel time as a function of the number of models --
shows the occupancy quantisation (wavefront slots per CU x 256 CUs) of the one-model-per-wavefront kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch

N, K, T = 32, 4, int(os.environ.get("T", 500))
dev = torch.device("cuda", 0)
for B in [int(b) for b in os.environ.get("BS", "256,512,1024,1536,1792,2048,3072,3584,4096").split(",")]:
    d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=dev, missing=0.3)
    kf = BatchedKalman(0, layout="time_major")
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    kf.enable_timing(True)
    buf = kf.alloc_projection(B)
    for i in range(2):
        kf.simulate_smoothed(d["phi"], d["q"], buffers=buf)
        f, s = kf.last_kernel_ms()
    print("B %5d  filter %8.2f ms  smoother %8.2f ms   per 256 models: %.3f / %.3f ms" % (B, f, s, f * 256 / B, s * 256 / B), flush=True)
    del kf, d, buf
    torch.cuda.empty_cache()
