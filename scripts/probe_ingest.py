"""Dev probe: throughput of the ingestion kernels (row f3) at the benchmark size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
B, N, K, T = 4096, 8, 2, 1000
dev = torch.device("cuda", 0)
d = make_dfm_batch_torch(B, N, K, T, seed=3, device=dev, missing=0.2)
raw = d["obs"] * 2.5 + 7.0
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
nbytes = raw.numel() * 8
for layout in ("time_major", "model_major"):
    kf = BatchedKalman(0, layout=layout)
    kf.set_observations(raw)
    t_std = timeit(lambda: kf.standardize())           # 3 reads + 1 write of the slab (L2 absorbs the re-reads)
    mask = (torch.rand(raw.shape, device=dev) < 0.1)
    kf.set_observations(raw)
    t_mask = timeit(lambda: kf.mask_observations(mask))
    t_pack = timeit(lambda: kf.pack_observations())
    print("%-11s standardize %.3f ms (%.0f GB/s of 2 x %.0f MB)   mask %.3f ms (%.0f GB/s)   pack %.3f ms (%.0f GB/s)" % (
        layout, t_std, 2 * nbytes / t_std / 1e6, nbytes / 1e6, t_mask, (2 * nbytes + nbytes / 8) / t_mask / 1e6,
        t_pack, (3 * nbytes + nbytes / N) / t_pack / 1e6))
