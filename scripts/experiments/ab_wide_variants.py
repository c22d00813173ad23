"""Same-box A/B of the wide-smoother variants at configs[3] size (4096 x (32,4), 30 % missing, projection path), interleaved.
  gpurun -- 'python scripts/experiments/ab_wide_variants.py [T [variant ...]]'      (default: mfma mfma_unfolded;
  (the block-path variants mfma_blk4 / mfma_blk4_unfolded were removed in round 4: README.md)"""
import sys

import torch

sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

B, N, K = 4096, 32, 4
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
VARIANTS = tuple(sys.argv[2:]) or ("mfma", "mfma_unfolded")
d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=torch.device("cuda", 0), missing=0.3)
res = {}
for rnd in range(2):
    for variant in VARIANTS:
        kf = BatchedKalman(layout="time_major").set_variant("wide_smoother", variant)
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        bufs = kf.alloc_projection(B)
        kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
        torch.cuda.synchronize()
        kf.enable_timing(True, accumulate=True)
        for _ in range(3):
            kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
        torch.cuda.synchronize()
        f_tot, f_n, s_tot, s_n = kf.kernel_ms_totals()
        print("%-14s round %d  filter %.2f ms  smoother %.2f ms" % (variant, rnd, f_tot / f_n, s_tot / s_n), flush=True)
        res.setdefault(variant, []).append((bufs["sim_means"].clone(), bufs["sim_vars"].clone()))
        kf.close()
        del bufs
a = res[VARIANTS[0]][0]
for v in VARIANTS[1:]:
    b = res[v][0]
    print("%s vs %s: bit-identical projection outputs: %s; max |d mean| %.2e, max |d var| %.2e"
          % (v, VARIANTS[0], torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), float((a[0] - b[0]).abs().max()),
             float((a[1] - b[1]).abs().max())))
