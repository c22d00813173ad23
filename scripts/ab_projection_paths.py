"""Same-box A/B of the two projection paths of the wide models at configs[3] size (4096 x (32,4), 30 % missing):
"tape" (filter_split_kernel OUT = 4 + smoother_dk_kernel) against "records" (filtered records + the RTS smoother).
  gpurun -- 'python scripts/ab_projection_paths.py [T [B]]'"""
import sys

import torch

sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

N, K = 32, 4
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=torch.device("cuda", 0), missing=0.3)
res = {}
for rnd in range(2):
    for path in ("tape", "records"):
        kf = BatchedKalman(layout="time_major")
        kf.projection_path = path
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        bufs = kf.alloc_projection(B)
        kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
        torch.cuda.synchronize()
        kf.enable_timing(True, accumulate=True)
        for _ in range(3):
            kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
        torch.cuda.synchronize()
        f_tot, f_n, s_tot, s_n = kf.kernel_ms_totals()
        print("%-8s round %d  filter %.2f ms  smoother %.2f ms  -> %.0f models/s" % (path, rnd, f_tot / f_n, s_tot / s_n,
                                                                                 B / ((f_tot / f_n + s_tot / s_n) / 1e3)), flush=True)
        res[path] = (bufs["sim_means"].clone(), bufs["sim_vars"].clone(), bufs["mle"].clone())
        kf.close()
        del bufs
a, b = res["tape"], res["records"]
print("tape vs records: max |d mean| %.2e, max |d var| %.2e, max rel d mle %.2e"
      % (float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max()), float(((a[2] - b[2]) / b[2]).abs().max())))
