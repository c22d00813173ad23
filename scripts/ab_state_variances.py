"""Same-box A/B of the two routes to the smoothed STATE means / variances of wide models at configs[3] size (4096 x (32,4),
30 % missing): the state tape (filter_split_kernel OUT = 4 with N + K entries per step + smoother_dk_kernel<..,STATE>) against
filtered records + the RTS kernel (smoother_mfma_kernel with the variance epilogue), and both against the plain projection
tape for scale.
  gpurun -- 'python scripts/ab_state_variances.py [T [B]]'"""
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
    for path in ("auto", "records", "projection"):
        kf = BatchedKalman(layout="time_major")
        kf.projection_path = "auto" if path == "projection" else path
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        if path == "projection":
            bufs = kf.alloc_projection(B)
            run = lambda: kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)  # noqa: E731
        else:
            bufs = kf.alloc_state_variances(B)
            run = lambda: kf.smooth_state_variances(d["phi"], d["q"], buffers=bufs)  # noqa: E731
        run()
        torch.cuda.synchronize()
        kf.enable_timing(True, accumulate=True)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        f_tot, f_n, s_tot, s_n = kf.kernel_ms_totals()
        print("%-10s round %d  filter %.2f ms  smoother %.2f ms  -> %.0f models/s" % (
            {"auto": "state tape", "records": "records", "projection": "proj tape"}[path], rnd, f_tot / f_n, s_tot / s_n,
            B / ((f_tot / f_n + s_tot / s_n) / 1e3)), flush=True)
        if path != "projection":
            res[path] = (bufs["S"].clone(), bufs["var"].clone(), bufs["mle"].clone())
        kf.close()
        del bufs
a, b = res["auto"], res["records"]
print("state tape vs records: max |d mean| %.2e, max |d var| %.2e, max rel d mle %.2e"
      % (float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max()), float(((a[2] - b[2]) / b[2]).abs().max())))
