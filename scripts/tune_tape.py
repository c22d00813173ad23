"""Timing experiments on the tape path (configs[3] size): libraries built with -DMK_TUNE=<bit> skip one phase (WRONG results,
timing only -- never the product library).  gpurun -- 'for m in 0 1 2 ...; do METRAN_HIP_LIBRARY=ab/libmetran_tune_$m.so python scripts/tune_tape.py $m; done'"""
import sys

import torch

sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

N, K, T, B = 32, 4, 2000, 4096
d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=torch.device("cuda", 0), missing=0.3)
kf = BatchedKalman(layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
bufs = kf.alloc_projection(B)
kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
torch.cuda.synchronize()
kf.enable_timing(True, accumulate=True)
for _ in range(3):
    kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
torch.cuda.synchronize()
f_tot, f_n, s_tot, s_n = kf.kernel_ms_totals()
print("tune %-4s filter %.2f ms  smoother %.2f ms" % (sys.argv[1] if len(sys.argv) > 1 else "-", f_tot / f_n, s_tot / s_n), flush=True)
