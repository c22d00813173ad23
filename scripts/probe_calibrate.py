"""Dev probe: where the time of calibrate_batch goes (kernel time vs wall time) and how many models are still active."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metran_amd.calibrate import calibrate_batch
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch

B = int(os.environ.get("B", "8192"))
d = make_dfm_batch_torch(B, 8, 2, 1000, seed=5000, device=torch.device("cuda:0"))
kf = BatchedKalman(0, layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
calibrate_batch(kf, maxiter=2)
torch.cuda.synchronize()
kf.enable_timing(True, accumulate=True)
t0 = time.perf_counter()
res = calibrate_batch(kf, maxiter=200, verbose=bool(int(os.environ.get("VERBOSE", "0"))), fd_below=int(os.environ.get("FD_BELOW", "0")))
print("frac at or below the generating parameters objective: %.4f; converged %.4f; median pgnorm %.2e" % (float((res.obj <= kf.loglik(d["phi"], d["q"]) + 1e-6).double().mean()), float(res.converged.double().mean()), float(res.pgnorm.median())))
torch.cuda.synchronize()
wall = time.perf_counter() - t0
f, nf, s, ns = kf.kernel_ms_totals()
print("wall %.3f s; filter-slot kernels %.3f s over %d launches; smoother-slot (adjoint) kernels %.3f s over %d launches; nit %d launches %d"
      % (wall, f / 1e3, nf, s / 1e3, ns, res.nit, res.launches))
