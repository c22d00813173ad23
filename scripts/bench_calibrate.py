"""End-to-end calibration throughput (SURVEY.md section 8, row f1 / BASELINE configs[4] in fp64):
R independent 8-series/2-factor DFMs calibrated in lock-step by metran_amd.calibrate.calibrate_batch.
Prints one JSON line; `evals_per_s` counts filter instances (objective evaluations), the unit the
reference's solver loop spends its time on (one get_mle = one filter run)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metran_amd.calibrate import calibrate_batch
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--T", type=int, default=1000)
ap.add_argument("--maxiter", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda", 0)
N, K = 8, 2
d = make_dfm_batch_torch(a.batch, N, K, a.T, seed=5000, device=dev)
kf = BatchedKalman(0, layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
calibrate_batch(kf, maxiter=2)  # warm-up (kernel load, allocator)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = calibrate_batch(kf, maxiter=a.maxiter)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
true_obj = kf.loglik(d["phi"], d["q"])
print(json.dumps({"workload": "calibrate_batch %dx(8 series, 2 factors), T=%d, fp64" % (a.batch, a.T),
                  "seconds": dt, "models_per_s": a.batch / dt, "iterations": int(res.nit), "nfev": int(res.nfev),
                  "launches": int(res.launches), "evals_per_s": res.nfev / dt,
                  "converged_frac": float(res.converged.double().mean()),
                  "frac_at_or_below_true_parameter_objective": float((res.obj <= true_obj + 1e-6).double().mean()),
                  "median_pgnorm": float(res.pgnorm.median())}))
