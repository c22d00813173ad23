import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from metran_amd.calibrate import calibrate_batch
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
dev = torch.device("cuda", 0)
for (B, N, K, T, miss, fdb) in ((8192, 8, 2, 1000, 0.0, 4096), (8192, 8, 2, 1000, 0.0, 8192), (512, 32, 4, 500, 0.3, 4096)):
    d = make_dfm_batch_torch(B, N, K, T, seed=5000, device=dev, missing=miss)
    kf = BatchedKalman(0, layout="time_major")
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    calibrate_batch(kf, maxiter=2)
    true_obj = kf.loglik(d["phi"], d["q"])
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = calibrate_batch(kf, maxiter=200, fd_below=fdb)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("B=%d (%d,%d) fd_below=%d: %.3f s nit %d launches %d nfev %d converged %.4f at/below true %.4f" % (B, N, K, fdb, dt, res.nit, res.launches, res.nfev,
              float(res.converged.double().mean()), float((res.obj <= true_obj + 1e-6).double().mean())), flush=True)
    kf.close(); del d; torch.cuda.empty_cache()
