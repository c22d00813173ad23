"""End-to-end calibration throughput (SURVEY.md section 8, row f1 / BASELINE configs[4]'s workload in fp64):
R independent 8-series/2-factor DFMs PER GPU calibrated in lock-step by ``calibrate_batch``; with ``--gpus N`` the
records are sharded over N ranks (``calibrate_sharded``: ``shard_range`` + rank-order gather, no collective in the
data path -- per-model parameters) and the run is weak-scaled like bench.py.

    python scripts/bench_calibrate.py [--gpus N] [--batch 8192] [--T 1000] [--maxiter 200]

From a plain ``python`` with ``--gpus N > 1`` the script re-executes itself under torch.distributed.run.  Rank 0 prints
one JSON line; ``evals_per_s`` counts filter instances (objective evaluations), the unit the reference's solver loop
spends its time on (one get_mle = one filter run)."""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--batch", type=int, default=8192, help="models per GPU")
ap.add_argument("--T", type=int, default=1000)
ap.add_argument("--maxiter", type=int, default=200)
ap.add_argument("--series", type=int, default=8)
ap.add_argument("--factors", type=int, default=2)
ap.add_argument("--missing", type=float, default=0.0)
ap.add_argument("--gradient", default="auto", choices=["auto", "adjoint", "fd"],
                help="fd = one launch of (n+1) x models differenced filter instances per gradient (what wide models had before round 3)")
ap.add_argument("--fd-below", type=int, default=4096,
                help="switch from the adjoint gradient to batched forward differences once (n+1) x active models <= this (0: never)")
a = ap.parse_args()

if a.gpus > 1 and "RANK" not in os.environ:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

import torch  # noqa: E402

from metran_amd.calibrate import calibrate_batch, calibrate_sharded  # noqa: E402
from metran_amd.distributed import init_from_env, shard_range  # noqa: E402
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

rank, world, local_rank = init_from_env()
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
N, K = a.series, a.factors
total = a.batch * world
lo, hi = shard_range(total, rank, world)
d = make_dfm_batch_torch(hi - lo, N, K, a.T, seed=5000 + rank, device=dev, missing=a.missing)
kf = BatchedKalman(local_rank, layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
calibrate_batch(kf, maxiter=2, gradient=a.gradient)  # warm-up (kernel load, allocator)
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
t0 = time.perf_counter()
res = calibrate_sharded(total, lambda lo_, hi_: kf, maxiter=a.maxiter, fd_below=a.fd_below, gradient=a.gradient)  # this rank's engine holds exactly [lo, hi)
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
nfev = torch.tensor([float(res.nfev)], dtype=torch.float64, device=dev)
if world > 1:
    torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
    torch.distributed.all_reduce(nfev)
dt = float(dt.item())
true_obj = kf.loglik(d["phi"], d["q"])
ok_local = (res.obj[lo:hi] <= true_obj + 1e-6).double().mean()
if rank == 0:
    print(json.dumps({"workload": "calibrate_sharded: %d x (%d series, %d factors) per GPU on %d GPU(s), T=%d, %d %% missing, fp64" % (
                          a.batch, N, K, world, a.T, round(100 * a.missing)),
                      "n_gpus": world, "models": int(res.alpha.shape[0]), "seconds": dt, "models_per_s": total / dt,
                      "gradient": ("batched forward differences" if a.gradient == "fd" else
                                   "adjoint" + (", forward differences once (n+1) x active models <= %d" % a.fd_below if a.fd_below else "")),
                      "iterations_rank0": int(res.nit), "nfev_all_ranks": int(nfev.item()), "evals_per_s": float(nfev.item()) / dt,
                      "converged_frac": float(res.converged.double().mean()),
                      "rank0_frac_at_or_below_true_parameter_objective": float(ok_local),
                      "median_pgnorm": float(res.pgnorm.median())}))
if world > 1:
    torch.distributed.destroy_process_group()
