"""Run under ``python -m torch.distributed.run --nproc-per-node 1`` by tests/test_rccl_gpu.py (GPU box): the distributed
code path of the repository on RCCL ("nccl") with ONE rank -- ShardedObjective (value, and value + gradient in one
all-reduce), gather_concat, calibrate_sharded -- against the same computation with no process group, bit for bit."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from metran_amd.calibrate import calibrate_batch, calibrate_sharded  # noqa: E402
from metran_amd.distributed import ShardedObjective, attach_communicator, gather_concat, init_from_env, shard_range, world  # noqa: E402
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch  # noqa: E402

rank, size, local_rank = init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl" and (rank, size) == (0, 1) == world()
R, N, K, T = 24, 8, 2, 200
d = make_dfm_batch(R, N, K, T, seed=77, missing=0.1)


def engine(lo, hi):
    kf = BatchedKalman(local_rank)
    kf.set_observations(d["obs"][lo:hi]).set_loadings(d["loadings"][lo:hi])
    return kf


lo, hi = shard_range(R, rank, size)
kf = engine(lo, hi)
alpha = torch.linspace(5.0, 25.0, N + K, dtype=torch.float64, device="cuda")


def local_loglik(a):
    phi, q = kf.params_from_alpha(a[None].repeat(hi - lo, 1))
    return kf.loglik(phi, q)


def local_vg(a):
    return kf.loglik_grad_alpha(a[None].repeat(hi - lo, 1))


C_ABI = "--c-abi" in sys.argv   # the all-reduce through the library's own communicator (mk_allreduce_sum) instead of torch.distributed's
if C_ABI:
    attach_communicator(kf)      # rank 0's unique id travels over the process group, every rank joins
obj = ShardedObjective(local_loglik, local_sum=kf.sum, engine=kf if C_ABI else None)
total = obj(alpha)                                         # fixed-order local sum + all_reduce on RCCL
plain = kf.sum(local_loglik(alpha)).reshape(1)[0]          # the same without any collective
tot2, grad = obj.value_and_grad(alpha, local_vg)           # P + 1 doubles in one all_reduce
v, g = local_vg(alpha)
gathered = gather_concat(local_loglik(alpha))              # all_gather path
ones = torch.ones(1, dtype=torch.float64, device="cuda")
dist.all_reduce(ones)
sharded = calibrate_sharded(R, engine, maxiter=60, stderr=True)
single = calibrate_batch(engine(0, R), maxiter=60, stderr=True)
torch.cuda.synchronize()
print("RCCL1 " + json.dumps({
    "backend": dist.get_backend(), "ranks": float(ones.item()), "c_abi_communicator": bool(kf.has_communicator()),
    "sum_bitwise": bool(total == plain), "total": float(total),
    "grad_bitwise": bool(torch.equal(grad, g.sum(0)) and tot2 == kf.sum(v).reshape(1)[0]),
    "gather_bitwise": bool(torch.equal(gathered, local_loglik(alpha))),
    "calibrate_bitwise": bool(torch.equal(sharded.alpha, single.alpha) and torch.equal(sharded.obj, single.obj)
                              and torch.equal(sharded.stderr, single.stderr)),
    "calibrate_converged": int(sharded.converged.sum()), "models": R}))
dist.destroy_process_group()
