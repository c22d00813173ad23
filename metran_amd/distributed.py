"""Multi-GPU sharding of the batch axis: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards trivially: every dynamic-factor model is independent, so
rank r owns a contiguous slice of the records and runs the same kernels on it -- there is NO
data-path collective.  The only exchange is the summed objective fed back to the solver
(shared-parameter calibration): each rank reduces its own -2 log L values in a fixed order
(``BatchedKalman.sum``) and one all-reduce(sum) of a single float64 (8 bytes, latency-bound)
combines the ranks.  The reference has no counterpart (single process, metran/solver.py:42-63).
"""
import os

__all__ = ["shard_range", "init_from_env", "world", "allreduce_sum", "gather_concat", "run_sharded", "ShardedObjective",
           "attach_communicator"]


def shard_range(n_items, rank, world_size):
    """Contiguous, balanced slice ``[lo, hi)`` of ``n_items`` for ``rank`` (first ``n % world``
    ranks get one extra item).  Deterministic and order-preserving, so concatenating the ranks'
    results in rank order reproduces the single-process order."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of size %d" % (rank, world_size))
    base, extra = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world_size, local_rank)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or "TORCHELASTIC_RUN_ID" in os.environ) and not dist.is_initialized():  # launched by torchrun
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def world():
    """``(rank, world_size)`` of the default process group; ``(0, 1)`` when torch.distributed is not initialised."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def run_sharded(n_items, local_fn):
    """The data-parallel pattern of the whole path: rank r works on its contiguous slice ``[lo, hi)`` of the
    ``n_items`` models -- ``local_fn(lo, hi) -> dict of tensors [hi-lo, ...]`` (e.g. ``calibrate_batch`` on an engine
    holding those records) -- and the per-model results are concatenated in rank order on every rank.  No collective
    inside ``local_fn``: models are independent."""
    rank, size = world()
    lo, hi = shard_range(n_items, rank, size)
    local = local_fn(lo, hi)
    out = {}
    for k, t in local.items():
        tail = tuple(t.shape[1:])
        out[k] = gather_concat(t.reshape(-1)).reshape((-1,) + tail)
    return out


def attach_communicator(engine):
    """Give ``engine`` (a ``BatchedKalman``) the library's own RCCL communicator over the ranks of the default process
    group (C ABI ``mk_comm_unique_id`` / ``mk_comm_init_rank``): rank 0 draws the 128-byte id, the group's own channel
    carries it to the others (``broadcast_object_list``: works on gloo and nccl alike), every rank joins.  From then on
    ``allreduce_sum(t, engine)`` runs ``mk_allreduce_sum`` -- the collective a C caller of the library gets -- instead of
    ``torch.distributed.all_reduce``.  Without a process group the engine forms a communicator of one rank."""
    import torch.distributed as dist

    rank, size = world()
    ident = [engine.comm_unique_id() if rank == 0 else None]
    if dist.is_available() and dist.is_initialized():
        dist.broadcast_object_list(ident, src=0)
    engine.init_communicator(size, rank, ident[0])
    return engine


def allreduce_sum(t, engine=None):
    """In-place all-reduce(sum) of a tensor.  With an ``engine`` that carries a communicator (``attach_communicator``):
    the C ABI's ``mk_allreduce_sum`` on the engine's stream.  Otherwise over the default process group
    (``torch.distributed``: "nccl" = RCCL on ROCm, gloo in the CPU tests); no-op when no group exists.  With a group of
    ONE rank the collective still runs (an identity, bit for bit): a job launched by torchrun on one GPU exercises the
    same RCCL call as on eight."""
    import torch.distributed as dist

    if engine is not None and engine.has_communicator():
        return engine.allreduce_sum(t)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def gather_concat(t):
    """All-gather rank-local 1-D results (possibly of different lengths) and concatenate them in
    rank order (optional reporting path; the hot path never needs it)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return t
    world = dist.get_world_size()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t.reshape(-1)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: int(s.item())] for p, s in zip(parts, sizes)])


class ShardedObjective:
    """Summed -2 log L of B models sharded over the ranks.

    ``local_loglik(params) -> 1-D tensor`` evaluates this rank's models (on the GPU:
    ``BatchedKalman.loglik``); ``local_sum`` reduces them deterministically
    (``BatchedKalman.sum``); the ranks are combined with one all-reduce.
    """

    def __init__(self, local_loglik, local_sum=None, engine=None):
        self.local_loglik = local_loglik
        self.local_sum = local_sum or (lambda v: v.sum())
        self.engine = engine   # a BatchedKalman with a communicator: the all-reduce goes through the C ABI (mk_allreduce_sum)

    def __call__(self, params):
        vals = self.local_loglik(params)
        total = self.local_sum(vals).reshape(1).clone()
        return allreduce_sum(total, self.engine)[0]

    def value_and_grad(self, params, local_value_and_grad):
        """Summed objective AND its gradient with respect to the SHARED parameters:
        ``local_value_and_grad(params) -> (values [b], grads [b,P])`` for this rank's models (on the GPU:
        ``BatchedKalman.loglik_grad_alpha``, the adjoint kernel); the local sums of both ride in ONE
        all-reduce of P+1 float64.  Returns ``(total, grad [P])``, identical on every rank."""
        import torch

        vals, grads = local_value_and_grad(params)
        packed = torch.cat([self.local_sum(vals).reshape(1), grads.sum(0).reshape(-1)]).clone()
        allreduce_sum(packed, self.engine)
        return packed[0], packed[1:]
