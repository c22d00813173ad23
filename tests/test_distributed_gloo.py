"""world_size-2 CPU test (gloo) of the N>1 path: batch sharding + the one collective of the
path (all-reduce of the summed -2 log L).  The per-rank evaluation that the GPU performs with
``BatchedKalman.loglik`` is stood in for by the oracle here -- the code under test is
``metran_amd.distributed`` (shard_range / ShardedObjective / allreduce_sum / gather_concat),
exactly what bench.py and a sharded solver run on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, N, K, T, SEED = 11, 4, 1, 60, 321  # 11 models over 2 ranks: ragged shards (6 + 5)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle
    from metran_amd.distributed import ShardedObjective, gather_concat, init_from_env, shard_range
    from metran_amd.params import phi_q_from_alpha
    from metran_amd.synthetic import make_dfm_batch

    r, w, _ = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = shard_range(B, rank, world)
    d = make_dfm_batch(hi - lo, N, K, T, seed=SEED, missing=0.2, start=lo)  # this rank's records only

    def local_loglik(alpha_shared):  # shared-parameter objective: same alpha for every model
        phi, q = phi_q_from_alpha(np.broadcast_to(alpha_shared, (hi - lo, N + K)), d["loadings"])
        ref = oracle.dfm_batch(d["obs"], phi, q, d["loadings"], smooth=False, outputs="mle")
        return torch.from_numpy(ref["mle"])

    obj = ShardedObjective(local_loglik)
    alpha = np.linspace(5.0, 25.0, N + K)
    total = float(obj(alpha))
    allv = gather_concat(local_loglik(alpha))
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.r_[total, allv.numpy()])

    # objective + gradient w.r.t. the shared parameters in one all-reduce of P+1 doubles; the local gradient
    # that the GPU gets from the adjoint kernel is its numpy restatement here
    import adjoint_ref

    def local_value_and_grad(alpha_shared):
        vals, grads = [], []
        for b in range(hi - lo):
            ph, qq = phi_q_from_alpha(alpha_shared, d["loadings"][b])
            m, gp, gq = adjoint_ref.gradient(d["obs"][b], ph, qq, d["loadings"][b])
            c = np.r_[1.0 - (d["loadings"][b] ** 2).sum(1), np.ones(K)]
            vals.append(m)
            grads.append((gp - 2.0 * ph * c * gq) * ph / alpha_shared ** 2)
        return torch.tensor(vals), torch.tensor(np.array(grads))

    tot, grad = obj.value_and_grad(alpha, local_value_and_grad)
    np.save(os.path.join(out_dir, "g%d.npy" % rank), np.r_[float(tot), grad.numpy()])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_objective_world2(tmp_path):
    sys.path.insert(0, ROOT)
    import oracle
    from metran_amd.params import phi_q_from_alpha
    from metran_amd.synthetic import make_dfm_batch

    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    d = make_dfm_batch(B, N, K, T, seed=SEED, missing=0.2)
    alpha = np.linspace(5.0, 25.0, N + K)
    phi, q = phi_q_from_alpha(np.broadcast_to(alpha, (B, N + K)), d["loadings"])
    ref = oracle.dfm_batch(d["obs"], phi, q, d["loadings"], smooth=False, outputs="mle")["mle"]
    r0 = np.load(tmp_path / "r0.npy")
    r1 = np.load(tmp_path / "r1.npy")
    assert r0[0] == r1[0]                                   # every rank holds the same reduced value
    assert abs(r0[0] - ref.sum()) <= 1e-12 * abs(ref.sum())  # sharding invariance of the objective
    np.testing.assert_array_equal(r0[1:], ref)              # rank-order concatenation = batch order
    np.testing.assert_array_equal(r1[1:], ref)
    # summed objective and gradient: identical on both ranks, equal to central differences of the full sum
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    np.testing.assert_array_equal(g0, g1)
    assert abs(g0[0] - ref.sum()) <= 1e-10 * abs(ref.sum())

    def total(a):
        ph, qq = phi_q_from_alpha(np.broadcast_to(a, (B, N + K)), d["loadings"])
        return oracle.dfm_batch(d["obs"], ph, qq, d["loadings"], smooth=False, outputs="mle")["mle"].sum()

    for i in range(N + K):
        h = 1e-5 * alpha[i]
        e = np.zeros(N + K)
        e[i] = h
        fd = (total(alpha + e) - total(alpha - e)) / (2 * h)
        assert abs(g0[1 + i] - fd) <= 1e-5 * max(1e-3, abs(fd))


def test_bench_launches_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` from a plain interpreter (how the driver may call it)
    re-executes itself under torch.distributed.run; checked here without a GPU in --dry-run mode (gloo): two
    ranks rendezvous on 127.0.0.1, the barrier / max-over-ranks / all-reduce plumbing runs, rank 0 prints ONE
    JSON line that says it is not a measurement."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["value"] is None and "dry_run" in res
    assert res["allreduce_check"] == 3.0  # ranks contributed 1 + 2
    assert res["config"]["name"] == "c2" and res["config"]["batch_per_gpu"] == 4096


def test_bench_eight_ranks_default_to_configs2(tmp_path):
    """``bench.py --gpus 8`` without ``--config`` is configs[2]'s per-GPU share (8192 models: 65536 over 8 GPUs), so the
    driver's 8-GPU line lands on the BASELINE configuration; fewer ranks stay on configs[1] (checked above).  Dry run (gloo)."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "1",
                          "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 8 and res["rccl_ranks"] == 8
    assert res["config"]["name"] == "c3" and res["config"]["batch_per_gpu"] == 8192 and res["config"]["total_batch"] == 65536
    assert "configs[2]" in res["config"]["workload"]


def _worker_calibrate(rank, world, port, out_dir):
    """Multi-GPU calibration paths on 2 CPU ranks: ``run_sharded`` (per-model parameters:
    slice, local work, rank-order gather -- what ``calibrate_sharded`` wraps around ``calibrate_batch``) and
    ``calibrate_shared`` (one parameter vector for all models of all ranks, scipy L-BFGS-B on the summed objective
    and summed gradient, one all-reduce of P+1 doubles per evaluation).  The per-rank filter evaluation that the GPU
    performs (mk_loglik_grad) is stood in for by the oracle's objective + adjoint restatement."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import adjoint_ref
    from metran_amd.calibrate import calibrate_shared
    from metran_amd.distributed import init_from_env, run_sharded, world as world_fn
    from metran_amd.params import phi_q_from_alpha
    from metran_amd.synthetic import make_dfm_batch

    init_from_env(backend="gloo")
    assert world_fn() == (rank, world)
    Bc, Tc = 5, 40  # 5 models over 2 ranks: shards of 3 and 2

    def local(lo, hi):  # per-model "results" that identify model and owner
        idx = torch.arange(lo, hi, dtype=torch.float64)
        return {"alpha": torch.stack([idx, idx * 10.0 + rank], 1), "obj": idx * idx}

    g = run_sharded(Bc, local)
    assert g["alpha"].shape == (Bc, 2) and g["obj"].tolist() == [float(i * i) for i in range(Bc)]
    assert g["alpha"][:, 1].tolist() == [0.0, 10.0, 20.0, 31.0, 41.0]  # models 3, 4 came from rank 1

    from metran_amd.distributed import shard_range

    lo, hi = shard_range(Bc, rank, world)
    d = make_dfm_batch(hi - lo, N, K, Tc, seed=SEED + 7, start=lo)

    def local_value_and_grad(alpha_shared):
        a = alpha_shared.numpy()
        vals, grads = [], []
        for b in range(hi - lo):
            ph, qq = phi_q_from_alpha(a, d["loadings"][b])
            m, gp, gq = adjoint_ref.gradient(d["obs"][b], ph, qq, d["loadings"][b])
            c = np.r_[1.0 - (d["loadings"][b] ** 2).sum(1), np.ones(K)]
            vals.append(m)
            grads.append((gp - 2.0 * ph * c * gq) * ph / a ** 2)
        return torch.tensor(vals), torch.tensor(np.array(grads))

    res = calibrate_shared(local_value_and_grad, np.full(N + K, 10.0), options={"maxiter": 6})
    np.save(os.path.join(out_dir, "c%d.npy" % rank), np.r_[res.fun, res.x, res.nfev])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_and_shared_calibration_world2(tmp_path):
    world = 2
    mp.spawn(_worker_calibrate, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    c0, c1 = np.load(tmp_path / "c0.npy"), np.load(tmp_path / "c1.npy")
    np.testing.assert_array_equal(c0, c1)  # every rank followed the same iteration to the same point
    sys.path.insert(0, ROOT)
    import oracle
    from metran_amd.params import phi_q_from_alpha
    from metran_amd.synthetic import make_dfm_batch

    d = make_dfm_batch(5, N, K, 40, seed=SEED + 7)

    def total(a):
        ph, qq = phi_q_from_alpha(np.broadcast_to(a, (5, N + K)), d["loadings"])
        return oracle.dfm_batch(d["obs"], ph, qq, d["loadings"], smooth=False, outputs="mle")["mle"].sum()

    x = c0[1:-1]
    assert abs(c0[0] - total(x)) <= 1e-9 * abs(c0[0])       # the reported optimum is the summed objective there
    assert c0[0] < total(np.full(N + K, 10.0)) - 1e-3        # and it improved on the start


def _worker_calibrate_sharded(rank, world, port, out_dir):
    """``calibrate_sharded`` ITSELF on 2 CPU ranks: every rank builds an engine over its slice of the models (the oracle-backed
    stand-in of tests/oracle_engine.py where the GPU box has ``BatchedKalman``), runs ``calibrate_batch`` on it, and the
    per-model results come back in model order on every rank."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from metran_amd.calibrate import calibrate_sharded
    from metran_amd.distributed import init_from_env
    from metran_amd.synthetic import make_dfm_batch
    from oracle_engine import OracleEngine

    init_from_env(backend="gloo")
    Bc = 5                                              # shards of 3 and 2
    built = []

    def build_engine(lo, hi):
        built.append((lo, hi))
        d = make_dfm_batch(hi - lo, 3, 1, 120, seed=SEED + 11, missing=0.2, start=lo)   # this rank's records only
        return OracleEngine(d["obs"], d["loadings"])

    res = calibrate_sharded(Bc, build_engine, gradient="adjoint", compact=0, stderr=False)
    assert built == [((0, 3), (3, 5))[rank]] and res.shard == built[0]
    assert res.alpha.shape == (Bc, 4) and res.converged.dtype == torch.bool and res.aic.shape == (Bc,)
    np.save(os.path.join(out_dir, "s%d.npy" % rank),
            np.c_[res.alpha.numpy(), res.obj.numpy(), res.pgnorm.numpy(), res.converged.numpy().astype(float)])
    dist.barrier()
    dist.destroy_process_group()


def test_calibrate_sharded_world2_equals_the_single_process_run(tmp_path):
    world = 2
    mp.spawn(_worker_calibrate_sharded, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    s0, s1 = np.load(tmp_path / "s0.npy"), np.load(tmp_path / "s1.npy")
    np.testing.assert_array_equal(s0, s1)               # every rank holds every model's result
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from metran_amd.calibrate import calibrate_batch
    from metran_amd.synthetic import make_dfm_batch
    from oracle_engine import OracleEngine

    d = make_dfm_batch(5, 3, 1, 120, seed=SEED + 11, missing=0.2)
    one = calibrate_batch(OracleEngine(d["obs"], d["loadings"]), gradient="adjoint", compact=0)
    assert bool(one.converged.all()) and np.all(s0[:, 6] == 1.0)
    # per-model parameters: a model's iterates do not depend on which flight it rides in
    np.testing.assert_allclose(s0[:, :4], one.alpha.numpy(), rtol=1e-9)
    np.testing.assert_allclose(s0[:, 4], one.obj.numpy(), rtol=1e-12)
