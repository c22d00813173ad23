"""GPU: the one collective of the design executed on RCCL on hardware.  No multi-GPU box is needed to
run ``init_process_group("nccl")`` + ``all_reduce``: world size 1 under ``torch.distributed.run``.  The 2-rank behaviour
(ragged shards, rank-order gather) is covered on CPU by tests/test_distributed_gloo.py; the driver's N = 1, 2, 4, 8 run
is the scaling measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_bench_under_torchrun_allreduces_on_rccl():
    """``bench.py --gpus 1`` launched the way the driver launches N > 1: the nccl branch, one all_reduce of the summed
    -2 log L per step inside the timed region, rccl_ranks summed by dist.all_reduce."""
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                     "--no-secondary"])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["rccl_ranks"] == 1 and res["n_gpus"] == 1
    assert res["collective"]["backend"] == "nccl" and res["collective"]["allreduces_in_timed_region"] == 2
    assert res["config"]["total_batch"] == res["config"]["batch_per_gpu"] == 4096
    assert res["value"] > 1e8                                   # > 100 k models/s x T = 1000 (north star)
    # without a process group the same run reports no collective instead of pretending
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                            "--no-secondary"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                           env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID")})
    assert plain.returncode == 0, plain.stderr[-3000:]
    p = json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][-1])
    assert p["collective"]["backend"] is None and p["collective"]["allreduces_in_timed_region"] == 0
    assert p["summed_mle"] == res["summed_mle"]                # the all-reduce over one rank is an identity, bit for bit


def test_sharded_objective_and_calibration_on_one_rccl_rank():
    out = _torchrun([os.path.join(ROOT, "tests", "rccl_world1_script.py")])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RCCL1 ")][-1][6:])
    assert res["backend"] == "nccl" and res["ranks"] == 1.0
    assert res["sum_bitwise"] and res["grad_bitwise"] and res["gather_bitwise"] and res["calibrate_bitwise"]
    assert res["calibrate_converged"] >= res["models"] - 2


def test_c_abi_allreduce_on_one_rccl_rank():
    """``mk_comm_unique_id`` -> ``mk_comm_init_rank`` -> ``mk_allreduce_sum`` through the C ABI, a communicator of ONE rank in
    this process (no torch.distributed): the collective runs on RCCL and is an identity bit for bit; before the
    communicator exists, and after ``mk_comm_destroy``, the call fails loudly."""
    import ctypes

    import torch

    from metran_amd._lib import MetranHipError
    from metran_amd.distributed import ShardedObjective, allreduce_sum, attach_communicator
    from metran_amd.engine import BatchedKalman

    kf = BatchedKalman(0)
    t = torch.arange(1.0, 12.0, dtype=torch.float64, device="cuda") / 7.0
    ref = t.clone()
    with pytest.raises(MetranHipError, match="no communicator"):
        kf.allreduce_sum(t)
    assert torch.equal(t, ref)
    attach_communicator(kf)                       # no process group: a communicator of one rank
    assert kf.has_communicator()
    for _ in range(3):
        kf.allreduce_sum(t)
    torch.cuda.synchronize()
    assert torch.equal(t, ref)
    assert allreduce_sum(t, kf) is t and torch.equal(t, ref)
    obj = ShardedObjective(lambda p: p * 2.0, local_sum=kf.sum, engine=kf)
    assert float(obj(ref)) == float(kf.sum(ref * 2.0))
    with pytest.raises(ValueError):
        kf.allreduce_sum(t.float())
    # raw C ABI, caller-owned communicator handed to a second context: mk_set_communicator
    L = kf._L
    assert L.mk_comm_destroy(kf._ctx) == 0
    with pytest.raises(MetranHipError, match="no communicator"):
        kf.allreduce_sum(t)
    kf.close()


def test_c_abi_allreduce_under_torchrun():
    out = _torchrun([os.path.join(ROOT, "tests", "rccl_world1_script.py"), "--c-abi"])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RCCL1 ")][-1][6:])
    assert res["c_abi_communicator"] and res["sum_bitwise"] and res["grad_bitwise"]
