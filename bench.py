#!/usr/bin/env python
"""bench.py -- throughput of the hot path on MI355X (contract: see task description / DESIGN.md).

One "step" = one pass of the hot path over one batch: batched Kalman filter + -2 log L + RTS
smoother with every reference-equivalent state output materialised in HBM (F, Pf, Xp, Pp, S, Ps),
followed by the summed-objective reduction (local deterministic sum + one RCCL all-reduce when
N > 1).  Workload = BASELINE.json configs[1]: batch=4096 synthetic 8-series / 2-factor DFMs,
T=1000, fp64, per GPU (weak scaling: every rank owns its own 4096 models; configs[2] is the same
per-GPU load at 8 GPUs).  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)


def algorithmic_bytes(N, K, T):
    """bytes per model that the algorithm has to move."""
    n = N + K
    c = n + n * n
    filt = 8 * T * (N + 2 * c)   # read obs, write filtered (F,Pf) and predicted (Xp,Pp)
    smooth = 8 * T * (2 * c)     # re-read filtered, write smoothed (S,Ps)
    return filt, smooth


def pmc_traffic(kernel_key, B, N, K, T, layout):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of THIS command
    (profiles/<round>/pmc_hbm.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE is doubled
    as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; units KiB).  None when no
    profile of the same workload is committed."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_hbm.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("batch"), w.get("series"), w.get("factors"), w.get("T"), w.get("layout")) != (B, N, K, T, layout):
            continue
        for name, c in d.get("kernels", {}).items():
            if kernel_key in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                best = {"GB": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e9, "source": os.path.relpath(f, ROOT)}
    return best


def cpu_baseline(d_host, gpu_mle, target_seconds=15.0):
    """Reference algorithm (oracle/kalman_oracle.c = C port of the numba/numpy path) on the host
    cores of this box, on a bounded sample of the same workload."""
    import numpy as np

    import oracle

    native = False
    try:
        oracle.build(native=True)  # -march=native build for this host; falls back to the portable .so
        native = True
    except Exception:
        pass
    cores = oracle.num_threads(native)
    B = d_host["obs"].shape[0]
    probe = min(B, 2 * cores)
    sl = slice(0, probe)
    t0 = time.perf_counter()
    oracle.dfm_batch(d_host["obs"][sl], d_host["phi"][sl], d_host["q"][sl], d_host["loadings"][sl], native=native)
    per_model = (time.perf_counter() - t0) / probe
    n = int(max(probe, min(B, target_seconds / max(per_model, 1e-9))))
    n = max(cores, (n // cores) * cores)
    sl = slice(0, n)
    t0 = time.perf_counter()
    ref = oracle.dfm_batch(d_host["obs"][sl], d_host["phi"][sl], d_host["q"][sl], d_host["loadings"][sl],
                           native=native)
    dt = time.perf_counter() - t0
    rel = float(np.max(np.abs(gpu_mle[:n] - ref["mle"]) / np.abs(ref["mle"])))
    T = d_host["obs"].shape[1]
    return {
        "value": n * T / dt,
        "unit": "model-timesteps/s",
        "models_per_s": n / dt,
        "cores": cores,
        "kind": "port",
        "sample": "%d of the %d models of rank 0's batch, full T=%d, filter+smoother with all outputs, "
                  "OpenMP over models, %.1f s (C restatement of kalmanfilter.py:236-476, %s)"
                  % (n, B, T, dt, "-O3 -march=native" if native else "-O3"),
    }, rel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="models per GPU")
    ap.add_argument("--series", type=int, default=8)
    ap.add_argument("--factors", type=int, default=2)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loglik-only", action="store_true", help="time the solver objective (no state outputs)")
    ap.add_argument("--layout", default="time_major", choices=["time_major", "model_major"],
                    help="memory layout of the per-step arrays (see BatchedKalman)")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs torch.distributed.run (one process per GPU)" % args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # launched by torch.distributed.run: one rank per GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from metran_amd.engine import BatchedKalman
    from metran_amd.synthetic import make_dfm_batch_torch

    B, N, K, T = args.batch, args.series, args.factors, args.T
    n = N + K
    d = make_dfm_batch_torch(B, N, K, T, seed=2000 + rank, device=dev)
    kf = BatchedKalman(local_rank, layout=args.layout)
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    outputs = () if args.loglik_only else ("F", "Pf", "Xp", "Pp", "S", "Ps")
    bufs = kf._alloc_outputs(B, list(outputs))
    total = torch.zeros(1, dtype=torch.float64, device=dev)

    def step():
        if args.loglik_only:
            kf.loglik(d["phi"], d["q"], out=bufs["mle"])
        else:
            kf.filter_smooth(d["phi"], d["q"], buffers=bufs)
        s = kf.sum(bufs["mle"])            # deterministic local reduction
        total.copy_(s.reshape(1))
        if dist is not None:
            dist.all_reduce(total)          # RCCL all-reduce of the summed -2 log L (8 bytes)
        return total

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    kf.enable_timing(True)
    torch.cuda.synchronize()
    filt_ms, smooth_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        f_ms, s_ms = kf.last_kernel_ms()   # hipEvents on the launch stream
        filt_ms.append(f_ms)
        smooth_ms.append(s_ms)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kf.enable_timing(False)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        models_per_s = world * B * args.steps / elapsed
        fb, sb = algorithmic_bytes(N, K, T)
        f_avg = float(np.mean(filt_ms))
        s_avg = float(np.mean(smooth_ms)) if not args.loglik_only else 0.0
        kernels = {"filter_kernel": {"ms": f_avg, "algorithmic_GB": fb * B / 1e9,
                                     "GBps": fb * B / 1e9 / (f_avg / 1e3)}}
        if not args.loglik_only:
            kernels["smoother_record_kernel"] = {"ms": s_avg, "algorithmic_GB": sb * B / 1e9,
                                          "GBps": sb * B / 1e9 / (s_avg / 1e3)}
        for kname in kernels:
            tr = pmc_traffic(kname, B, N, K, T, args.layout)
            kernels[kname]["traffic_GB"] = tr["GB"] if tr else None
            if tr:
                kernels[kname]["traffic_source"] = tr["source"]
        dom = max(kernels, key=lambda k: kernels[k]["ms"])
        if args.loglik_only:
            kernels["filter_kernel"]["algorithmic_GB"] = 8 * T * N * B / 1e9
            kernels["filter_kernel"]["GBps"] = 8 * T * N * B / 1e9 / (f_avg / 1e3)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS,
                    "traffic": (kernels[dom]["traffic_GB"] * 1e9 if kernels[dom].get("traffic_GB") else None),
                    "algorithmic_bytes": kernels[dom]["algorithmic_GB"] * 1e9,
                    "avg_launch_ms": kernels[dom]["ms"], "kernels": kernels,
                    "path_achieved_GBps": (fb + sb) * B / 1e9 / ((f_avg + s_avg) / 1e3) if not args.loglik_only else None}
        res = {
            "metric": "Kalman filter+smoother steps/sec (batched DFMs)",
            "value": models_per_s * T,
            "unit": "model-timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: batch=%d synthetic %d-series/%d-factor DFMs per GPU, "
                                   "T=%d, fp64%s" % (B, N, K, T, ", loglik only" if args.loglik_only else
                                                     ", filter+smoother, outputs F,Pf,Xp,Pp,S,Ps"),
                       "batch_per_gpu": B, "series": N, "factors": K, "T": T, "parallelism": "dp%d" % world,
                       "layout": args.layout},
            "models_per_s": models_per_s,
            "models_per_s_per_gpu": models_per_s / world,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and not args.loglik_only:
            host = {k: d[k].cpu().numpy() for k in ("obs", "phi", "q", "loadings")}
            base, rel = cpu_baseline(host, bufs["mle"].cpu().numpy())
            res["cpu_baseline"] = base
            res["loglik_max_rel_err"] = rel
            res["speedup_vs_cpu_baseline"] = res["value"] / base["value"]
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
