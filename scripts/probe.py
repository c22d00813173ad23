"""One parametrised probe for the GPU box (replaces the thirteen one-off ``probe_*.py`` of rounds 1-4, whose numbers live in
profiles/).  Kernel times are hipEvent pairs around the launches (``BatchedKalman.enable_timing``); wall times include the host.

    python scripts/probe.py kernels  [--shape 8,2] [--batch 4096] [--T 1000] [--missing 0] [--what full project state objective grad filter]
                                     [--layout time_major] [--reps 5] [--ramp 40] [--check 4]
    python scripts/probe.py calibrate [--shape 8,2] [--batch 8192] [--T 1000] [--fd-below 4096] [--gradient auto] [--trace]
    python scripts/probe.py dropin                      the unmodified reference class on examples/data (bench.secondary_dropin)
    python scripts/probe.py factor   [--R 4096] [--T 1000] [--N 8 32] [--K 2 4]
    python scripts/probe.py ingest   [--batch 4096]

``kernels --ramp K`` prints the kernel time of K consecutive launches (the clock / power-state ramp after idling);
``--check M`` compares M models with the oracle.  A shape outside the ahead-of-time list is built or found by metran_amd.jit;
``METRAN_HIP_JIT_FLAGS=-DMK_NO_TILED_SWEEPS`` gives the one-statement-per-FMA sweeps of 11 <= n <= 16 for an A/B."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _engine(N, K, T, B, missing, layout, seed=2000):
    import torch

    from metran_amd.engine import BatchedKalman
    from metran_amd.synthetic import make_dfm_batch_torch

    d = make_dfm_batch_torch(B, N, K, T, seed=seed, device=torch.device("cuda", 0), missing=missing)
    kf = BatchedKalman(0, layout=layout)
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    return kf, d


def kernels(a):
    import numpy as np
    import torch

    N, K = (int(v) for v in a.shape.split(","))
    kf, d = _engine(N, K, a.T, a.batch, a.missing, a.layout)
    kf.projection_path = a.path
    runs = {
        "full": lambda b: kf.filter_smooth(d["phi"], d["q"], buffers=b),
        "filter": lambda b: kf.filter(d["phi"], d["q"], buffers=b),
        "project": lambda b: kf.simulate_smoothed(d["phi"], d["q"], buffers=b),
        "state": lambda b: kf.smooth_state_variances(d["phi"], d["q"], buffers=b),
        "objective": lambda b: kf.loglik(d["phi"], d["q"], out=b),
        "grad": lambda b: kf.loglik_grad(d["phi"], d["q"]),
    }
    for what in a.what:
        bufs = {"full": lambda: kf._alloc_outputs(a.batch, ["F", "Pf", "Xp", "Pp", "S", "Ps"]),
                "filter": lambda: kf._alloc_outputs(a.batch, ["F", "Pf", "Xp", "Pp"]),
                "project": lambda: kf.alloc_projection(a.batch), "state": lambda: kf.alloc_state_variances(a.batch),
                "objective": lambda: torch.empty(a.batch, dtype=torch.float64, device=kf.device), "grad": lambda: None}[what]()
        run = runs[what]
        if a.ramp:
            kf.enable_timing(True)
            seq = []
            for _ in range(a.ramp):
                run(bufs)
                torch.cuda.synchronize()
                seq.append(kf.last_kernel_ms())
            print(what, "ramp (filter, smoother ms):", " ".join("%.2f/%.2f" % s for s in seq))
            kf.enable_timing(False)
        run(bufs)
        torch.cuda.synchronize()
        kf.enable_timing(True, accumulate=True)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            res = run(bufs)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.reps
        f, fn, s, sn = kf.kernel_ms_totals()
        kf.enable_timing(False)
        line = {"what": what, "shape": [N, K], "batch": a.batch, "T": a.T, "filter_ms": f / max(fn, 1), "second_kernel_ms": s / max(sn, 1),
                "wall_ms": 1e3 * wall, "models_per_s": a.batch / wall, "specialised": kf.specialised(),
                "tape": bool(isinstance(res, dict) and res.get("_tape"))}
        if a.check and what in ("full", "project", "state"):
            import oracle

            idx = np.linspace(0, a.batch - 1, a.check).astype(int)
            ref = oracle.dfm_batch(*(d[k][idx].cpu().numpy() for k in ("obs", "phi", "q", "loadings")))
            line["mle_rel_err"] = float(np.max(np.abs(res["mle"][idx].cpu().numpy() - ref["mle"]) / np.abs(ref["mle"])))
            if what == "full":
                line["Ps_abs_err"] = float(np.max(np.abs(res["Ps"][idx].cpu().numpy() - ref["Ps"])))
            if what == "state":
                line["S_abs_err"] = float(np.max(np.abs(res["S"][idx].cpu().numpy() - ref["S"])))
                line["var_abs_err"] = float(np.max(np.abs(res["var"][idx].cpu().numpy() - np.diagonal(ref["Ps"], axis1=2, axis2=3))))
        print(json.dumps(line), flush=True)
        del bufs, res
        torch.cuda.empty_cache()


def calibrate(a):
    import builtins

    import torch

    from metran_amd.calibrate import calibrate_batch

    N, K = (int(v) for v in a.shape.split(","))
    kf, d = _engine(N, K, a.T, a.batch, a.missing, "time_major", seed=5000)
    calibrate_batch(kf, maxiter=2)
    warm = kf.subset(torch.ones(a.batch, dtype=torch.bool, device=kf.device).nonzero().squeeze(1)[: max(2, a.batch // 2)])   # the compaction path's torch kernels
    calibrate_batch(warm, maxiter=1)
    warm.close()
    warm = kf.subset(torch.arange(min(a.batch, 8), device=kf.device))   # ... and the differenced tail's
    calibrate_batch(warm, maxiter=2, fd_below=10 ** 9)
    warm.close()
    if a.fd_below is None:
        a.fd_below = 4096 if N + K <= 16 else 2048
    stamps, real_print = [], builtins.print
    if a.trace:  # per-iteration wall time and flight size from calibrate_batch's own verbose lines
        builtins.print = lambda *x, **k: (torch.cuda.synchronize(), stamps.append((time.perf_counter(), " ".join(str(v) for v in x))))
    kf.enable_timing(True, accumulate=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        res = calibrate_batch(kf, maxiter=a.maxiter, fd_below=a.fd_below, gradient=a.gradient, verbose=a.trace, own_search_above=a.own_above)
    finally:
        builtins.print = real_print
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    f, fn, s, sn = kf.kernel_ms_totals()
    prev = t0
    for k, (t, line) in enumerate(stamps):
        if a.trace_all or k % 5 == 0 or k > len(stamps) - 5:
            print("it %3d active %5s  %.2f ms" % (k + 1, line.split("active")[1].split()[0], 1e3 * (t - prev)))
        prev = t
    print(json.dumps({"models": a.batch, "shape": [N, K], "seconds": dt, "models_per_s": a.batch / dt, "nit": int(res.nit),
                      "launches": int(res.launches), "nfev": int(res.nfev), "converged_frac": float(res.converged.double().mean()),
                      "forward_kernel_ms_total": f, "forward_launches": fn, "backward_kernel_ms_total": s, "backward_launches": sn}))


def dropin(a):
    import bench

    print(json.dumps(bench.secondary_dropin(), indent=1))


def factor(a):
    import bench
    import torch

    for N, K in zip(a.N, a.K):
        print(json.dumps(bench.secondary_factor_analysis(torch.device("cuda", 0), R=a.R, T=a.T, N=N, K=K, reps=3,
                                                         scipy_subset=256 if N > 16 else 0)))


def ingest(a):
    import torch

    kf, d = _engine(8, 2, 1000, a.batch, 0.2, "time_major", seed=3)
    raw = d["obs"] * 2.5 + 7.0

    def timeit(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    mask = torch.rand(d["obs"].shape, device=kf.device) < 0.1
    print(json.dumps({"standardize_ms": 1e3 * timeit(lambda: kf.standardize(raw)),
                      "mask_ms": 1e3 * timeit(lambda: kf.mask_observations(mask)),
                      "pack_ms": 1e3 * timeit(lambda: kf.pack_observations())}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("kernels")
    p.add_argument("--shape", default="8,2")
    p.add_argument("--batch", type=int, default=4096)
    p.add_argument("--T", type=int, default=1000)
    p.add_argument("--missing", type=float, default=0.0)
    p.add_argument("--what", nargs="+", default=["full"], choices=["full", "filter", "project", "state", "objective", "grad"])
    p.add_argument("--layout", default="time_major")
    p.add_argument("--reps", type=int, default=5)
    p.add_argument("--path", default="auto", choices=["auto", "tape", "records"], help="projection / state outputs of wide models")
    p.add_argument("--ramp", type=int, default=0)
    p.add_argument("--check", type=int, default=0)
    p.set_defaults(fn=kernels)
    p = sub.add_parser("calibrate")
    p.add_argument("--shape", default="8,2")
    p.add_argument("--batch", type=int, default=8192)
    p.add_argument("--T", type=int, default=1000)
    p.add_argument("--missing", type=float, default=0.0)
    p.add_argument("--fd-below", type=int, default=None, help="default: 4096 instances (models of at most 16 states), 2048 (wider)")
    p.add_argument("--gradient", default="auto")
    p.add_argument("--maxiter", type=int, default=200)
    p.add_argument("--own-above", type=int, default=None, help="own line search per model above this flight size")
    p.add_argument("--trace", action="store_true")
    p.add_argument("--trace-all", action="store_true", help="with --trace: every iteration's line, not every fifth")
    p.set_defaults(fn=calibrate)
    sub.add_parser("dropin").set_defaults(fn=dropin)
    p = sub.add_parser("factor")
    p.add_argument("--R", type=int, default=4096)
    p.add_argument("--T", type=int, default=1000)
    p.add_argument("--N", type=int, nargs="+", default=[8, 32])
    p.add_argument("--K", type=int, nargs="+", default=[2, 4])
    p.set_defaults(fn=factor)
    p = sub.add_parser("ingest")
    p.add_argument("--batch", type=int, default=4096)
    p.set_defaults(fn=ingest)
    a = ap.parse_args()
    a.fn(a)
