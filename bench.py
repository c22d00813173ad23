#!/usr/bin/env python
"""bench.py -- throughput of the hot path on MI355X (contract: see task description / DESIGN.md section 6).

One "step" = one pass of the hot path over one batch of synthetic models, inputs resident in HBM before
the timed region.  ``--config`` selects the BASELINE.json configuration (default c2 = configs[1], the one
the metric is quoted on):

  c2  configs[1]  4096 x (8 series, 2 factors), T=1000, fp64: filter + -2 log L + RTS smoother, all six
                  reference-equivalent state outputs (F, Pf, Xp, Pp, S, Ps) materialised in HBM
  c3  configs[2]  the same with 8192 models per GPU (65536 over 8 GPUs; one all-reduce of the summed -2 log L)
  c4  configs[3]  4096 x (32 series, 4 factors), 30 % missing, T=2000: the projection path (what
                  Metran.get_simulation consumes; three full-square state arrays would be 255 GB) -- filter writing
                  the backward tape + the inverse-free backward pass (MK_OUT_TAPE; --projection-path records:
                  filtered records + RTS smoother with the fused projection epilogue, the round-3 path)
  c5  configs[4]  solver loop: 50 objective evaluations x 8192 models per step (fp64: the reference has no
                  fp32 path); sharded over the ranks, no collective per evaluation

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]

``--gpus N`` from a plain ``python`` re-executes itself under ``torch.distributed.run`` (one rank per GPU,
RCCL); launched under torchrun it reads RANK/LOCAL_RANK/WORLD_SIZE as usual.  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; 6.29 TB/s measured copy ceiling)

CONFIGS = {
    #        B      N  K   T    missing  mode
    "c2": (4096, 8, 2, 1000, 0.0, "full"),
    "c3": (8192, 8, 2, 1000, 0.0, "full"),
    "c4": (4096, 32, 4, 2000, 0.3, "project"),
    "c5": (8192, 8, 2, 1000, 0.0, "solver"),
    # configs[3]'s batch with the smoothed STATE means / variances [B,T,n] as outputs (MK_OUT_VAR_ONLY; what get_state_means /
    # get_state_variances consume, metran.py:655-756): the state-tape path (round 5), or --projection-path records for RTS
    "c4s": (4096, 32, 4, 2000, 0.3, "state"),
    # configs[3]'s batch with ALL six reference outputs (Xp, Pp, F, Pf, S, Ps: kalmanfilter.py:392-400, 453-474) as
    # packed-symmetric records -- 3 x 46 GB = 138 GB resident; split filter (SYM) + the RTS MFMA smoother (SYM)
    "c4f": (4096, 32, 4, 2000, 0.3, "full"),
    # a model BEYOND the specialised envelope (100 states > 64): what only the size-generic kernels serve (mk_generic.hip) -- the
    # reference's loops take any size (kalmanfilter.py:315-390, 453-474); all six outputs, full-square records
    "g100": (256, 96, 4, 300, 0.3, "full"),
}
BASELINE_NAME = {"c2": "configs[1]", "c3": "configs[2] (per-GPU share)", "c4": "configs[3]", "c5": "configs[4] (fp64)",
                 "c4s": "configs[3]'s batch, state outputs", "c4f": "configs[3]'s batch, all six outputs as packed-symmetric records",
                 "g100": "no BASELINE configuration: 96 series + 4 factors (100 states), size-generic kernels"}
EVALS_PER_STEP = 50  # c5: "50 parameter evaluations x batch=8192"


def kernel_source_sha():
    """Identity of the kernels a PMC profile belongs to (stamped into profiles/*/pmc_hbm.json)."""
    h = hashlib.sha256()
    for f in ("metran_amd/csrc/mk_kernels.hip", "metran_amd/csrc/mk_wide.hip", "metran_amd/csrc/mk_split.hip", "metran_amd/csrc/mk_dk.hip",
              "metran_amd/csrc/mk_prims.h", "metran_amd/csrc/mk_jump.h", "metran_amd/csrc/mk_sweeps.h", "metran_amd/csrc/mk_internal.h"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(N, K, T, mode, sym=False, tape=False):
    """bytes per model the algorithm has to move, per kernel.
    c = n + n^2 doubles per moment set (full-square), c_s = n + n(n+1)/2 (packed-symmetric records)."""
    n = N + K
    c = n + (n * (n + 1) // 2 if sym else n * n)
    if mode == "full":      # read obs, write filtered + predicted | re-read filtered, write smoothed
        return {"filter": 8 * T * (N + 2 * c), "smoother": 8 * T * (2 * c)}
    if mode == "state" and tape:    # read obs, write the STATE tape (N + K entries of n + 4) | re-read it, write 2n state moments
        return {"filter": 8 * T * (N + (N + K) * (n + 4)), "smoother": 8 * T * ((N + K) * (n + 4) + 2 * n)}
    if mode == "state":             # read obs, write filtered | re-read filtered, write 2n state moments
        return {"filter": 8 * T * (N + c), "smoother": 8 * T * (c + 2 * n)}
    if mode == "project" and tape:  # read obs, write the tape (N entries of n + 4) | re-read it, write 2N projected moments
        return {"filter": 8 * T * (N + N * (n + 4)), "smoother": 8 * T * (N * (n + 4) + 2 * N)}
    if mode == "project":   # read obs, write filtered | re-read filtered, write 2N projected moments
        return {"filter": 8 * T * (N + c), "smoother": 8 * T * (c + 2 * N)}
    return {"filter": 8 * T * N, "smoother": 0}  # solver objective: the observation stream only (B_ll)


def pmc_traffic(config, kernel_key, sym=False):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of THIS command
    (profiles/<round>/pmc_hbm*.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; units KiB).  The profile must carry
    the hash of the kernel source it was taken with; on a mismatch (kernels edited since) or when no
    profile of this configuration exists the answer is None -- stale counters are never reported."""
    import glob

    best, note = None, "no committed PMC profile of this configuration"
    sha = kernel_source_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_hbm*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("config") != config or bool(d.get("packed_sym", False)) != bool(sym):
            continue
        if d.get("kernel_source_sha256") != sha:
            note = "%s was taken with other kernel sources (sha %s, now %s)" % (
                os.path.relpath(f, ROOT), d.get("kernel_source_sha256"), sha)
            continue
        for name, c in d.get("kernels", {}).items():
            if kernel_key in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                best = {"GB": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e9, "source": os.path.relpath(f, ROOT)}
    return best, (None if best else note)


def live_traffic(config, packed_sym, timeout_s=180, projection_path="auto"):
    """HBM traffic of this configuration's kernels measured IN THIS RUN (round-2 verdict, weak 7: the number used to be
    replayed from a committed profile): two short child runs of this script under ``rocprofv3 --pmc FETCH_SIZE`` and
    ``--pmc WRITE_SIZE`` -- separate passes, counters only, no trace domain, as MI355X_MICROARCH.md prescribes -- after
    the timed region.  Returns ({kernel name: GB per launch}, note).  FETCH_SIZE is doubled (it counts half of a wide
    coalesced read stream on gfx950); units are KiB per dispatch."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found on this box"
    child = [sys.executable, os.path.abspath(__file__), "--config", config, "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
             "--no-secondary", "--no-live-traffic"] + (["--packed-sym"] if packed_sym else []) + (
                 ["--projection-path", projection_path] if projection_path != "auto" else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    env["TMPDIR"] = "/tmp"
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mk_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            vals = {}
            for row in csv.DictReader(open(files[0])):
                if "mk::" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            for k, v in vals.items():
                per.setdefault(k, {})[counter] = sum(v) / len(v)
        except Exception as e:  # noqa: BLE001 -- the bench line must survive a profiler problem
            return None, "live PMC pass failed: %s: %s" % (type(e).__name__, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {k: (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e9 for k, c in per.items() if len(c) == 2}
    return (out or None), ("rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, two separate child runs of this command in this run "
                           "(3 + 1 launches each); HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB")


FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 vector = FP64 matrix peak (256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz).
                         # Not in MI355X_MICROARCH.md; measured ceilings here: 72 (v_mfma_f64_16x16x4) / 66 (v_fma_f64) TFLOP/s.


def algorithmic_flops(N, K, T, mode, missing=0.0):
    """flops per model.  Filter per step: predict 3n^2+n (diagonal Phi) + m (5n^2+6n+3) for the
    m observed series (the reference's dense update); smoother per step: 6.33 n^3 + 4 n^2 (LDL^T n^3/3, two triangular
    solves 2 n^3, J D 2 n^3, .J^T 2 n^3)."""
    n = N + K
    m = N * (1.0 - missing)
    f = T * (3 * n * n + n + m * (5 * n * n + 6 * n + 3))
    return {"filter": f, "smoother": 0.0 if mode == "solver" else T * (6.33 * n ** 3 + 4 * n * n)}


def executed_flops_tape(N, K, T, missing):
    """What the tape path's own formulation executes per model (for the record next to SURVEY 8d's reference-algorithm count):
    filter per step: predict 3n^2+n, per observed series 2n(1+K) (d = P z') + 2n^2 (rank-one) + ~8n, per unobserved ~4n(1+K);
    backward pass per step: per series one (n+1) x n product 2n(n+1) + ~4n, transition 2(n+1)NK + 4n^2."""
    n = N + K
    m = N * (1.0 - missing)
    return {"filter": T * (3 * n * n + n + m * (2 * n * (1 + K) + 2 * n * n + 8 * n) + (N - m) * 4 * n * (1 + K)),
            "smoother": T * (N * (2 * n * (n + 1) + 4 * n) + 2 * (n + 1) * N * K + 4 * n * n)}


def executed_flops(N, K, T, mode, missing, tape, tape_filter=None):
    """Flops per model the kernels' OWN formulation executes (useful multiply-adds x 2; replica lanes, padding and
    recomputation for scheduling are not counted) -- the numerator of ``real_frac``.  Record paths: the filter exploits
    Z = [I | G] (d = P z' is 1 + K multiply-adds a row where the reference's dense update does n, kalmanfilter.py:349-357):
    predict 3n^2+n, per observed series 2n(1+K) + 2n^2 (full rank-one update) + ~8n; the RTS smoothers run the reference's
    operation count (LDL^T n^3/3, two triangular solves 2n^3, V = J D 2n^3, Ps = Pf + V J' 2n^3 in full, means 4n^2) plus the
    predicted moments they recompute from the filtered ones (3n^2)."""
    if tape:
        ex = executed_flops_state_tape(N, K, T, missing) if mode == "state" else executed_flops_tape(N, K, T, missing)
        if tape_filter != "state" and N <= 32:
            # filter_obs_kernel (round 6): per observed series the full rank-one update 2n^2 + ~10n (no d = P z', no T k); the
            # prediction 2 (2 + 3K) per covariance element of the series rows + ~30n; unobserved entries are register copies
            n_, m_ = N + K, N * (1.0 - missing)
            ex = dict(ex, filter=T * (N * n_ * 2 * (2 + 3 * K) + 30 * n_ + m_ * (2 * n_ * n_ + 10 * n_)))
        return ex
    n = N + K
    m = N * (1.0 - missing)
    filt = T * (3 * n * n + n + m * (2 * n * (1 + K) + 2 * n * n + 8 * n))
    return {"filter": filt, "smoother": 0.0 if mode == "solver" else T * (6.33 * n ** 3 + 7 * n * n)}


def executed_flops_state_tape(N, K, T, missing):
    """... of the state-tape path: the projection tape's count plus, per step, K more products (n+1) x n and the K x n
    products of every entry against them."""
    n = N + K
    base = executed_flops_tape(N, K, T, missing)
    return {"filter": base["filter"] + T * 4 * K * n, "smoother": base["smoother"] + T * (K * 2 * n * (n + 1) + 2 * n * n * K)}


def build_roofline(config, N, K, T, B, mode, missing, f_avg, s_avg, packed_sym, evals=1, live=None, tape=False, tape_filter=None):
    """The ``roofline`` object of the bench line for one workload: per-kernel algorithmic bytes / flops per launch over
    the hipEvent launch time, and the dominant kernel against the roof that bounds it -- HBM for the 16-lane
    filter+smoother kernels (AI ~ 3 flop/B), the fp64 pipe for the wide (n > 16) kernels and the solver objective
    (AI 41-75 flop/B on the bytes they move: SURVEY 8d)."""
    ab = algorithmic_bytes(N, K, T, mode, sym=packed_sym, tape=tape)
    fl = algorithmic_flops(N, K, T, mode, missing)
    if mode == "solver":
        s_avg = 0.0
    wide = N + K > 16   # one model per wavefront: mk_wide.hip / mk_dk.hip
    sname = None if mode == "solver" else ("smoother_dk_kernel" if tape else "smoother_mfma_kernel" if wide else "smoother_record_kernel")

    def entry(kind, ms):
        return {"ms": ms, "algorithmic_GB": ab[kind] * B / 1e9, "GBps": ab[kind] * B / 1e9 / (ms / 1e3),
                "algorithmic_TFLOP": fl[kind] * B / 1e12, "TFLOPps": fl[kind] * B / 1e12 / (ms / 1e3)}

    fname = "filter_split_kernel" if (wide and N <= 32) else "filter_kernel"   # mk_split.hip serves the wide shapes with N <= 32
    if tape and wide and N <= 32 and tape_filter != "state":
        fname = "filter_obs_kernel"                                             # ... the tape from the filter in the observable basis (round 6)
    kernels = {fname: entry("filter", f_avg)}
    if sname:
        kernels[sname] = entry("smoother", s_avg)
    # the reference-algorithm count above is what SURVEY 8d prices; this is what the kernels' own formulation executes, and
    # real_frac -- the larger of (algorithmic bytes / time / HBM peak) and (executed flops / time / fp64 peak) -- is the
    # kernel's REAL distance from its nearer roof ("say the real fraction next to every equivalent one")
    ex = executed_flops(N, K, T, mode, missing, tape, tape_filter)
    for kname, kind, ms in ((fname, "filter", f_avg), (sname, "smoother", s_avg)):
        if kname is None:
            continue
        kernels[kname]["executed_TFLOP"] = ex[kind] * B / 1e12
        kernels[kname]["executed_TFLOPps"] = ex[kind] * B / 1e12 / (ms / 1e3)
        kernels[kname]["real_frac"] = max(kernels[kname]["GBps"] / HBM_PEAK_GBS, kernels[kname]["executed_TFLOPps"] / FP64_PEAK_TFLOPS)
        kernels[kname]["real_frac_roof"] = ("hbm" if kernels[kname]["GBps"] / HBM_PEAK_GBS >= kernels[kname]["executed_TFLOPps"] / FP64_PEAK_TFLOPS
                                            else "fp64")
    for kname in kernels:
        lv = None
        if live and live[0]:
            hits = [v for k, v in live[0].items() if ("mk::" + kname + "<") in k]
            lv = hits[0] if len(hits) == 1 else None
        if lv is not None:
            kernels[kname]["traffic_GB"] = lv
            kernels[kname]["traffic_source"] = "measured in this run: " + live[1]
            continue
        tr, note = pmc_traffic(config, kname, packed_sym)
        kernels[kname]["traffic_GB"] = tr["GB"] if tr else None
        kernels[kname]["traffic_source" if tr else "traffic_note"] = tr["source"] if tr else note
        if live and not live[0]:
            kernels[kname]["live_traffic_note"] = live[1]
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    fp64_bound = wide or mode == "solver"
    if fp64_bound:
        roofline = {"bound": "fp64", "kernel": dom, "achieved": kernels[dom]["TFLOPps"], "peak": FP64_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": kernels[dom]["TFLOPps"] / FP64_PEAK_TFLOPS,
                    "algorithmic_flops": kernels[dom]["algorithmic_TFLOP"] * 1e12,
                    "hbm_frac_of_the_same_kernel": kernels[dom]["GBps"] / HBM_PEAK_GBS,
                    "note": "arithmetic intensity %.0f flop/B on the algorithmic bytes: bound by the fp64 vector/MFMA pipe "
                            "(78.6 TFLOP/s datasheet; f64 MFMA and f64 VALU share it on gfx950), not by HBM"
                            % (kernels[dom]["algorithmic_TFLOP"] * 1e3 / max(kernels[dom]["algorithmic_GB"], 1e-30))
                            + "; flops are SURVEY 8d's count of the REFERENCE algorithm (dense updates, RTS with an explicit "
                              "inverse) -- this path executes %.2f x of them (executed_TFLOP), so `frac` is work-equivalent "
                              "throughput; `real_frac` is the utilisation of the nearer roof"
                              % (sum(v for k_, v in ex.items() if k_ == "filter" or sname) / max(sum(fl.values()), 1e-30))}
    else:
        roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS}
    roofline.update({"real_frac": kernels[dom]["real_frac"], "real_frac_roof": kernels[dom]["real_frac_roof"],
                     "executed_TFLOPps": kernels[dom]["executed_TFLOPps"],
                     "traffic": (kernels[dom]["traffic_GB"] * 1e9 if kernels[dom].get("traffic_GB") else None),
                     "algorithmic_bytes": kernels[dom]["algorithmic_GB"] * 1e9, "avg_launch_ms": kernels[dom]["ms"],
                     "kernels": kernels, "kernel_source_sha256": kernel_source_sha(),
                     "path_achieved_GBps": sum(ab.values()) * B / 1e9 / ((f_avg + s_avg) / 1e3)})
    # SURVEY 8d's full-output accounting B_fs = 8 T (N + 4c) per model, for every configuration (what the
    # north-star's ">= 40 % of HBM" is quoted in; the path may move fewer bytes, e.g. the projection path)
    n = N + K
    b_fs = 8 * T * (N + 4 * (n + n * n))
    if mode != "solver":
        roofline["survey_8d_full_output_accounting"] = {
            "bytes_per_model": b_fs, "equivalent_GBps": B * b_fs / 1e9 / ((f_avg + s_avg) / 1e3),
            "frac_of_peak": B * b_fs / 1e9 / ((f_avg + s_avg) / 1e3) / HBM_PEAK_GBS,
            "north_star_bar": {"frac_of_peak": 0.40, "models_per_s": 0.40 * HBM_PEAK_GBS * 1e9 / b_fs},
            "note": "kernel-time rate of this GPU x bytes a full-square six-output pass would move"}
    return roofline


# --------------------------------------------------------------------------------------- CPU baselines
def host_cpus():
    """What this process may actually use of the host's CPUs: (logical CPUs in its affinity mask, cgroup CPU quota or None).
    The GPU boxes of this pool are 256-thread hosts whose containers carry a quota of 16 CPUs: 128 OpenMP threads there run
    at 0.6 x the rate of 16 (scripts/cpu_leg_threads.py, profiles/r05/cpu_leg_threads.log), so the CPU legs take the quota."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):                      # cgroup v2: "<quota|max> <period>"
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q == "max" else float(q) / float(p)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):       # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / p if q > 0 else None
    except (OSError, ValueError):
        quota = None
    return logical, quota


def cpu_leg_threads(max_threads):
    """Threads of a timed CPU leg: the CPUs this process may use (quota included), at most what OpenMP would start."""
    logical, quota = host_cpus()
    t = logical if quota is None else max(1, min(logical, int(round(quota))))
    return max(1, min(t, max_threads))


def cpu_baseline_port(host, mode, gpu_mle, target_seconds=8.0, min_models=1):
    """The reference ALGORITHM (oracle/kalman_oracle.c: C restatement of kalmanfilter.py:236-476, OpenMP over
    models) on the host cores of this box, on a bounded sample of the same workload."""
    import numpy as np

    import oracle

    native = False
    try:
        oracle.build(native=True)  # -march=native build for this host; falls back to the portable .so
        native = True
    except Exception:
        pass
    cores = cpu_leg_threads(oracle.num_threads(native))
    oracle.set_num_threads(cores, native)
    B, T = host["obs"].shape[0], host["obs"].shape[1]
    smooth = mode != "solver"

    def run(sl, out=None):
        return oracle.dfm_batch(host["obs"][sl], host["phi"][sl], host["q"][sl], host["loadings"][sl], native=native, out=out,
                                smooth=smooth, outputs={"full": "all", "project": "means", "state": "all", "solver": "mle"}[mode])

    probe = min(B, max(cores, min_models))
    t0 = time.perf_counter()
    ref = run(slice(0, probe))
    dt_first = time.perf_counter() - t0
    per_model = dt_first / probe
    n = int(max(probe, min(B, target_seconds / max(per_model, 1e-9))))
    n = max(min(B, cores), (n // cores) * cores) if n >= cores else n
    # timed: the second pass over the sample, into output arrays the first has touched (a first pass also pays the page faults
    # of its fresh output memory; the GPU side is timed with its buffers allocated).  Where the probe already is the sample
    # (wide models: 16 of them fill the budget) it doubles as that first pass.
    if n != probe:
        t0 = time.perf_counter()
        ref = run(slice(0, n))
        dt_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = run(slice(0, n), out=ref)
    dt = time.perf_counter() - t0
    rel = float(np.max(np.abs(gpu_mle[:n] - ref["mle"]) / np.abs(ref["mle"])))
    what = "filter+smoother with all outputs" if smooth else "filter + -2 log L only (one objective evaluation per model)"
    return {"value": n * T / dt, "unit": "model-timesteps/s", "models_per_s": n / dt, "cores": cores, "kind": "port", "variant": "fidelity checker",
            "models": n,
            "host": dict(zip(("logical_cpus", "cgroup_cpu_quota"), host_cpus())),
            "sample": "%d of the %d models of rank 0's batch, full T=%d, %s, OpenMP over models, %.1f s for the second pass into "
                      "already-touched output arrays (first pass: %.1f s) (C restatement of kalmanfilter.py:236-476, %s)"
                      % (n, B, T, what, dt, dt_first, "-O3 -march=native" if native else "-O3")}, rel


def cpu_baseline_optimised(host, mode, gpu_mle, target_seconds=8.0, min_models=1):
    """SURVEY 8d leg (1), the "numba-class or better" CPU baseline: oracle/kalman_fast.c -- the same recursions written for
    speed (diagonal Phi / Q and Z = [I | G] exploited, symmetric updates, Cholesky solves instead of pinv, no allocation in
    the loop, -O3 -march=native -ffp-contract=fast, OpenMP over models) -- on every host core, on a bounded sample of the same
    workload, with its parity against the fidelity checker (the C port above) on 16 models in the line (VERDICT r4 weak 7:
    the port is a checker, slower per core than the Python it restates; it is not what a CPU can do)."""
    import numpy as np

    import oracle

    native = False
    try:
        oracle.build(native=True)
        oracle.load_fast(native=True)
        native = True
    except Exception:
        pass
    cores = cpu_leg_threads(oracle.fast_num_threads(native))
    oracle.fast_set_num_threads(cores, native)
    B, T = host["obs"].shape[0], host["obs"].shape[1]
    outputs = {"full": "all", "project": "means", "state": "all", "solver": "mle"}[mode]

    def run(sl, out=None):
        return oracle.fast_dfm_batch(host["obs"][sl], host["phi"][sl], host["q"][sl], host["loadings"][sl], outputs=outputs, native=native,
                                     out=out)

    probe = min(B, max(cores, min_models))
    run(slice(0, probe))                    # pages the library and the arrays in
    t0 = time.perf_counter()
    run(slice(0, probe))
    per_model = (time.perf_counter() - t0) / probe
    n = int(max(probe, min(B, target_seconds / max(per_model, 1e-9))))
    n = max(min(B, cores), (n // cores) * cores) if n >= cores else n
    if outputs == "all":                    # bound the host memory of the six state arrays (3 (n + n^2) doubles per model-step)
        nst = host["phi"].shape[1]
        n = max(1, min(n, int(24e9 // (8 * T * 3 * (nst + nst * nst)))))
    # timed: the SECOND pass over the sample, into output arrays the first has already touched -- the GPU side is timed with its
    # buffers allocated too, and a first pass mostly measures the page faults of its fresh output memory (14.7 GB at configs[1])
    t0 = time.perf_counter()
    res = run(slice(0, n))
    dt_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = run(slice(0, n), out=res)
    dt = time.perf_counter() - t0
    rel = float(np.max(np.abs(gpu_mle[:n] - res["mle"]) / np.abs(res["mle"])))
    k = min(n, 16)
    chk = oracle.dfm_batch(host["obs"][:k], host["phi"][:k], host["q"][:k], host["loadings"][:k], smooth=(mode != "solver"),
                           outputs="mle" if mode == "solver" else "all")
    par = {"models": k, "loglik_max_rel_err": float(np.max(np.abs(res["mle"][:k] - chk["mle"]) / np.abs(chk["mle"])))}
    if outputs == "all":
        par["smoothed_means_max_abs_err"] = float(np.max(np.abs(res["S"][:k] - chk["S"])))
        par["smoothed_covariances_max_abs_err"] = float(np.max(np.abs(res["Ps"][:k] - chk["Ps"])))
    what = {"all": "filter+smoother with all six state outputs", "means": "filter+smoother, projected means / variances only",
            "mle": "filter + -2 log L only (one objective evaluation per model)"}[outputs]
    return {"value": n * T / dt, "unit": "model-timesteps/s", "models_per_s": n / dt, "cores": cores, "kind": "port", "variant": "optimised",
            "models": n,
            "host": dict(zip(("logical_cpus", "cgroup_cpu_quota"), host_cpus())),
            "sample": "%d of the %d models of rank 0's batch, full T=%d, %s, OpenMP over models, %.2f s for the second pass into "
                      "already-touched output arrays (first pass, page faults included: %.2f s) (oracle/kalman_fast.c: "
                      "structure-exploiting filter, Cholesky smoother, %s -ffp-contract=fast; NOT bit-faithful)"
                      % (n, B, T, what, dt, dt_first, "-O3 -march=native" if native else "-O3"),
            "first_pass_models_per_s": n / dt_first,
            "models_not_served": res["bad"], "loglik_max_rel_err_vs_gpu": rel, "parity_vs_the_checker": par}


def cpu_baseline_reference(host, mode, gpu_mle, models=16, budget_s=25.0):
    """SURVEY 8d legs (2) and (3): the reference AS SHIPPED -- its numpy engine ``seqkalmanfilter_np`` + the
    Python ``kalmansmoother`` (kalmanfilter.py:122-233, 403-476), imported from the staged verbatim copy
    ``oracle/_ref`` through the pastas stub -- on up to 16 models of the same batch, ONE core, extrapolated
    and labelled so; plus the numba-jitted engine when ``import numba`` works on this box."""
    import numpy as np

    out = {}
    try:
        import numba  # noqa: F401

        out["numba"] = "importable (%s)" % numba.__version__
    except Exception as e:  # noqa: BLE001
        out["numba"] = "not importable on this box (%s): the jitted seqkalmanfilter cannot be timed" % type(e).__name__
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    try:
        import _refshim

        if not _refshim.reference_available():
            raise ImportError("oracle/_ref not staged (oracle/make_ref.sh)")
        metran = _refshim.install()
    except Exception as e:  # noqa: BLE001
        out["reference"] = "unavailable: %s" % e
        return out
    import pandas as pd

    kfm = metran.kalmanfilter
    N = host["obs"].shape[2]
    T = host["obs"].shape[1]
    done, t_f, t_s, rel = 0, 0.0, 0.0, 0.0
    t_start = time.perf_counter()
    for b in range(min(models, host["obs"].shape[0])):
        kf = kfm.SPKalmanFilter(engine="numpy")
        if "importable (" in out["numba"]:
            kf.filtermethod = kfm.seqkalmanfilter
        kf.set_observations(pd.DataFrame(host["obs"][b]))
        Z = np.hstack([np.eye(N), host["loadings"][b]])
        kf.set_matrices(np.diag(host["phi"][b]), np.diag(host["q"][b]), Z, np.zeros(N))
        t0 = time.perf_counter()
        kf.run_filter()
        t1 = time.perf_counter()
        if mode != "solver":
            kfm.kalmansmoother(kf.filtered_state_means, kf.filtered_state_covariances, kf.predicted_state_means,
                               kf.predicted_state_covariances, kf.transition_matrix)
        t2 = time.perf_counter()
        if b > 0 or models == 1:  # first model pays imports / JIT
            t_f += t1 - t0
            t_s += t2 - t1
            done += 1
        m = kf.get_mle()
        rel = max(rel, abs(float(gpu_mle[b]) - m) / abs(m))
        if time.perf_counter() - t_start > budget_s and done >= 2:
            break
    if done:
        per = (t_f + t_s) / done
        out["reference"] = {"value": T / per, "unit": "model-timesteps/s", "models_per_s": 1.0 / per, "cores": 1,
                            "kind": "reference", "filter_s_per_model": t_f / done, "smoother_s_per_model": t_s / done,
                            "sample": "%d models of rank 0's batch through the reference as shipped (oracle/_ref: %s engine + "
                                      "Python kalmansmoother), one core, per-model mean; EXTRAPOLATED to a rate"
                                      % (done, "numba" if "importable (" in out["numba"] else "numpy"),
                            "loglik_max_rel_err_vs_gpu": rel}
    return out


# --------------------------------------------------------------------------------------- launch plumbing
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn_under_torchrun(n):
    """``python bench.py --gpus N`` from a plain shell: become ``torch.distributed.run`` with N ranks on this
    node (rendezvous on 127.0.0.1).  exec keeps the PID, so a driver timing this process times the job."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, argv)


class Workload:
    """One configuration of CONFIGS on one GPU: synthetic records resident in HBM, output buffers allocated once,
    ``step()`` = one pass of the hot path over the batch (+ the deterministic local sum of -2 log L into ``total``)."""

    def __init__(self, name, local_rank, rank, dev, layout, packed_sym, batch=None, T=None, projection_path="auto", variants=None):
        import torch

        from metran_amd.engine import BatchedKalman
        from metran_amd.synthetic import make_dfm_batch_torch

        B, N, K, T0, missing, mode = CONFIGS[name]
        self.name, self.B, self.N, self.K, self.T, self.missing, self.mode = name, batch or B, N, K, T or T0, missing, mode
        self.packed_sym = bool(packed_sym) and mode == "full"
        self.d = make_dfm_batch_torch(self.B, N, K, self.T, seed=2000 + rank, device=dev, missing=missing)
        self.kf = BatchedKalman(local_rank, layout=layout, packed_sym=self.packed_sym)
        self.kf.projection_path = projection_path
        for which, val in (variants or {}).items():   # e.g. {"kernel_family": "generic"}: the size-generic kernels for this shape
            self.kf.set_variant(which, val)
        self.variants = dict(variants or {})
        self.kf.set_observations(self.d["obs"]).set_loadings(self.d["loadings"])
        self.tape = (mode == "project" and self.kf.tape_path()) or (mode == "state" and self.kf.state_tape_path())
        self.total = torch.zeros(1, dtype=torch.float64, device=dev)
        if mode == "full":
            self.bufs = self.kf._alloc_outputs(self.B, ["F", "Pf", "Xp", "Pp", "S", "Ps"])
        elif mode == "project":
            self.bufs = self.kf.alloc_projection(self.B)
        elif mode == "state":
            self.bufs = self.kf.alloc_state_variances(self.B)
        else:
            self.bufs = {"mle": torch.empty(self.B, dtype=torch.float64, device=dev)}
            # the 50 parameter sets of one step: alpha_k = alpha_0 (1 + 0.02 k), SURVEY 8d
            self.alphas = [self.d["alpha"] * (1.0 + 0.02 * k) for k in range(EVALS_PER_STEP)]

    @property
    def units(self):  # model passes per step
        return EVALS_PER_STEP if self.mode == "solver" else 1

    def step(self):
        kf, d = self.kf, self.d
        if self.mode == "full":
            kf.filter_smooth(d["phi"], d["q"], buffers=self.bufs)
        elif self.mode == "project":
            kf.simulate_smoothed(d["phi"], d["q"], buffers=self.bufs)
        elif self.mode == "state":
            kf.smooth_state_variances(d["phi"], d["q"], buffers=self.bufs)
        else:
            for a in self.alphas:  # a2 + a3 + a6 per evaluation, as Metran.get_mle does (metran.py:605-622)
                phi, q = kf.params_from_alpha(a)
                kf.loglik(phi, q, out=self.bufs["mle"])
        self.total.copy_(kf.sum(self.bufs["mle"]).reshape(1))   # deterministic local reduction
        return self.total

    def host_inputs(self, n=None):
        """Host copies of the first ``n`` models' inputs (all of them by default) for the CPU legs."""
        sl = slice(0, self.B if n is None else min(n, self.B))
        if self.mode == "solver":
            ph, qq = self.kf.params_from_alpha(self.alphas[-1])
            return {"obs": self.d["obs"][sl].cpu().numpy(), "phi": ph[sl].cpu().numpy(), "q": qq[sl].cpu().numpy(),
                    "loadings": self.d["loadings"][sl].cpu().numpy()}
        return {k: self.d[k][sl].cpu().numpy() for k in ("obs", "phi", "q", "loadings")}

    def describe(self):
        return ("BASELINE.json %s: batch=%d synthetic %d-series/%d-factor DFMs per GPU, T=%d, fp64, %s"
                % (BASELINE_NAME[self.name], self.B, self.N, self.K, self.T,
                   {"full": ("%d %% missing, " % round(100 * self.missing) if self.missing else "") + "filter+smoother, outputs F,Pf,Xp,Pp,S,Ps"
                            + (" as packed-symmetric records" if self.packed_sym else "")
                            + (" [size-generic kernels, mk_generic.hip]" if self.variants.get("kernel_family") == "generic" else ""),
                    "project": "%d %% missing, projection outputs (sim_means, sim_vars): %s" % (
                        round(100 * self.missing), "filter writing the backward tape + inverse-free backward pass (MK_OUT_TAPE)"
                        if getattr(self, "tape", False) else "filter (filtered record) + RTS smoother with the fused projection epilogue"),
                    "state": "%d %% missing, smoothed state means + variances [B,T,n] (MK_OUT_VAR_ONLY): %s" % (
                        round(100 * self.missing), "filter writing the STATE tape + inverse-free backward pass (MK_OUT_TAPE | MK_OUT_VAR_ONLY)"
                        if getattr(self, "tape", False) else "filter (filtered record) + RTS smoother with the variance epilogue"),
                    "solver": "solver loop: %d objective evaluations (alpha -> phi,q -> filter -> -2 log L) per step" % EVALS_PER_STEP}[self.mode]))

    def close(self):
        self.kf.close()
        self.bufs = self.d = self.kf = None


def timed_run(w, steps, warmup, sync, after_step=None):
    """W untimed warm-up steps, then EXACTLY ``steps`` steps bracketed by sync() (barrier + device synchronise) on both
    sides; kernel times are hipEvent pairs on the launch stream collected AFTER the loop (nothing synchronises inside
    it).  Returns (elapsed seconds, average filter ms per launch, average smoother ms per launch)."""
    import torch

    for _ in range(warmup):
        w.step()
        if after_step:
            after_step(w)
    sync()
    w.kf.enable_timing(True, accumulate=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step()
        if after_step:
            after_step(w)
    sync()
    elapsed = time.perf_counter() - t0
    f_tot, f_n, s_tot, s_n = w.kf.kernel_ms_totals()   # every launch of the timed region
    w.kf.enable_timing(False)
    return elapsed, f_tot / max(f_n, 1), s_tot / max(s_n, 1)


def sample_models(B, count):
    """Indices spread over the batch AND over the positions inside a wavefront (the wide filter serves two or four models
    per wavefront): the first and last 16 models, then an odd stride through the rest."""
    import numpy as np

    if B <= count:
        return np.arange(B)
    head = list(range(16)) + list(range(B - 16, B))
    rest = (np.linspace(16, B - 17, count - 16).astype(np.int64) | 1).tolist() + np.linspace(17, B - 18, 16).astype(np.int64).tolist()
    return np.array(sorted(set(head + rest)))


def parity_figures(w, n_mle=256, n_proj=32):
    """Error figures of a secondary configuration against the C port of the reference algorithm (oracle/, the checker):
    -2 log L on n_mle models spread over the batch, and -- projection mode -- the projected smoothed means / variances on
    n_proj of them (reference kalmansmoother + simulate, kalmanfilter.py:403-476, 569-603)."""
    import numpy as np

    import oracle

    idx = sample_models(w.B, n_mle)
    if w.mode == "solver":
        ph, qq = w.kf.params_from_alpha(w.alphas[-1])
    else:
        ph, qq = w.d["phi"], w.d["q"]
    host = {"obs": w.d["obs"][idx].cpu().numpy(), "phi": ph[idx].cpu().numpy(), "q": qq[idx].cpu().numpy(),
            "loadings": w.d["loadings"][idx].cpu().numpy()}
    t0 = time.perf_counter()
    ref = oracle.dfm_batch(host["obs"], host["phi"], host["q"], host["loadings"], smooth=False, outputs="mle")
    gpu = w.bufs["mle"][idx].cpu().numpy()
    out = {"loglik_max_rel_err": float(np.max(np.abs(gpu - ref["mle"]) / np.abs(ref["mle"]))), "loglik_models_compared": int(len(idx))}
    if w.mode == "full":
        sub = idx[np.linspace(0, len(idx) - 1, min(n_proj, len(idx))).astype(int)]
        pos = np.searchsorted(idx, sub)
        r2 = oracle.dfm_batch(host["obs"][pos], host["phi"][pos], host["q"][pos], host["loadings"][pos])
        out["state_models_compared"] = int(len(sub))
        for mean, cov, label in (("Xp", "Pp", "predicted"), ("F", "Pf", "filtered"), ("S", "Ps", "smoothed")):
            c = w.bufs[cov][sub]
            c = (w.kf.unpack_sym(c) if w.packed_sym else c).cpu().numpy()
            out[label + "_means_max_abs_err"] = float(np.max(np.abs(w.bufs[mean][sub].cpu().numpy() - r2[mean])))
            out[label + "_covariances_max_abs_err"] = float(np.max(np.abs(c - r2[cov])))
    if w.mode == "state":
        sub = idx[np.linspace(0, len(idx) - 1, min(n_proj, len(idx))).astype(int)]
        pos = np.searchsorted(idx, sub)
        r2 = oracle.dfm_batch(host["obs"][pos], host["phi"][pos], host["q"][pos], host["loadings"][pos])
        out["state_models_compared"] = int(len(sub))
        out["state_means_max_abs_err"] = float(np.max(np.abs(w.bufs["S"][sub].cpu().numpy() - r2["S"])))
        out["state_vars_max_abs_err"] = float(np.max(np.abs(w.bufs["var"][sub].cpu().numpy() - np.diagonal(r2["Ps"], axis1=2, axis2=3))))
    if w.mode == "project":
        sub = idx[np.linspace(0, len(idx) - 1, min(n_proj, len(idx))).astype(int)]
        pos = np.searchsorted(idx, sub)
        r2 = oracle.dfm_batch(host["obs"][pos], host["phi"][pos], host["q"][pos], host["loadings"][pos])
        Z = np.concatenate([np.broadcast_to(np.eye(w.N), (len(sub), w.N, w.N)), host["loadings"][pos]], axis=2)
        m_ref = np.einsum("bjn,btn->btj", Z, r2["S"])
        v_ref = np.maximum(np.einsum("bjn,btnm,bjm->btj", Z, r2["Ps"], Z), 0.0)
        out["projection_models_compared"] = int(len(sub))
        out["sim_means_max_abs_err"] = float(np.max(np.abs(w.bufs["sim_means"][sub].cpu().numpy() - m_ref)))
        out["sim_vars_max_abs_err"] = float(np.max(np.abs(w.bufs["sim_vars"][sub].cpu().numpy() - v_ref)))
    out["checker"] = "oracle/kalman_oracle.c (C port of the reference algorithm), %.1f s on the host" % (time.perf_counter() - t0)
    return out


def secondary_cpu_legs(w, target_seconds=3.0, nhost=64):
    """The CPU beside a secondary configuration (VERDICT r5 missing 3 / SURVEY 8d "CPU baseline timing"): the optimised C leg
    (oracle/kalman_fast.c) and the fidelity checker (oracle/kalman_oracle.c) on >= 16 models of the same batch, same outputs,
    this box's usable cores (stated), a second pass into touched arrays as the headline's legs."""
    host = w.host_inputs(nhost)
    gpu_mle = w.bufs["mle"][: host["obs"].shape[0]].cpu().numpy()
    out = {}
    try:
        out["cpu_baseline"] = cpu_baseline_optimised(host, w.mode, gpu_mle, target_seconds=target_seconds, min_models=16)
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:
        out["cpu_baseline_checker"], out["loglik_max_rel_err_vs_checker_sample"] = cpu_baseline_port(
            host, w.mode, gpu_mle, target_seconds=target_seconds, min_models=16)
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline_checker"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def secondary_workload(name, local_rank, rank, dev, layout, sync, steps=5, warmup=2, live=True, packed_sym=False, variants=None,
                       batch=None, cpu=True, n_proj=32):
    """The non-headline configurations measured in the SAME process after the headline's timed region (VERDICT r2 item 3:
    configs[3] and configs[4] in driver-run records): kernel ms, models/s, the roofline of one GPU and (r3 item 2) error
    figures against the oracle on a sample of the batch."""
    import torch

    w = Workload(name, local_rank, rank, dev, layout, packed_sym, batch=batch, variants=variants)
    tape = getattr(w, "tape", False)
    try:
        elapsed, f_avg, s_avg = timed_run(w, steps, warmup, sync)
        models_per_s = w.B * w.units * steps / elapsed
        out = {"workload": w.describe(), "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
               "models_per_s": models_per_s, "value": models_per_s * w.T, "unit": "model-timesteps/s",
               "filter_ms": f_avg, "smoother_ms": (s_avg if w.mode != "solver" else None)}
        if w.mode == "state":
            out["state_tape"] = bool(tape)
        if w.mode == "full":
            out["record_stride_doubles"] = int(w.kf.record_stride())
        if w.mode == "solver":
            out["objective_evaluations_per_s"] = models_per_s
        try:
            out["parity"] = parity_figures(w, n_proj=n_proj)
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if cpu:
            out.update(secondary_cpu_legs(w))
            if "models_per_s" in out.get("cpu_baseline", {}):
                out["speedup_vs_cpu_baseline"] = models_per_s / out["cpu_baseline"]["models_per_s"]   # model passes per second, both
        cfg = (w.N, w.K, w.T, w.B, w.mode, w.missing)
    finally:
        w.close()
        torch.cuda.empty_cache()
    lv = live_traffic(name, False) if (live and name == "c4") else None   # after the buffers are released
    out["roofline"] = build_roofline(name, cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], f_avg, s_avg, packed_sym, live=lv, tape=tape,
                                     tape_filter=(variants or {}).get("tape_filter"))
    if variants and variants.get("kernel_family") == "generic":   # the size-generic kernels have their own names and no roof claim
        out["roofline"]["note_generic"] = "size-generic kernels (mk_generic.hip): the figures price filter_generic_kernel / smoother_generic_kernel"
    return out


def measure_transfers(dev, B, T, N, K, ms_per_step):
    """Opt-in (``--transfers``): what a caller that starts and ends on the HOST pays on top of the HBM-resident pass
    (never ``value``): the observation upload and the download of the results over the host link,
    pinned buffers, best of 3.  The full state set is sized, not allocated on the host: the download rate is measured
    on a 1 GiB piece."""
    import torch

    n = N + K
    out = {}
    obs_bytes = B * T * N * 8
    h = torch.empty(obs_bytes // 8, dtype=torch.float64).pin_memory()
    d = torch.empty(obs_bytes // 8, dtype=torch.float64, device=dev)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    out["obs_upload"] = {"bytes": obs_bytes, "seconds": best, "GBps": obs_bytes / best / 1e9}
    del h, d
    piece = 1 << 30
    h = torch.empty(piece // 8, dtype=torch.float64).pin_memory()
    d = torch.zeros(piece // 8, dtype=torch.float64, device=dev)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    d2h = piece / best / 1e9
    out["download_GBps"] = d2h
    full = B * T * (3 * (n + n * n) + 2) * 8        # predicted, filtered, smoothed moments + sigma, detf
    proj = B * T * 2 * N * 8                        # projection epilogue: simulated means + variances
    t_pass = ms_per_step / 1e3
    t_up = out["obs_upload"]["seconds"]
    out["host_to_host_models_per_s"] = {
        "all_state_outputs": B / (t_up + t_pass + full / (d2h * 1e9)),
        "projection_outputs_only": B / (t_up + t_pass + proj / (d2h * 1e9)),
        "note": "upload + one resident pass (the headline's ms_per_step) + download, serial, one GPU; the projection line "
                "prices the download only (its pass is the configs[1] pass with the fused epilogue)",
        "bytes": {"all_state_outputs": full, "projection_outputs_only": proj}}
    return out


def secondary_factor_analysis(dev, R=4096, T=1000, N=8, K=2, reps=3, scipy_subset=0):
    """SURVEY section 8 row f4 in the driver's record: ``FactorAnalysisBatch.solve`` on R synthetic block-structure models
    (observations generated on the device, 10 % missing): correlations -> eigenvalues / MAP test -> minres -> loadings ->
    varimax, best of ``reps`` calls after one warm-up call, with the host's share split out (the batched ``numpy.linalg.eig``
    that supplies LAPACK's pair ORDER, factoranalysis.py:396-398; the lock-step scipy L-BFGS-B of the models that may leave
    their start vector, :173-217).  ``scipy_subset`` > 0: additionally the first that many models with ``always_scipy=True``
    (the reference's minimisation for EVERY model, as the one-model mirror class runs it)."""
    import torch

    from metran_amd import factoranalysis as fa_mod
    from metran_amd.factoranalysis import FactorAnalysisBatch

    g = torch.Generator(device=dev).manual_seed(100 + N)
    load = torch.zeros(R, N, K, dtype=torch.float64, device=dev)
    cols = (torch.arange(N, device=dev) * K) // N
    load[:, torch.arange(N, device=dev), cols] = 0.6 + 0.3 * torch.rand(R, N, dtype=torch.float64, device=dev, generator=g)
    f = torch.randn(R, T, K, dtype=torch.float64, device=dev, generator=g)
    y = torch.einsum("rtk,rnk->rtn", f, load)
    y += torch.randn(R, T, N, dtype=torch.float64, device=dev, generator=g) * torch.sqrt(1.0 - (load ** 2).sum(2))[:, None, :]
    y[torch.rand(R, T, N, device=dev, generator=g) < 0.1] = float("nan")
    del f
    fb = FactorAnalysisBatch()
    host = {"eig_order_s": 0.0, "eig_order_calls": 0, "lockstep_s": 0.0, "lockstep_models": 0}
    orig_eig, orig_lock = fa_mod.eig_order, FactorAnalysisBatch._lockstep_minres

    def timed_eig(*a, **k):
        t0 = time.perf_counter()
        r = orig_eig(*a, **k)
        host["eig_order_s"] += time.perf_counter() - t0
        host["eig_order_calls"] += 1
        return r

    def timed_lock(self, *a, **k):
        t0 = time.perf_counter()
        e0 = host["eig_order_s"]
        r = orig_lock(self, *a, **k)
        host["lockstep_s"] += (time.perf_counter() - t0) - (host["eig_order_s"] - e0)   # eig calls inside are booked as eig
        host["lockstep_models"] += int(len(r))
        return r

    fa_mod.eig_order = timed_eig
    FactorAnalysisBatch._lockstep_minres = timed_lock
    try:
        fb.solve(obs=y)
        torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            for k in host:
                host[k] = 0 if k.endswith(("calls", "models")) else 0.0
            t0 = time.perf_counter()
            r = fb.solve(obs=y)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, dict(host), r)
        dt, h, r = best
        hist = torch.bincount(r.nfactors.to(torch.int64).clamp(min=0)).tolist()
        out = {"workload": "FactorAnalysisBatch.solve: %d x (%d series, %d true factors), T=%d, 10 %% missing, fp64" % (R, N, K, T),
               "seconds": dt, "models_per_s": R / dt, "models_moved_by_lbfgsb": int((~r.stalled).sum().item()),
               "nfactors_histogram": {str(i): c for i, c in enumerate(hist) if c},
               "split_s": {"host_eig_order": h["eig_order_s"], "host_eig_order_calls": h["eig_order_calls"],
                           "lockstep_scipy": h["lockstep_s"], "lockstep_models": h["lockstep_models"],
                           "device_kernels_and_transfers": dt - h["eig_order_s"] - h["lockstep_s"]}}
        if scipy_subset:
            m = min(scipy_subset, R)
            for k in host:
                host[k] = 0 if k.endswith(("calls", "models")) else 0.0
            t0 = time.perf_counter()
            r2 = fb.solve(obs=y[:m], always_scipy=True)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t0
            out["always_scipy_subset"] = {"models": m, "seconds": d2, "models_per_s": m / d2,
                                          "host_eig_order_s": host["eig_order_s"], "host_eig_order_calls": host["eig_order_calls"],
                                          "lockstep_scipy_s": host["lockstep_s"],
                                          "models_moved_by_lbfgsb": int((~r2.stalled).sum().item()),
                                          "max_abs_loading_difference_vs_stall_checked_path":
                                              float((r2.factors - r.factors[:m, :, :r2.factors.shape[2]]).abs().max())
                                              if r2.factors.shape[2] <= r.factors.shape[2] else None}
        return out
    finally:
        fa_mod.eig_order = orig_eig
        FactorAnalysisBatch._lockstep_minres = orig_lock


def secondary_calibration(local_rank, dev, B=8192, N=8, K=2, T=1000, maxiter=200, missing=0.0, cpu=True, cpu_budget_s=6.0):
    """Row f1 in the driver's record: ``calibrate_batch`` (lock-step L-BFGS-B on the adjoint gradient) of B independent
    models from the default start to convergence."""
    import torch

    from metran_amd.calibrate import calibrate_batch
    from metran_amd.engine import BatchedKalman
    from metran_amd.synthetic import make_dfm_batch_torch

    d = make_dfm_batch_torch(B, N, K, T, seed=5000, device=dev, missing=missing)
    kf = BatchedKalman(local_rank, layout="time_major")
    try:
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        torch.cuda.synchronize()
        t_warm = time.perf_counter()    # what the FIRST calibration of a process pays on top: reported, not timed in
        calibrate_batch(kf, maxiter=2)  # warm-up: kernel load, allocator
        # ... and the compaction path, which two iterations do not reach: the first gather / nonzero of a process loads those
        # torch kernels (~100 ms, once per process -- not a cost of the calibration)
        warm = kf.subset(torch.arange(B, device=dev)[torch.ones(B, dtype=torch.bool, device=dev)].nonzero().squeeze(1)[: max(2, B // 2)])
        calibrate_batch(warm, maxiter=1)
        warm.close()
        warm = kf.subset(torch.arange(min(B, 8), device=dev))     # ... and the differenced tail's (several step lengths per launch)
        calibrate_batch(warm, maxiter=2, fd_below=10 ** 9)
        warm.close()
        torch.cuda.synchronize()
        t_warm = time.perf_counter() - t_warm
        t0 = time.perf_counter()
        fd_below = 4096 if N + K <= 16 else 2048   # (differenced gradients for the last stragglers: one round of wavefronts)
        res = calibrate_batch(kf, maxiter=maxiter, fd_below=fd_below)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        true_obj = kf.loglik(d["phi"], d["q"])
        out = {"workload": "calibrate_batch: %d x (%d series, %d factors), T=%d, %d %% missing, fp64, adjoint gradient (forward differences once "
                           "%d x active models <= %d), the L-BFGS step on the device (mk_lbfgs.hip)" % (B, N, K, T, round(100 * missing), N + K + 1, fd_below),
               "seconds": dt, "models_per_s": B / dt, "iterations": int(res.nit), "objective_evaluations": int(res.nfev),
               "launches": int(res.launches), "passes": int(res.passes),
               "first_use_warmup_s": t_warm,   # kernel load, the adjoint workspace, torch's gather / nonzero kernels: once per process
               "models_at_the_iteration_limit": int((res.nit_model >= maxiter).sum().item()),
               "converged_frac": float(res.converged.double().mean()),
               "frac_at_or_below_true_parameter_objective": float((res.obj <= true_obj + 1e-6).double().mean())}
        # real fraction of the fp64 pipe over the WHOLE calibration (wall time, host work included): every objective evaluation
        # priced at the filter's executed flops; the backward passes of the adjoint gradient are not counted (a floor)
        ex = executed_flops(N, K, T, "solver", missing, False)["filter"]
        out["real_frac"] = int(res.nfev) * ex / dt / 1e12 / FP64_PEAK_TFLOPS
        out["real_frac_note"] = "objective evaluations x executed filter flops / wall seconds / 78.6 TFLOP/s (adjoint backward passes not counted)"
        if cpu:
            try:
                k = min(B, 16)
                out["cpu_baseline"] = cpu_calibration_leg(d["obs"][:k].cpu().numpy(), d["loadings"][:k].cpu().numpy(), res.alpha[:k].cpu().numpy(),
                                                          res.obj[:k].cpu().numpy(), budget_s=cpu_budget_s)
                out["speedup_vs_cpu_baseline"] = out["models_per_s"] / out["cpu_baseline"]["models_per_s"]
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        return out
    finally:
        kf.close()
        torch.cuda.empty_cache()


def cpu_calibration_leg(obs, loadings, gpu_alpha, gpu_obj, budget_s=8.0, min_models=2, max_models=16, pmin=1e-5):
    """The CPU beside ``calibrate_batch`` (row f1): what ``Metran.solve`` does for ONE model -- scipy L-BFGS-B from alpha = 10,
    bounds alpha >= pmin, 2-point forward differences with scipy's step 1e-8 (metran/solver.py:248-255) -- with the objective
    from the optimised C leg (oracle/kalman_fast.c) and the P + 1 points of every gradient evaluated as ONE OpenMP batch over
    this box's usable cores; models one after the other until ``budget_s`` is spent (at least ``min_models``).  Also the
    parity of the GPU calibration: its objective and parameters against scipy's on the same models."""
    import numpy as np
    from scipy.optimize import minimize

    import oracle
    from metran_amd.params import phi_q_from_alpha

    native = False
    try:
        oracle.build(native=True)
        oracle.load_fast(native=True)
        native = True
    except Exception:
        pass
    N, K = loadings.shape[1], loadings.shape[2]
    n = N + K
    cores = cpu_leg_threads(oracle.fast_num_threads(native))
    oracle.fast_set_num_threads(cores, native)
    eps = 1e-8
    done, nfev, t_tot, gpu_vs_cpu_obj = 0, 0, 0.0, 0.0
    obj_above, obj_below, alpha_rel = 0.0, 0.0, 0.0
    for b in range(min(max_models, obs.shape[0])):
        ob = np.repeat(obs[b][None], n + 1, 0)
        ld = np.repeat(loadings[b][None], n + 1, 0)

        def fun(x):
            nonlocal nfev
            pts = np.repeat(x[None], n + 1, 0)
            pts[1:] += eps * np.eye(n)
            phi, q = phi_q_from_alpha(pts, ld)
            f = oracle.fast_dfm_batch(ob, phi, q, ld, outputs="mle", native=native)["mle"]
            nfev += n + 1
            return float(f[0]), (f[1:] - f[0]) / eps

        t0 = time.perf_counter()
        r = minimize(fun, np.full(n, 10.0), jac=True, method="l-bfgs-b", bounds=[(pmin, None)] * n)
        t_tot += time.perf_counter() - t0
        done += 1
        # the GPU's optimum re-evaluated by the CPU objective, against scipy's own: above = the GPU calibration ended higher
        # (worse) than scipy, below = lower (scipy's differenced line search stalled first)
        ph, qq = phi_q_from_alpha(np.asarray(gpu_alpha[b], dtype=np.float64)[None], loadings[b][None])
        f_gpu_on_cpu = float(oracle.fast_dfm_batch(obs[b][None], ph, qq, loadings[b][None], outputs="mle", native=native)["mle"][0])
        gpu_vs_cpu_obj = max(gpu_vs_cpu_obj, abs(float(gpu_obj[b]) - f_gpu_on_cpu) / abs(f_gpu_on_cpu))
        obj_above = max(obj_above, (f_gpu_on_cpu - r.fun) / abs(r.fun))
        obj_below = max(obj_below, (r.fun - f_gpu_on_cpu) / abs(r.fun))
        alpha_rel = max(alpha_rel, float(np.max(np.abs(gpu_alpha[b] - r.x) / np.maximum(np.abs(r.x), 1.0))))
        if t_tot > budget_s and done >= min_models:
            break
    return {"models_per_s": done / t_tot, "cores": cores, "kind": "port", "models": done, "seconds": t_tot, "objective_evaluations": nfev,
            "sample": "scipy L-BFGS-B per model (the reference's optimiser and differencing, solver.py:248-255) on oracle/kalman_fast.c, "
                      "the P+1 points of a gradient as one OpenMP batch, models one after the other",
            "parity_of_the_gpu_calibration": {"models": done, "objective_max_rel_excess_over_scipy": obj_above,
                                              "objective_max_rel_gain_over_scipy": obj_below,
                                              "gpu_objective_vs_cpu_objective_at_the_gpu_optimum_max_rel": gpu_vs_cpu_obj,
                                              "alpha_max_rel_diff_vs_scipy": alpha_rel}}


def wide_dropin_dataset(N=32, K=4, T=2000, missing=0.3, seed=7100):
    """A seeded 32-series / 4-factor dataset in Metran's input format (daily pandas Series, 30 % of the values missing at
    random): block loadings 0.6-0.9 (8 series per common factor), alpha ~ U(5, 40) days, simulated from the model itself."""
    import numpy as np
    import pandas as pd

    rng = np.random.default_rng(seed)
    load = np.zeros((N, K))
    load[np.arange(N), (np.arange(N) * K) // N] = 0.6 + 0.3 * rng.random(N)
    alpha = rng.uniform(5.0, 40.0, N + K)
    phi = np.exp(-1.0 / alpha)
    q = np.concatenate([(1.0 - phi[:N] ** 2) * (1.0 - (load ** 2).sum(1)), 1.0 - phi[N:] ** 2])
    x = np.zeros(N + K)
    Y = np.empty((T, N))
    for t in range(T):
        x = phi * x + np.sqrt(q) * rng.standard_normal(N + K)
        Y[t] = x[:N] + load @ x[N:]
    Y[rng.random((T, N)) < missing] = np.nan
    idx = pd.date_range("2000-01-01", periods=T, freq="D")
    return [pd.Series(Y[:, j], index=idx, name="s%02d" % j).dropna() for j in range(N)], load


def secondary_dropin_wide(reference_evals=2):
    """What a Metran user with a few dozen series has (VERDICT r5 missing 5 / next 3c): ONE wide model calibrated through the
    UNMODIFIED reference class (oracle/_ref) -- 32 series, 4 common factors, T = 2000, 30 % missing.  The reference's own
    factor analysis settles on 1-2 factors for any such dataset (its MAP test, factoranalysis.py:219-267), so the four-factor
    loadings are handed to the instance the way a user who knows them would (``mt.get_factors`` bound on the INSTANCE; the
    class is untouched).  Timed: ``solve()`` with scipy's own differencing on the HIP engine (every ``get_mle`` one B = 1
    launch), ``solve(solver=HipSolve)`` (objective + P differences per launch), ``solve(solver=HipSolveAdjoint)``; the
    reference engine on the host is timed on ``reference_evals`` objective evaluations and its ``solve()`` EXTRAPOLATED by
    scipy's evaluation count (0.7 s per evaluation x ~1500); objective parity = the reference engine's value AT the
    optimum each HIP route found."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import _refshim

    if not _refshim.reference_available():
        return {"error": "oracle/_ref not staged (oracle/make_ref.sh where the reference is mounted)"}
    metran = _refshim.install()
    import metran_amd.kalmanfilter as hip
    from metran_amd.solver import HipSolve, HipSolveAdjoint

    series, load = wide_dropin_dataset()

    def model():
        mt = metran.Metran(series, name="wide")
        mt.factors, mt.nfactors = load.copy(), load.shape[1]
        mt.get_factors = lambda oseries=None: mt.factors      # the user's loadings; solve() asks the instance (metran.py:1022)
        return mt

    def run(solver=None, **kw):
        mt = model()
        t0 = time.perf_counter()
        mt.solve(report=False, **({"solver": solver} if solver else {}), **kw)
        return mt, {"solve_s": time.perf_counter() - t0, "nfev": int(mt.fit.nfev), "obj": float(mt.fit.obj_func)}

    out = {"workload": "one synthetic 32-series / 4-factor model, T=2000 daily, 30 % missing, through the reference's unmodified "
                       "Metran class (loadings handed to the instance): solve() on the HIP engine vs the reference engine on the host"}
    hip.install(metran)
    try:
        run(HipSolve, options={"maxiter": 2})          # warm-up outside the timings: context, kernel load, upload
        mts = {}
        for key, solver in (("hip_engine_scipy_solver", None), ("hip_solver_fd", HipSolve), ("hip_solver_adjoint", HipSolveAdjoint)):
            mts[key], out[key] = run(solver)
        popt = {k: m.parameters.optimal.values.copy() for k, m in mts.items()}
    finally:
        hip.uninstall(metran)
    # the reference engine on the host (numpy engine; numba is not importable): a few objective evaluations, timed
    mt = model()
    mt.get_factors(mt.oseries)
    mt._init_kalmanfilter(mt.oseries, engine="numpy")
    mt.set_init_parameters()
    t0 = time.perf_counter()
    for _ in range(reference_evals):
        mt.get_mle(mt.parameters.initial.values)
    per_eval = (time.perf_counter() - t0) / reference_evals
    nf = out["hip_engine_scipy_solver"]["nfev"]
    out["reference_engine_on_host"] = {"get_mle_s": per_eval, "evaluations_timed": reference_evals, "solve_s_extrapolated": per_eval * nf,
                                       "extrapolated_by": "scipy's nfev on the HIP engine (%d)" % nf,
                                       "engine": "seqkalmanfilter_np, one core (numba not importable)"}
    for k in mts:
        ref_obj = float(mt.get_mle(popt[k]))            # the reference engine's objective at the optimum this route found
        out[k]["obj_rel_err_vs_reference_engine_at_the_same_optimum"] = abs(out[k]["obj"] - ref_obj) / abs(ref_obj)
        out[k]["solve_speedup_vs_reference_engine_extrapolated"] = out["reference_engine_on_host"]["solve_s_extrapolated"] / out[k]["solve_s"]
    return out


def secondary_dropin():
    """BASELINE configs[0] / north_star's "drop-in ... with the CPU path timed in the same run": the UNMODIFIED reference class
    (oracle/_ref, the staged verbatim copy, through the pastas stub) on examples/data (5 series, 1 factor, T = 6255):
    ``Metran.solve()`` + ``get_simulation()`` (/root/reference/metran/metran.py:991-1042, 831-883) once with the reference's own
    numpy engine on the host and once with ``metran_amd.kalmanfilter.install`` (every ``get_mle`` = one B = 1 launch of the HIP
    engine), then plug point A: ``solve(solver=HipSolve)`` (objective + forward differences in one launch) and
    ``HipSolveAdjoint``.  Wall seconds, nfev and the objective of each."""
    import glob

    import pandas as pd

    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import _refshim

    if not _refshim.reference_available():
        return {"error": "oracle/_ref not staged (oracle/make_ref.sh where the reference is mounted)"}
    metran = _refshim.install()
    import metran_amd.kalmanfilter as hip
    from metran_amd.solver import HipSolve, HipSolveAdjoint

    files = sorted(glob.glob(os.path.join(_refshim.REFERENCE_ROOT, "examples", "data", "*_res.csv")))
    series = []
    for f in files:
        x = pd.read_csv(f, header=0, index_col=0, parse_dates=True).squeeze()
        x.name = os.path.basename(f).split("_")[0]
        series.append(x)

    def run(solver=None):
        mt = metran.Metran(series, name="B21B0214")
        t0 = time.perf_counter()
        if solver is None:
            mt.solve(report=False)
        else:
            mt.solve(solver=solver, report=False)
        t1 = time.perf_counter()
        sim = mt.get_simulation(series[-1].name)
        t2 = time.perf_counter()
        return {"solve_s": t1 - t0, "get_simulation_s": t2 - t1, "nfev": int(mt.fit.nfev), "obj": float(mt.fit.obj_func),
                "simulation_rows": int(sim.shape[0])}

    out = {"workload": "BASELINE.json configs[0]: examples/data (5 series, 1 factor, T=6255), the reference's Metran class, "
                       "solve() + get_simulation(); reference objective at the optimum 2332.3270694, nfev 77"}
    out["reference_engine_on_host"] = dict(run(), engine="seqkalmanfilter_np + Python kalmansmoother (numba %s), one core"
                                           % ("importable" if "numba" in sys.modules else "not importable"))
    hip.install(metran)
    try:
        run(HipSolveAdjoint)   # warm-up outside the timings: context creation, kernel load, first upload
        out["hip_engine_scipy_solver"] = dict(run(), engine="metran_amd.kalmanfilter.install(metran): seqkalmanfilter_hip / "
                                              "kalmansmoother_hip, scipy L-BFGS-B unchanged (77 x get_mle = 77 launches)")
        out["hip_solver_fd"] = dict(run(HipSolve), engine="Metran.solve(solver=HipSolve): objective + P forward differences per launch")
        out["hip_solver_adjoint"] = dict(run(HipSolveAdjoint), engine="Metran.solve(solver=HipSolveAdjoint): adjoint gradient")
    finally:
        hip.uninstall(metran)
    ref = out["reference_engine_on_host"]
    for k in ("hip_engine_scipy_solver", "hip_solver_fd", "hip_solver_adjoint"):
        out[k]["obj_minus_reference"] = out[k]["obj"] - ref["obj"]
        out[k]["solve_speedup_vs_reference_engine"] = ref["solve_s"] / out[k]["solve_s"]
    return out


# --------------------------------------------------------------------------------------- the stdout line
def _sig(x, digits=5):
    """Floats to ``digits`` significant digits (the line is read by people and kept in an 8 KB tail by the driver)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):   # (objectives and the headline value keep their digits: they are compared, not read)
        return {k: _sig(v, 12 if k in ("obj", "summed_mle", "value", "models_per_s", "ms_per_step") else digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


NOTES = {
    "frac": "roofline.frac: SURVEY 8d accounting -- algorithmic bytes (hbm) or the REFERENCE algorithm's flops (fp64: dense updates, RTS with an "
            "explicit inverse) per launch / hipEvent launch time / peak; work-equivalent throughput where the path executes fewer flops",
    "real_frac": "real_frac: max(algorithmic bytes / time / 8 TB/s, flops the kernel's own formulation EXECUTES / time / 78.6 TFLOP/s) -- the "
                 "utilisation of the nearer roof; 8d_frac: kernel-time rate x bytes a full-square six-output pass WOULD move / 8 TB/s (hypothetical)",
    "cpu": "cpu: oracle/kalman_fast.c (optimised C, OpenMP over models, same outputs) on `cores` of this box, n models, second pass; chk: "
           "oracle/kalman_oracle.c, the bit-faithful checker; f1: scipy L-BFGS-B + forward differences per model on kalman_fast.c. Context, not credit",
    "k": "secondary.*.roofline.k: per kernel [ms per launch (hipEvents), algorithmic GB/s, executed TFLOP/s, real_frac, PMC traffic GB or null]; "
         "ms: [filter, smoother]; bar_models_per_s: 40 % of 8 TB/s in SURVEY 8d's full-output accounting",
    "parity": "parity: against oracle/kalman_oracle.c on a sample spread over the batch: mle = max relative error of -2 log L (n_mle models); "
              "other keys = max abs error of that output (n models)",
    "peaks": "peaks: HBM 8000 GB/s (MI355X_MICROARCH.md; 6290 measured copy ceiling); fp64 78.6 TFLOP/s datasheet vector = matrix (not in the guide; "
             "measured ceilings here 72 MFMA / 66 FMA TFLOP/s)",
}


def _compact_roofline(r, with_kernels=True):
    out = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "real_frac", "real_frac_roof", "traffic",
                                 "algorithmic_bytes", "avg_launch_ms") if k in r}
    if "survey_8d_full_output_accounting" in r:
        out["8d_frac"] = r["survey_8d_full_output_accounting"]["frac_of_peak"]
    if with_kernels:
        out["kernels"] = {k: {"ms": v["ms"], "GBps": v["GBps"], "exec_TFLOPps": v.get("executed_TFLOPps"), "real_frac": v.get("real_frac"),
                              "traffic_GB": v.get("traffic_GB"), "alg_GB": v["algorithmic_GB"]} for k, v in r.get("kernels", {}).items()}
    return out


def _compact_cpu(c, brief=False):
    if not isinstance(c, dict) or "error" in c or "models_per_s" not in c:
        return c
    if brief:
        return {"models_per_s": c["models_per_s"], "cores": c.get("cores"), "n": c.get("models")}
    return {k: c[k] for k in ("models_per_s", "cores", "models", "kind", "variant") if k in c}


def _compact_parity(par):
    if not isinstance(par, dict) or "error" in par:
        return par
    ren = {"loglik_max_rel_err": "mle", "loglik_models_compared": "n_mle", "projection_models_compared": "n", "state_models_compared": "n",
           "sim_means_max_abs_err": "sim_means", "sim_vars_max_abs_err": "sim_vars", "state_means_max_abs_err": "S", "state_vars_max_abs_err": "var",
           "predicted_means_max_abs_err": "Xp", "predicted_covariances_max_abs_err": "Pp", "filtered_means_max_abs_err": "F",
           "filtered_covariances_max_abs_err": "Pf", "smoothed_means_max_abs_err": "S", "smoothed_covariances_max_abs_err": "Ps"}
    return {ren[k]: v for k, v in par.items() if k in ren}


def _compact_secondary(name, v):
    if "error" in v and len(v) <= 2:
        return v
    out = {}
    for k in ("models_per_s", "objective_evaluations_per_s", "seconds", "iterations", "launches", "converged_frac", "real_frac", "first_use_warmup_s",
              "models_at_the_iteration_limit",
              "frac_at_or_below_true_parameter_objective", "state_tape", "models_moved_by_lbfgsb", "nfactors_histogram", "bench_wall_s"):
        if k in v and v[k] is not None:
            out[k] = v[k]
    if v.get("filter_ms") is not None:
        out["ms"] = [v["filter_ms"], v.get("smoother_ms")]
    if "parity" in v:
        out["parity"] = _compact_parity(v["parity"])
    if "roofline" in v:
        rr = v["roofline"]
        out["roofline"] = {"bound": rr["bound"], "frac": rr["frac"], "real_frac": rr.get("real_frac"), "roof": rr.get("real_frac_roof")}
        if "survey_8d_full_output_accounting" in rr:
            out["roofline"]["8d_frac"] = rr["survey_8d_full_output_accounting"]["frac_of_peak"]
        ks = v["roofline"].get("kernels", {})
        out["roofline"]["k"] = {k.replace("_kernel", ""): [x["ms"], x["GBps"], x.get("executed_TFLOPps"), x.get("real_frac"), x.get("traffic_GB")]
                                for k, x in ks.items()}
        bar = v["roofline"].get("survey_8d_full_output_accounting", {}).get("north_star_bar")
        if bar:
            out["bar_models_per_s"] = bar["models_per_s"]
    if "cpu_baseline" in v:
        out["cpu"] = _compact_cpu(v["cpu_baseline"], brief=True)
        par = v["cpu_baseline"].get("parity_of_the_gpu_calibration") if isinstance(v["cpu_baseline"], dict) else None
        if par:
            out["parity"] = {"n": par["models"], "obj_above_scipy": par["objective_max_rel_excess_over_scipy"],
                             "obj_below_scipy": par["objective_max_rel_gain_over_scipy"]}
    if "cpu_baseline_checker" in v:
        out["chk"] = _compact_cpu(v["cpu_baseline_checker"], brief=True)
    if "split_s" in v:
        out["split_s"] = {k: v["split_s"][k] for k in ("host_eig_order", "lockstep_scipy", "device_kernels_and_transfers")}
    if "always_scipy_subset" in v:
        a = v["always_scipy_subset"]
        out["always_scipy"] = {k: a[k] for k in ("models", "models_per_s", "models_moved_by_lbfgsb", "lockstep_scipy_s", "host_eig_order_s") if k in a}
    for k in ("reference_engine_on_host", "hip_engine_scipy_solver", "hip_solver_fd", "hip_solver_adjoint"):   # the drop-in entries
        if k in v:
            out[k.replace("hip_", "").replace("reference_engine_on_host", "ref_host")] = {
                kk: vv for kk, vv in v[k].items() if kk in ("solve_s", "get_simulation_s", "nfev", "obj", "get_mle_s", "solve_s_extrapolated",
                                                           "obj_rel_err_vs_reference_engine_at_the_same_optimum")}
    return out


def compact_line(res, full_path=None):
    """The ONE stdout line: the contract's keys, the headline's roofline with its kernels, the CPU legs, and every secondary
    as numbers (prose lives once in ``notes``; the verbose record goes to ``--full-record``).  Ordered so that the LAST 8 KB
    -- what the driver's record keeps of stdout -- hold every secondary entry."""
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data", "config") if k in res}
    if "roofline" in res:
        out["roofline"] = _compact_roofline(res["roofline"])
        out["roofline"]["path_achieved_GBps"] = res["roofline"].get("path_achieved_GBps")
        out["roofline"]["kernel_source_sha256"] = res["roofline"].get("kernel_source_sha256")
    if "cpu_baseline" in res:
        c = res["cpu_baseline"]
        out["cpu_baseline"] = {k: c[k] for k in ("value", "unit", "models_per_s", "cores", "kind", "variant", "sample", "host") if k in c}
        if isinstance(out["cpu_baseline"].get("sample"), str):
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:160]
    for k in ("speedup_vs_cpu_baseline", "speedup_vs_cpu_baseline_checker", "loglik_max_rel_err", "models_per_s", "models_per_s_per_gpu",
              "summed_mle", "rccl_ranks", "dry_run", "allreduce_check"):
        if k in res:
            out[k] = res[k]
    if "cpu_baseline_checker" in res:
        out["cpu_baseline_checker"] = _compact_cpu(res["cpu_baseline_checker"])
    ref = res.get("cpu_baseline_reference_as_shipped")
    if isinstance(ref, dict):
        r = ref.get("reference")
        out["cpu_baseline_reference_as_shipped"] = ({k: r[k] for k in ("models_per_s", "cores", "kind", "loglik_max_rel_err_vs_gpu")}
                                                    if isinstance(r, dict) else r)
        out["numba"] = ref.get("numba", "")[:40]
    if "collective" in res:
        c = res["collective"]
        out["collective"] = {"backend": c["backend"], "allreduces_in_timed_region": c["allreduces_in_timed_region"],
                             "ms_per_allreduce": c["ms_per_allreduce"], "per_rank": c["per_rank"]}
    if "pcie" in res:
        out["pcie"] = res["pcie"]
    out["notes"] = dict(NOTES, full_record=(os.path.relpath(full_path, ROOT) if full_path else "not written (--full-record)"))
    if "secondary" in res:
        out["secondary"] = {k: _compact_secondary(k, v) for k, v in res["secondary"].items()}
    return _sig(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: c2 (BASELINE configs[1], 4096 models per GPU) -- except with --gpus 8, where it is c3 = configs[2]'s "
                         "per-GPU share (65536 models over 8 GPUs = 8192 per GPU), so that the driver's 8-GPU run lands on the "
                         "BASELINE configuration without a flag")
    ap.add_argument("--batch", type=int, default=None, help="models per GPU (default: the configuration's)")
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--packed-sym", action="store_true",
                    help="c2/c3: packed-symmetric records (MK_PACKED_SYM; n + n(n+1)/2 doubles per moment set)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 PMC child runs after the timed region (one GPU only)")
    ap.add_argument("--transfers", action="store_true",
                    help="one GPU: also time the host-link transfers a host-to-host caller pays (reported under 'pcie', never in value)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs[3] / configs[4] lines that the default one-GPU run appends after the headline")
    ap.add_argument("--only", default=None,
                    help="comma-separated names of the secondary entries to run (default: all); e.g. --only c4,c4_full_sym")
    ap.add_argument("--full-record", default=None,
                    help="file that receives the VERBOSE record (every workload description, note and per-kernel figure); the "
                         "stdout line is its compact form.  Default: gpurun_out/bench_full.json when gpurun_out/ exists")
    ap.add_argument("--layout", default="time_major", choices=["time_major", "model_major"])
    ap.add_argument("--projection-path", default="auto", choices=["auto", "tape", "records"],
                    help="c4: 'records' = filtered records + RTS smoother (round 3) instead of the tape path")
    ap.add_argument("--tape-filter", default=None, choices=["observable", "state"],
                    help="c4 / c4s: the writer of the backward tape (mk_set_kernel_variant MK_VARIANT_TAPE_FILTER; default: the library's)")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check without a GPU (gloo): launch, barrier, max-over-ranks, one JSON line with value null")
    args = ap.parse_args()
    config_defaulted = args.config is None
    if config_defaulted:
        args.config = "c3" if args.gpus == 8 else "c2"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        respawn_under_torchrun(args.gpus)  # does not return
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np  # noqa: F401
    import torch

    B, N, K, T, missing, mode = CONFIGS[args.config]
    B = args.batch or B
    T = args.T or T
    steps = args.steps if args.steps is not None else (50 if mode == "full" else 3)
    warmup = args.warmup if args.warmup is not None else (10 if mode == "full" else 1)

    dist = None
    backend = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # launched by torch.distributed.run: one rank per GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        backend = dist.get_backend()
    dev = torch.device("cpu") if args.dry_run else torch.device("cuda", local_rank)
    if not args.dry_run:
        torch.cuda.set_device(local_rank)

    def sync():
        if not args.dry_run:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    w = None
    tape = False
    collectives = 0
    if args.dry_run:
        total = torch.zeros(1, dtype=torch.float64, device=dev)
        for _ in range(warmup):
            total.fill_(float(rank + 1))
            if dist is not None:
                dist.all_reduce(total)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            total.fill_(float(rank + 1))
            if dist is not None:
                dist.all_reduce(total)
                collectives += 1
        sync()
        elapsed = time.perf_counter() - t0
        f_avg = s_avg = 0.0
    else:
        w = Workload(args.config, local_rank, rank, dev, args.layout, args.packed_sym, batch=args.batch, T=args.T,
                     projection_path=args.projection_path, variants=({"tape_filter": args.tape_filter} if args.tape_filter else None))
        total = w.total
        tape = getattr(w, "tape", False)

        def after_step(wl):
            nonlocal collectives
            if dist is not None and wl.mode != "solver":
                dist.all_reduce(wl.total)          # RCCL all-reduce of the summed -2 log L (8 bytes)
                collectives += 1

        elapsed, f_avg, s_avg = timed_run(w, steps, warmup, sync, after_step)
        collectives -= warmup if (dist is not None and mode != "solver") else 0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    nranks = torch.ones(1, dtype=torch.float64, device=dev)
    per_rank_ms = ar_ms = None
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(nranks)  # ranks that actually took part in the collective
        # so that a 1 -> N run explains itself: every rank's own kernel times, and -- measured AFTER the
        # timed region, never inside it -- the wall time of the 8-byte all-reduce alone
        km = torch.tensor([f_avg, s_avg, elapsed], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(km) for _ in range(world)]
        dist.all_gather(gathered, km)
        per_rank_ms = [{"rank": i, "filter_ms": float(g[0]), "smoother_ms": float(g[1]), "elapsed_s": float(g[2])}
                       for i, g in enumerate(gathered)]
        probe = torch.zeros(1, dtype=torch.float64, device=dev)
        for _ in range(5):
            dist.all_reduce(probe)
        sync()
        t0 = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(probe)
        if not args.dry_run:
            torch.cuda.synchronize()
        ar_ms = 1e3 * (time.perf_counter() - t0) / 50
    elapsed = float(el.item())

    if rank == 0:
        units = EVALS_PER_STEP if mode == "solver" else 1     # model passes per step
        ms_per_step = 1e3 * elapsed / steps
        models_per_s = world * B * units * steps / elapsed
        workload = w.describe() if w is not None else "dry run of %s" % BASELINE_NAME[args.config]
        if config_defaulted and args.gpus == 8:
            workload += (" [default of --gpus 8: configs[2] = 65536 models over 8 GPUs; the 1/2/4-GPU lines of the same sweep "
                         "default to configs[1]'s 4096 models per GPU -- per-GPU kernel rates of the two are equal within 2 %, "
                         "profiles/r04/bench_c3.json]")
        res = {
            "metric": "Kalman filter+smoother steps/sec (batched DFMs)",
            "value": None if args.dry_run else models_per_s * T,
            "unit": "model-timesteps/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "batch_per_gpu": B, "total_batch": B * world, "series": N,
                       "factors": K, "T": T, "parallelism": "dp%d" % world, "layout": args.layout,
                       "packed_sym": bool(args.packed_sym)},
            "rccl_ranks": int(nranks.item()),
            # how rccl_ranks was obtained: the process-group backend whose all_reduce summed one 1.0 per rank
            # ("nccl" = RCCL on ROCm), or None when no process group exists (plain `python bench.py`, one GPU)
            "collective": {"backend": backend, "allreduces_in_timed_region": collectives,
                           "what": "summed -2 log L, 8 bytes, once per step" if backend else "none (no process group)",
                           "ms_per_allreduce": ar_ms, "ms_per_allreduce_note": (
                               "50 back-to-back 8-byte all-reduces after the timed region, rank 0's wall clock" if backend else None),
                           "per_rank": per_rank_ms},
        }
        if args.dry_run:
            res["dry_run"] = "plumbing only (gloo, no kernels): not a measurement"
            res["allreduce_check"] = float(total.item())
        else:
            res["models_per_s"] = models_per_s
            res["models_per_s_per_gpu"] = models_per_s / world
            res["summed_mle"] = float(total.item())
            if world == 1 and not args.no_cpu_baseline:
                host = w.host_inputs()
                gpu_mle = w.bufs["mle"].cpu().numpy()
                # `cpu_baseline` is the FAIR leg: oracle/kalman_fast.c, what a CPU can do with the same
                # recursions; the fidelity checker (the bit-faithful C port, slower per core than it need be) and the reference as
                # shipped ride under their own keys.  None of them is credit.
                chk, rel = cpu_baseline_port(host, mode, gpu_mle)
                res["loglik_max_rel_err"] = rel
                try:
                    res["cpu_baseline"] = cpu_baseline_optimised(host, mode, gpu_mle)
                    res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
                except Exception as e:  # noqa: BLE001 -- the line must survive: fall back to the checker as the baseline
                    res["cpu_baseline"] = dict(chk, note="optimised leg failed (%s: %s): this is the fidelity checker" % (type(e).__name__, e))
                res["cpu_baseline_checker"] = chk
                res["speedup_vs_cpu_baseline_checker"] = res["value"] / chk["value"]
                res["cpu_baseline_reference_as_shipped"] = cpu_baseline_reference(host, mode, gpu_mle)
    if w is not None:
        w.close()
        w = None
        torch.cuda.empty_cache()
    if rank == 0 and not args.dry_run:
        live = None
        if world == 1 and dist is None and not args.no_live_traffic and args.batch is None and args.T is None:
            live = live_traffic(args.config, args.packed_sym, projection_path=args.projection_path)  # after the timed region
        res["roofline"] = build_roofline(args.config, N, K, T, B, mode, missing, f_avg, s_avg, args.packed_sym, live=live, tape=tape,
                                         tape_filter=args.tape_filter)
    # ---- the other BASELINE configurations, AFTER the headline's timed region (one GPU, default headline only) ----
    if (rank == 0 and world == 1 and not args.dry_run and not args.no_secondary and args.config == "c2"
            and args.batch is None and args.T is None and not args.packed_sym):
        res["secondary"] = {}
        cpu = not args.no_cpu_baseline

        def sw(name, **kw):
            kw["cpu"] = kw.get("cpu", True) and cpu
            return lambda: secondary_workload(name, local_rank, rank, dev, args.layout, sync, **kw)

        generic = {"kernel_family": "generic"}
        entries = (("c4", sw("c4")), ("c5", sw("c5")), ("c4_state_variances", sw("c4s")),
                   # all six reference outputs of configs[3]'s batch, packed-symmetric records, 138 GB resident (VERDICT r5 missing 2)
                   ("c4_full_sym", sw("c4f", packed_sym=True, steps=3, warmup=1, live=False)),
                   # the size-generic kernel family -- what a shape without a specialised module runs
                   ("generic_c2", sw("c2", variants=generic, steps=3, warmup=1, live=False, cpu=False)),
                   ("generic_c4", sw("c4", variants=generic, batch=512, steps=2, warmup=1, live=False, cpu=False, n_proj=8)),
                   # ... and a shape ONLY that family serves: 100 states, all six outputs, with the CPU beside it
                   ("generic_96x4", sw("g100", steps=2, warmup=1, live=False, n_proj=4)),
                   ("f4_factor_analysis", lambda: secondary_factor_analysis(dev)),
                   ("f4_factor_analysis_32x4", lambda: secondary_factor_analysis(dev, N=32, K=4, reps=2, scipy_subset=256)),
                   ("f1_calibration", lambda: secondary_calibration(local_rank, dev, cpu=cpu)),
                   ("f1_calibration_32x4", lambda: secondary_calibration(local_rank, dev, B=512, N=32, K=4, T=500, missing=0.3, cpu=cpu,
                                                                         cpu_budget_s=10.0)),
                   ("c1_dropin", secondary_dropin), ("c1w_dropin", secondary_dropin_wide))
        only = set(args.only.split(",")) if args.only else None
        for name, fn in entries:
            if only is not None and name not in only:
                continue
            t_sec = time.perf_counter()
            try:
                res["secondary"][name] = fn()
            except Exception as e:  # noqa: BLE001 -- the headline line must survive a failure here
                res["secondary"][name] = {"error": "%s: %s" % (type(e).__name__, e)}
            res["secondary"][name]["bench_wall_s"] = time.perf_counter() - t_sec
    if rank == 0 and world == 1 and args.transfers and not args.dry_run:
        try:
            res["pcie"] = measure_transfers(dev, B, T, N, K, res["ms_per_step"])
        except Exception as e:  # noqa: BLE001 -- the headline line must survive a failure here
            res["pcie"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        full_path = args.full_record or (os.path.join(ROOT, "gpurun_out", "bench_full.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
        if full_path:
            try:
                with open(full_path, "w") as fh:
                    json.dump(res, fh)
            except OSError:
                full_path = None
        print(json.dumps(compact_line(res, full_path)))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
