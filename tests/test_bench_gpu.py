"""GPU: the bench line as the driver gets it (``python bench.py``), with the non-headline BASELINE configurations attached
after the headline's timed region, and throughput FLOORS so that a performance regression of a kernel turns the GPU test
tier red.  Floors are ~75 % of the slowest value measured over the round's leases, not targets -- except
where a bar exists (configs[3]: SURVEY 8d's 37.3 k models/s), which IS the floor.

Round 6: the stdout line is the COMPACT form (numbers; prose once under ``notes``; every secondary inside the last 8 KB, which
is what the driver's record keeps of stdout); the verbose record goes to ``--full-record``.  Both are checked."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAR_C4 = 0.40 * 8000e9 / (8 * 2000 * (32 + 4 * (36 + 36 * 36)))   # 40 % of 8 TB/s in full-output accounting = 37 313 models/s
# smooth_state_variances at configs[3]'s shape: the bar of the same accounting (the floor used to be the OLD
# path's 31.7 k, so the tier stayed green while the line sat under the bar)
STATE_VARIANCES_FLOOR = BAR_C4
# (end of round 6: 42.4 k measured -- the bar is passed by 13 %, so the floor IS the bar, as the round-5 verdict asked; earlier in the
# round the line sat 1-2 % under it and the floor was the bar less the box-to-box spread.  The line prints the bar: bar_models_per_s.)
C4_FLOOR = 42000.0   # configs[3], projection outputs: 46.8-47.3 k measured at the end of round 6 (the round-5 verdict asked for 45 k in the line); the floor leaves 11 % for a slow lease (the objective kernel has ranged over 10 % between leases)
SECONDARIES = ("c4", "c5", "c4_state_variances", "c4_full_sym", "generic_c2", "generic_c4", "generic_96x4", "f4_factor_analysis", "f4_factor_analysis_32x4",
               "f1_calibration", "f1_calibration_32x4", "c1_dropin", "c1w_dropin")


@pytest.fixture(scope="module")
def both(tmp_path_factory):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    full = str(tmp_path_factory.mktemp("bench") / "full.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "10", "--full-record", full],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # ONE JSON line
    return json.loads(lines[0]), json.load(open(full)), lines[0]


@pytest.fixture(scope="module")
def line(both):
    return both[0]


@pytest.fixture(scope="module")
def full(both):
    return both[1]


def test_every_secondary_is_inside_the_drivers_tail(both):
    """the driver's record keeps the last ~8 KB of stdout; the line used to be 15 KB and lost configs[3]."""
    line, _, raw = both
    tail = raw[-8000:]
    for name in SECONDARIES:
        assert name in line["secondary"], name
        assert '"%s": {' % name in tail, "%s starts before the last 8000 characters of the line (%d)" % (name, len(raw))
    assert set(line["notes"]) >= {"frac", "real_frac", "cpu", "parity", "peaks", "full_record"}


def test_every_workload_carries_real_frac_and_a_cpu_leg(line):
    """VERDICT r5 next 2 / missing 3: the real fraction next to every equivalent one, a CPU figure beside every configuration."""
    r = line["roofline"]
    assert 0.0 < r["real_frac"] <= 1.0 and r["real_frac_roof"] in ("hbm", "fp64")
    assert line["cpu_baseline"]["variant"] == "optimised" and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    assert line["cpu_baseline_checker"]["variant"] == "fidelity checker"
    assert line["cpu_baseline"]["models_per_s"] > line["cpu_baseline_checker"]["models_per_s"]
    for name in ("c4", "c4_state_variances", "c4_full_sym", "c5"):
        s = line["secondary"][name]
        assert "error" not in s, (name, s)
        assert 0.0 < s["roofline"]["real_frac"] <= 1.0, (name, s["roofline"])
        assert s["roofline"]["real_frac"] <= s["roofline"]["frac"] * 1.02 + 0.4, (name, s["roofline"])
        assert s["cpu"]["models_per_s"] > 0 and s["cpu"]["cores"] >= 1 and s["cpu"]["n"] >= 16, (name, s["cpu"])
        assert s["chk"]["models_per_s"] > 0 and s["chk"]["n"] >= 16, (name, s["chk"])
        assert s["parity"]["mle"] < 1e-9 and s["parity"]["n_mle"] >= 256, (name, s["parity"])
    for name in ("f1_calibration", "f1_calibration_32x4"):
        s = line["secondary"][name]
        assert 0.0 < s["real_frac"] < 1.0 and s["cpu"]["models_per_s"] > 0 and s["cpu"]["n"] >= 2, (name, s)
        # against scipy L-BFGS-B (the reference's optimiser and differencing) on the CPU objective, same models: the GPU calibration
        # never ends HIGHER than scipy beyond its stopping tolerance (obj_below_scipy: where scipy's differenced search stalled first)
        assert s["parity"]["obj_above_scipy"] < 2e-6 and s["parity"]["n"] >= 2, (name, s["parity"])


def test_headline_is_configs1_with_roofline(line, full):
    c = line["config"]
    assert (c["name"], c["batch_per_gpu"], c["total_batch"], c["series"], c["factors"], c["T"]) == ("c2", 4096, 4096, 8, 2, 1000)
    assert line["dtype"] == "f64" and line["unit"] == "model-timesteps/s" and line["n_gpus"] == 1
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["real_frac"] - r["frac"]) < 1e-4                       # HBM is the nearer roof of the 16-lane smoother
    assert line["models_per_s"] > 1.0e6          # floor: 1.0 M models/s (measured 1.32-1.36 M); north star 100 k
    assert r["frac"] > 0.40                      # north star: >= 40 % of the HBM roofline on the dominant kernel
    # HBM traffic of the dominant kernel: measured in this very run (rocprofv3 PMC child passes), within the record
    # pad of the algorithmic bytes
    dom = full["roofline"]["kernels"][r["kernel"]]
    assert dom["traffic_source"].startswith("measured in this run"), dom
    assert 0.98 < r["traffic"] / r["algorithmic_bytes"] < 1.10


def test_secondary_configs3_throughput_floor(line, full):
    s, f = line["secondary"]["c4"], full["secondary"]["c4"]
    assert "error" not in s, s
    assert "32-series/4-factor" in f["workload"] and "T=2000" in f["workload"]
    assert f["roofline"]["bound"] == "fp64" and f["roofline"]["unit"] == "TFLOP/s" and f["roofline"]["peak"] == 78.6
    assert "tape" in f["workload"]               # the inverse-free path (round 4) is what the default run measures
    # the floor IS the north-star bar in SURVEY 8d's full-output accounting (0.40 x 8 TB/s / 85.76 MB per model = 37 313 models/s;
    # a floor of 32 k would have stayed green with configs[3] back under it)
    assert 37300.0 < s["bar_models_per_s"] < 37330.0 and abs(s["bar_models_per_s"] - BAR_C4) < 1.0
    assert s["models_per_s"] >= s["bar_models_per_s"] and s["models_per_s"] >= C4_FLOOR, s
    assert 0.0 < s["roofline"]["frac"] < 1.0
    # the real fraction is printed beside the equivalent ones: 0.2x executed against 0.4x / 0.6 equivalent
    assert s["roofline"]["real_frac"] < s["roofline"]["frac"] and s["roofline"]["real_frac"] < s["roofline"]["8d_frac"]
    # error figures against the oracle ride in the line
    par = s["parity"]
    assert "error" not in par, par
    assert par["n_mle"] >= 256 and par["mle"] < 1e-9, par
    assert par["n"] >= 32 and par["sim_means"] < 1e-9 and par["sim_vars"] < 1e-9, par
    # HBM traffic of its kernels measured in the run
    for k in f["roofline"]["kernels"].values():
        assert k.get("traffic_source", "").startswith("measured in this run"), k


def test_secondary_state_variances_of_wide_models(line, full):
    """smooth_state_variances on configs[3]'s batch in the driver's line -- on the state tape, parity
    figures against the oracle riding along."""
    s, f = line["secondary"]["c4_state_variances"], full["secondary"]["c4_state_variances"]
    assert "error" not in s, s
    assert s["state_tape"] and "STATE tape" in f["workload"] and "T=2000" in f["workload"]
    par = s["parity"]
    assert par["n_mle"] >= 256 and par["mle"] < 1e-9, par
    assert par["n"] >= 32 and par["S"] < 1e-9 and par["var"] < 1e-9, par
    assert f["roofline"]["bound"] == "fp64" and "smoother_dk_kernel" in f["roofline"]["kernels"]
    assert s["models_per_s"] >= STATE_VARIANCES_FLOOR, s


def test_secondary_full_symmetric_records_of_wide_models(line, full):
    """VERDICT r5 missing 2: all six reference outputs (kalmanfilter.py:392-400, 453-474) of configs[3]'s batch as packed-symmetric
    records -- 138 GB resident -- in the driver's line with parity on every array and its real GB/s."""
    s, f = line["secondary"]["c4_full_sym"], full["secondary"]["c4_full_sym"]
    assert "error" not in s, s
    assert "packed-symmetric" in f["workload"] and "T=2000" in f["workload"] and "batch=4096" in f["workload"]
    par = s["parity"]
    assert par["n_mle"] >= 256 and par["mle"] < 1e-9 and par["n"] >= 32, par
    for k in ("Xp", "Pp", "F", "Pf"):
        assert par[k] < 1e-10, (k, par)
    assert par["S"] < 1e-9 and par["Ps"] < 1e-9, par
    ks = f["roofline"]["kernels"]
    assert set(ks) == {"filter_split_kernel", "smoother_mfma_kernel"}   # (records are in the state basis: the round-3 split filter)
    resident = 3 * 4096 * 2000 * 8 * f["record_stride_doubles"]
    assert 130e9 < resident < 145e9, resident
    assert abs(ks["smoother_mfma_kernel"]["algorithmic_GB"] - 4096 * 2000 * 8 * 2 * (36 + 36 * 37 // 2) / 1e9) < 0.01
    assert s["models_per_s"] >= 28000.0, s      # floor; measured 31-32 k (the bar of 37.3 k is NOT met: DESIGN section 6)


def test_secondary_generic_kernels(line):
    """what a shape without a specialised module runs, measured -- the size-generic kernel family at
    configs[1]'s and configs[3]'s shapes, parity at the specialised kernels' bar."""
    g2, g4 = line["secondary"]["generic_c2"], line["secondary"]["generic_c4"]
    assert "error" not in g2 and "error" not in g4, (g2, g4)
    assert g2["parity"]["mle"] < 1e-9 and g2["parity"]["S"] < 1e-9 and g2["parity"]["Ps"] < 1e-9, g2["parity"]
    assert g4["parity"]["mle"] < 1e-9 and g4["parity"]["sim_means"] < 1e-9 and g4["parity"]["sim_vars"] < 1e-9, g4["parity"]
    assert g2["models_per_s"] > 120000.0 and g4["models_per_s"] > 1800.0, (g2, g4)   # floors (round 6, second form of the family: 180 k / 2.7 k measured; the first: 19.8 k / 700); the ratio to the specialised rate is in INTEGRATION.md
    assert g2["models_per_s"] < line["models_per_s"] and g4["models_per_s"] < line["secondary"]["c4"]["models_per_s"]
    # a shape ONLY this family serves (100 states): all six outputs against the oracle, and the host's cores beside it
    g9 = line["secondary"]["generic_96x4"]
    assert "error" not in g9, g9
    assert g9["parity"]["mle"] < 1e-9 and g9["parity"]["S"] < 1e-9 and g9["parity"]["Ps"] < 1e-9 and g9["parity"]["Pf"] < 1e-9, g9["parity"]
    assert g9["models_per_s"] > 700.0, g9                         # floor (1 060 measured; the first form of the family: 250)
    assert g9["cpu"]["cores"] >= 1 and g9["models_per_s"] > 3.0 * g9["cpu"]["models_per_s"], g9   # 9.5 x 16 cores measured


def test_secondary_configs4_solver_loop(line, full):
    s = line["secondary"]["c5"]
    assert "error" not in s, s
    assert s["roofline"]["bound"] == "fp64"
    assert s["objective_evaluations_per_s"] >= 4.5e6, s   # floor (round 2: 6.4 M evaluations/s)
    assert s["parity"]["n_mle"] >= 256 and s["parity"]["mle"] < 1e-9, s["parity"]
    # the objective exploits Z = [I | G]; the line says what it executes, and the note says frac is equivalent
    assert s["roofline"]["real_frac"] < 0.9 * s["roofline"]["frac"]
    assert "work-equivalent" in full["secondary"]["c5"]["roofline"]["note"]


def test_secondary_factor_analysis_and_calibration(line):
    """Rows f4 and f1 in the driver-run line: throughput floors well under the round-3 measurements (400 k models/s for the
    factor analysis of 8-series models, 8 k models/s for the calibration of 8192 8/2 models)."""
    f4 = line["secondary"]["f4_factor_analysis"]
    assert "error" not in f4, f4
    assert f4["models_per_s"] >= 50000.0, f4
    assert sum(f4["nfactors_histogram"].values()) == 4096 and "0" not in f4["nfactors_histogram"]
    f1 = line["secondary"]["f1_calibration"]
    assert "error" not in f1, f1
    assert f1["converged_frac"] > 0.95 and f1["frac_at_or_below_true_parameter_objective"] > 0.95, f1
    f1w = line["secondary"]["f1_calibration_32x4"]                 # wide models beside it: 512 x (32,4), T = 500
    assert "error" not in f1w and f1w["converged_frac"] > 0.95 and f1w["models_per_s"] >= 100.0, f1w
    # (round 5: 3.9 s for 512 models TO CONVERGENCE, 200 iterations -- profiles/r05/ab_line_search.log; the round-3 figure of
    # 2.75 s was 60 iterations with a third of the models converged)
    assert f1["models_per_s"] >= 10000.0, f1    # round 5: 12-18.6 k (0.44-0.68 s for 8192 models); round 4: 10.9 k on its lease
    # round 6: maxiter counts quasi-Newton iterations per model (the round-5 advice): nobody stops at the limit, and what the
    # first calibration of a process pays on top of the timed one is in the line
    assert f1["models_at_the_iteration_limit"] == 0 and f1w["models_at_the_iteration_limit"] == 0, (f1, f1w)
    assert f1["iterations"] < 200 and f1w["iterations"] < 200 and f1["first_use_warmup_s"] > 0


def test_secondary_dropin_configs0(line, full):
    """BASELINE configs[0] in the driver's line: the unmodified reference class on examples/data, solve()
    + get_simulation() with the reference's own engine on the host and with the HIP engine installed, plus plug point A."""
    c1, f = line["secondary"]["c1_dropin"], full["secondary"]["c1_dropin"]
    assert "error" not in c1, c1
    ref = c1["ref_host"]
    assert abs(ref["obj"] - 2332.3270694) < 1e-5 and ref["nfev"] == 77
    assert abs(f["reference_engine_on_host"]["obj"] - 2332.3270694) < 1e-5
    for k in ("hip_engine_scipy_solver", "hip_solver_fd", "hip_solver_adjoint"):
        assert abs(f[k]["obj"] - 2332.3270694) < 1e-5, (k, f[k])
        assert f[k]["solve_s"] > 0 and f[k]["get_simulation_s"] > 0 and f[k]["simulation_rows"] == f["reference_engine_on_host"]["simulation_rows"]
    assert abs(c1["engine_scipy_solver"]["nfev"] - 77) <= 14
    assert c1["solver_adjoint"]["nfev"] < 40
    # round 5: the single-record route walks the observed steps only, simulate / decompose run on the device:
    # 0.164 s / 0.016 s measured through the unmodified class (round 4: 0.374 / 0.049), 0.093 s with the class's set_observations
    # vectorised; the bars asked for were 0.15 / 0.02 -- floors with a margin for a slow host
    assert c1["engine_scipy_solver"]["solve_s"] < 0.15 and c1["engine_scipy_solver"]["get_simulation_s"] < 0.03, c1
    assert c1["solver_fd"]["solve_s"] < 0.05, c1     # 0.010 s: plug point A no longer pays the 70 ms loop either


def test_secondary_dropin_wide(line, full):
    """VERDICT r5 missing 5 / next 3c: ONE 32-series / 4-factor model through the unmodified class -- the latency-bound regime a
    Metran user is in -- with objective parity against the reference engine at the optimum each route found."""
    c = line["secondary"]["c1w_dropin"]
    assert "error" not in c, c
    for k in ("engine_scipy_solver", "solver_fd", "solver_adjoint"):
        assert c[k]["obj_rel_err_vs_reference_engine_at_the_same_optimum"] < 1e-9, (k, c[k])
        assert c[k]["solve_s"] > 0 and c[k]["nfev"] > 0
    objs = [c[k]["obj"] for k in ("engine_scipy_solver", "solver_fd", "solver_adjoint")]
    assert max(objs) - min(objs) < 1e-3 * abs(objs[0])                      # the three routes end in the same basin
    assert c["ref_host"]["get_mle_s"] > 0 and c["ref_host"]["solve_s_extrapolated"] > c["solver_fd"]["solve_s"]


def test_secondary_factor_analysis_wide(line, full):
    """Row f4 where it is hard: 4096 x 32 series with four true factors -- (nearly) every model takes the
    multi-factor path (varimax, the host eig order), and the host / device split of the call is in the record."""
    f4 = line["secondary"]["f4_factor_analysis_32x4"]
    assert "error" not in f4, f4
    hist = {int(k): v for k, v in f4["nfactors_histogram"].items()}
    assert sum(hist.values()) == 4096 and sum(v for k, v in hist.items() if k >= 2) >= 4000, hist   # the MAP test's choice
    sp = f4["split_s"]
    assert sp["host_eig_order"] > 0 and sp["device_kernels_and_transfers"] > 0
    assert f4["models_per_s"] >= 8000.0, f4          # round 3 (builder-side probe): 26 k models/s
    sub = f4["always_scipy"]
    assert sub["models"] == 256
