"""CPU: the accounting behind bench.py's ``roofline`` object is SURVEY.md section 8(d)'s -- the per-model byte and flop figures
the survey prints (B_fs = 8 T [N + 4c], B_ll = 8 T N, the 40 %-of-HBM bars) come out of ``bench.algorithmic_bytes`` /
``algorithmic_flops`` / ``build_roofline`` for the BASELINE configurations, and the record the line carries is consistent
(frac = achieved / peak, achieved = algorithmic work of one launch / its duration).  No GPU: the timings are made up."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_full_output_bytes_are_the_surveys():
    ab = bench.algorithmic_bytes(8, 2, 1000, "full")                      # configs[1] / [2]: c = 110
    assert ab == {"filter": 8 * 1000 * (8 + 2 * 110), "smoother": 8 * 1000 * 2 * 110}
    assert sum(ab.values()) == 3_584_000                                  # B_fs
    assert sum(bench.algorithmic_bytes(8, 2, 1000, "full", sym=True).values()) == 2_144_000      # packed c_s = 65
    assert sum(bench.algorithmic_bytes(32, 4, 2000, "full").values()) == 85_760_000              # configs[3], c = 1332
    assert sum(bench.algorithmic_bytes(32, 4, 2000, "full", sym=True).values()) == 45_440_000
    assert bench.algorithmic_bytes(8, 2, 1000, "solver") == {"filter": 64_000, "smoother": 0}    # B_ll
    assert sum(bench.algorithmic_bytes(5, 1, 6255, "full").values()) == 8 * 6255 * (5 + 4 * 42)  # configs[0]: 8.66 MB


def test_projection_paths_move_fewer_bytes_than_the_full_pass():
    full = sum(bench.algorithmic_bytes(32, 4, 2000, "full").values())
    rec = bench.algorithmic_bytes(32, 4, 2000, "project")
    tape = bench.algorithmic_bytes(32, 4, 2000, "project", tape=True)
    c, n, N, T = 36 + 36 * 36, 36, 32, 2000
    assert rec == {"filter": 8 * T * (N + c), "smoother": 8 * T * (c + 2 * N)}
    # the tape: N entries of n + 4 doubles per step, written once and read once (mk_tape_stride, include/metran_hip.h)
    assert tape == {"filter": 8 * T * (N + N * (n + 4)), "smoother": 8 * T * (N * (n + 4) + 2 * N)}
    assert sum(tape.values()) < sum(rec.values()) < full


def test_flops_are_the_surveys():
    fl = bench.algorithmic_flops(8, 2, 1000, "full")
    assert abs(fl["filter"] / 1000 - 4.8e3) < 0.1e3 and abs(fl["smoother"] / 1000 - 6.7e3) < 0.1e3   # 11.5 kflop / step
    fl = bench.algorithmic_flops(32, 4, 2000, "project", missing=0.3)
    assert abs(fl["filter"] / 2000 - 154e3) < 2e3 and abs(fl["smoother"] / 2000 - 300e3) < 2e3        # 455 kflop / step
    assert abs(sum(fl.values()) - 0.91e9) < 0.01e9
    assert bench.algorithmic_flops(8, 2, 1000, "solver")["smoother"] == 0.0
    ex = bench.executed_flops_tape(32, 4, 2000, 0.3)
    assert sum(ex.values()) < 0.5 * sum(fl.values())                      # the inverse-free formulation executes far fewer


@pytest.mark.parametrize("cfg", ["c2", "c4"])
def test_roofline_record_is_consistent(cfg, monkeypatch):
    B, N, K, T, missing, mode = bench.CONFIGS[cfg]
    monkeypatch.setattr(bench, "pmc_traffic", lambda *a, **k: (None, "not looked up in this test"))
    tape = cfg == "c4"
    f_ms, s_ms = (1.4, 1.7) if cfg == "c2" else (50.0, 52.0)
    r = bench.build_roofline(cfg, N, K, T, B, mode, missing, f_ms, s_ms, False, tape=tape)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-15
    dom = r["kernels"][r["kernel"]]
    assert dom["ms"] == max(k["ms"] for k in r["kernels"].values()) == r["avg_launch_ms"]
    if cfg == "c2":
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
        # achieved = algorithmic bytes of ONE launch of the dominant kernel / its duration
        assert abs(r["achieved"] - B * 8 * T * 2 * 110 / 1e9 / (s_ms / 1e3)) < 1e-9
        assert r["algorithmic_bytes"] == pytest.approx(B * 8 * T * 2 * 110)
    else:
        assert r["bound"] == "fp64" and r["peak"] == 78.6 and r["kernel"] == "smoother_dk_kernel"
        assert "filter_obs_kernel" in r["kernels"]          # round 6: the tape's default writer (the filter in the observable basis)
        r_state = bench.build_roofline(cfg, N, K, T, B, mode, missing, f_ms, s_ms, False, tape=tape, tape_filter="state")
        assert "filter_split_kernel" in r_state["kernels"]  # ... and the round-4 writer behind --tape-filter state
        # real_frac: the utilisation of the NEARER roof from what the kernels execute; never above the work-equivalent frac here
        for rr in (r, r_state):
            d = rr["kernels"][rr["kernel"]]
            assert rr["real_frac"] == pytest.approx(max(d["GBps"] / 8000.0, d["executed_TFLOPps"] / 78.6)) and 0 < rr["real_frac"] < rr["frac"]
        # (the observable-basis writer executes MORE useful flops -- its prediction is 2 + 3K multiply-adds per covariance element --
        # in FEWER instructions: the state-basis writer's per-update work is replicated on every lane and counted once)
        assert r["kernels"]["filter_obs_kernel"]["executed_TFLOP"] > r_state["kernels"]["filter_split_kernel"]["executed_TFLOP"]
        assert r["algorithmic_flops"] == pytest.approx(B * bench.algorithmic_flops(N, K, T, mode, missing)["smoother"])
    fs = r["survey_8d_full_output_accounting"]
    assert fs["bytes_per_model"] == (3_584_000 if cfg == "c2" else 85_760_000)
    bar = fs["north_star_bar"]["models_per_s"]                            # section 8(d): 893 k and 37.3 k models/s
    assert abs(bar - (893e3 if cfg == "c2" else 37.3e3)) < (1e3 if cfg == "c2" else 0.05e3)
    models_per_s = B / ((f_ms + s_ms) / 1e3)
    assert fs["frac_of_peak"] == pytest.approx(models_per_s * fs["bytes_per_model"] / 8e12)
    assert (fs["frac_of_peak"] >= 0.40) == (models_per_s >= bar)


def test_parity_sample_covers_the_batch_and_the_wavefront_positions():
    s = bench.sample_models(4096, 256)
    assert len(s) >= 256 and len(set(s.tolist())) == len(s) and s.min() == 0 and s.max() == 4095
    assert {0, 1, 2, 3}.issubset(set((s % 4).tolist())) and np.all(np.diff(s) > 0)
    assert np.array_equal(bench.sample_models(10, 256), np.arange(10))
    assert np.array_equal(bench.sample_models(4096, 32), np.unique(bench.sample_models(4096, 32)))


def test_kernel_source_sha_covers_every_kernel_source():
    import glob

    listed = open(os.path.join(ROOT, "bench.py")).read()
    for f in glob.glob(os.path.join(ROOT, "metran_amd", "csrc", "mk_*.hip")):
        name = os.path.basename(f)
        if name in ("mk_capi.hip", "mk_factor.hip", "mk_ingest.hip", "mk_shape.hip", "mk_generic.hip", "mk_lbfgs.hip"):
            continue    # not filter / smoother kernels of a bench configuration (mk_generic: unspecialised shapes; mk_lbfgs: the calibration driver)
        assert ("metran_amd/csrc/" + name) in listed, name
    assert len(bench.kernel_source_sha()) == 16


def test_cpu_legs_take_the_cpus_the_process_may_use():
    """bench.py's CPU legs and the oracle's OpenMP loops start no more threads than the process may use -- its affinity mask,
    capped by the cgroup CPU quota (the GPU boxes: 256 hardware threads, quota 16; 128 threads ran at 0.6 x the rate of 16,
    profiles/r05/cpu_leg_threads.log)."""
    import os

    import bench
    import oracle

    logical, quota = bench.host_cpus()
    assert logical >= 1 and (quota is None or quota > 0)
    t = bench.cpu_leg_threads(128)
    assert 1 <= t <= min(128, logical) and (quota is None or t <= max(1, round(quota)))
    assert bench.cpu_leg_threads(1) == 1
    assert 1 <= oracle.usable_cpus() <= (os.cpu_count() or 1)
    assert oracle.num_threads() <= oracle.usable_cpus() and oracle.fast_num_threads() <= oracle.usable_cpus()
