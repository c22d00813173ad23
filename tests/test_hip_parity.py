"""GPU parity tests: HIP kernels (through the C ABI / ctypes) vs the reference-generated
goldens and vs the CPU oracle on seeded inputs.

Tolerances (fp64): -2 log L within 1e-9 relative (north-star bar; observed ~1e-13);
filtered/predicted moments 1e-10 absolute on O(1) states; smoothed moments 1e-9 (Cholesky
solve vs the reference's SVD pseudo-inverse, cond(Pp) ~ 1e1-1e3).
"""
import numpy as np
import pytest

import oracle
from conftest import SYNTH_GOLDENS, golden_models, rel_err
from metran_amd.synthetic import make_dfm_batch

pytestmark = pytest.mark.gpu

MLE_RTOL = 1e-9
FILT_ATOL = 1e-10
SMOOTH_ATOL = 1e-9


@pytest.fixture(scope="module", params=["model_major", "time_major", "time_major-shipped-defaults"])
def kf(request):
    """Both memory layouts of the per-step arrays ([B,T,...] and [T,B,...] behind [B,T,...] views) under the tier's kernel
    choice (conftest: the split wide filter forced, because the tier's batches are small), and once more with the SHIPPED
    defaults (``wide_filter: auto`` -- one state per lane for small batches, the split layout above two
    models per SIMD -- used to be reached by two tests only): goldens, seeded batches and the full-size configs[3] test run
    under what a user gets."""
    import torch

    from metran_amd.engine import BatchedKalman

    assert torch.cuda.is_available()
    layout = request.param.split("-")[0]
    k = BatchedKalman(0, layout=layout)
    if request.param.endswith("shipped-defaults"):
        for which, (_, names) in BatchedKalman._VARIANTS.items():
            k.set_variant(which, names[0])          # value 0 of every selector = the library's default
        assert k.get_variant("wide_filter") == "auto"
    return k


def _np(t):
    return t.detach().cpu().numpy()


def _check_against(r, ref, i, ts=None, smooth=True):
    sc = int(ref["sigmacount"])
    assert int(_np(r["sigmacount"])[i]) == sc
    assert abs(_np(r["mle"])[i] - float(ref["mle"])) <= MLE_RTOL * abs(float(ref["mle"]))
    np.testing.assert_allclose(_np(r["sigmas"])[i, :sc], np.asarray(ref["sigmas"])[:sc], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(_np(r["detfs"])[i, :sc], np.asarray(ref["detfs"])[:sc], rtol=0, atol=1e-11)
    assert not _np(r["sigmas"])[i, sc:].any() and not _np(r["detfs"])[i, sc:].any()
    sel = slice(None) if ts is None else ts
    np.testing.assert_allclose(_np(r["F"])[i], ref["F"], rtol=0, atol=FILT_ATOL)
    np.testing.assert_allclose(_np(r["Xp"])[i], ref["Xp"], rtol=0, atol=FILT_ATOL)
    np.testing.assert_allclose(_np(r["Pf"])[i][sel], ref["Pf"], rtol=0, atol=FILT_ATOL)
    np.testing.assert_allclose(_np(r["Pp"])[i][sel], ref["Pp"], rtol=0, atol=FILT_ATOL)
    if smooth:
        np.testing.assert_allclose(_np(r["S"])[i], ref["S"], rtol=0, atol=SMOOTH_ATOL)
        np.testing.assert_allclose(_np(r["Ps"])[i][sel], ref["Ps"], rtol=0, atol=SMOOTH_ATOL)


@pytest.mark.parametrize("fname", SYNTH_GOLDENS)
def test_golden_synthetic(kf, fname):
    """Every reference-generated synthetic fixture (incl. missing data, empty first step,
    never-observed series, single observation, inf as missing)."""
    for i, m in golden_models(fname):
        kf.set_observations(m["obs"][None]).set_loadings(m["loadings"][None])
        r = kf.filter_smooth(m["phi"][None], m["q"][None])
        _check_against(r, m, 0, ts=m["tsel"])
        assert int(_np(r["status"])[0]) == 0


def test_golden_g1_real_data(kf, g1):
    """examples/data, T=6255 with 5912 empty steps."""
    kf.set_observations(g1["obs"][None]).set_loadings(g1["loadings"][None])
    r = kf.filter_smooth(g1["phi"][None], g1["q"][None])
    mle = float(_np(r["mle"])[0])
    assert abs(mle - 2332.327069381027) <= MLE_RTOL * 2332.0   # reference value
    assert round(mle, 2) == 2332.33                              # notebook-stored value
    ref = dict(g1)
    ref["sigmacount"] = len(g1["sigmas"])
    ref["mle"] = g1["mle_star"]
    _check_against(r, ref, 0, ts=g1["tsel"])
    # device-side parameter map (Metran._get_matrices) from alpha
    phi, q = kf.params_from_alpha(g1["alpha_star"][None])
    np.testing.assert_allclose(_np(phi)[0], g1["phi"], rtol=1e-15)
    np.testing.assert_allclose(_np(q)[0], g1["q"], rtol=1e-14)
    mle10 = float(_np(kf.loglik(*kf.params_from_alpha(g1["alpha_10"][None])))[0])
    assert abs(mle10 - 2384.792799342231) <= MLE_RTOL * 2384.0
    # projection epilogues (simulate / decompose)
    sm, sv = kf.simulate(g1["Z_scaled"], r["S"], r["Ps"])
    np.testing.assert_allclose(_np(sm)[0], g1["sim_means"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(_np(sv)[0], g1["sim_vars"], rtol=0, atol=1e-9)
    sdf, cdf = kf.decompose(g1["Z_scaled"], r["S"])
    ts = g1["tsel"]
    np.testing.assert_allclose(_np(sdf)[0][ts], g1["sdf_means"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(_np(cdf)[0][:, ts], g1["cdf_means"], rtol=0, atol=1e-9)


def test_golden_g1_masked(kf, g1):
    obs = g1["obs"].copy()
    obs[int(g1["mask_t"]), 4] = np.nan
    kf.set_observations(obs[None]).set_loadings(g1["loadings"][None])
    r = kf.filter_smooth(g1["phi"][None], g1["q"][None])
    want = float(g1["masked_mle_star"])
    assert abs(float(_np(r["mle"])[0]) - want) <= MLE_RTOL * abs(want)
    sm, _ = kf.simulate(g1["Z_scaled"], r["S"], r["Ps"])
    np.testing.assert_allclose(_np(sm)[0][:, 4] + g1["oseries_mean"][4], g1["masked_sim_005"].ravel(), atol=1e-8)


def test_golden_g2_seeded(kf, g2):
    kf.set_observations(g2["obs"][None]).set_loadings(g2["loadings"][None])
    r = kf.filter_smooth(g2["phi"][None], g2["q"][None])
    assert abs(float(_np(r["mle"])[0]) - 2431.3389452203646) <= MLE_RTOL * 2431.0
    np.testing.assert_allclose(_np(r["S"])[0], g2["S"], atol=SMOOTH_ATOL)
    np.testing.assert_allclose(_np(r["Ps"])[0][g2["tsel"]], g2["Ps"], atol=SMOOTH_ATOL)


@pytest.mark.parametrize("N,K,T,B,missing", [(8, 2, 200, 37, 0.0), (8, 2, 120, 64, 0.2), (5, 1, 90, 19, 0.3),
                                              (2, 1, 150, 5, 0.1), (32, 4, 40, 6, 0.3), (14, 3, 50, 3, 0.2)])
def test_vs_oracle_seeded(kf, N, K, T, B, missing):
    """Seeded batches (ragged batch sizes: not multiples of the models-per-workgroup)."""
    d = make_dfm_batch(B, N, K, T, seed=100 * N + K, missing=missing, first_step="random")
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    r = kf.filter_smooth(d["phi"], d["q"])
    assert rel_err(_np(r["mle"]), ref["mle"]) < MLE_RTOL
    np.testing.assert_array_equal(_np(r["sigmacount"]), ref["sigmacount"])
    for k, tol in (("F", FILT_ATOL), ("Pf", FILT_ATOL), ("Xp", FILT_ATOL), ("Pp", FILT_ATOL), ("S", SMOOTH_ATOL),
                   ("Ps", SMOOTH_ATOL), ("sigmas", 1e-10), ("detfs", 1e-10)):
        np.testing.assert_allclose(_np(r[k]), ref[k], rtol=0, atol=tol, err_msg=k)
    assert not _np(r["status"]).any()


def test_loglik_only_and_shared_records(kf):
    """S parameter sets per observation record (instance i reads record i % R): the layout the
    finite-difference gradient of solver.py needs (metran/solver.py:248-255)."""
    R, S, N, K, T = 6, 11, 8, 2, 100
    d = make_dfm_batch(R, N, K, T, seed=5, missing=0.1)
    rng = np.random.default_rng(0)
    alpha = rng.uniform(3, 50, size=(S, R, N + K))
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    phi, q = kf.params_from_alpha(alpha.reshape(S * R, -1))
    mle = _np(kf.loglik(phi, q)).reshape(S, R)
    from metran_amd.params import phi_q_from_alpha

    for s in range(S):
        p, qq = phi_q_from_alpha(alpha[s], d["loadings"])
        ref = oracle.dfm_batch(d["obs"], p, qq, d["loadings"], smooth=False, outputs="mle")
        assert rel_err(mle[s], ref["mle"]) < MLE_RTOL


def test_initial_state_and_obsvar(kf):
    """Non-default x0 / P0 (run_filter arguments, kalmanfilter.py:696-750) and R > 0."""
    N, K, T, B = 8, 2, 60, 4
    d = make_dfm_batch(B, N, K, T, seed=9, missing=0.15)
    rng = np.random.default_rng(1)
    n = N + K
    x0 = rng.normal(size=(B, n))
    A = rng.normal(size=(B, n, n))
    P0 = A @ A.transpose(0, 2, 1) / n + np.eye(n)
    obsvar = rng.uniform(0.01, 0.2, size=(B, N))
    kf.set_observations(d["obs"]).set_loadings(d["loadings"], obsvar)
    r = kf.filter_smooth(d["phi"], d["q"], x0=x0, P0=P0)
    from metran_amd.params import observation_matrix

    for b in range(B):
        o, oi, oc = oracle.set_observations(d["obs"][b])
        sg, df, sc, F, Pf, Xp, Pp = oracle.seqkalmanfilter(o, np.diag(d["phi"][b]), np.diag(d["q"][b]),
                                                           observation_matrix(d["loadings"][b]), obsvar[b], oi, oc,
                                                           x0[b], P0[b])
        S, Ps = oracle.kalmansmoother(F, Pf, Xp, Pp, np.diag(d["phi"][b]))
        np.testing.assert_allclose(_np(r["F"])[b], F, atol=FILT_ATOL)
        np.testing.assert_allclose(_np(r["Pf"])[b], Pf, atol=FILT_ATOL)
        np.testing.assert_allclose(_np(r["S"])[b], S, atol=SMOOTH_ATOL)
        np.testing.assert_allclose(_np(r["Ps"])[b], Ps, atol=SMOOTH_ATOL)
        assert abs(_np(r["mle"])[b] - oracle.get_mle(sg[:sc], df[:sc], oc)) < 1e-9 * abs(_np(r["mle"])[b])


def test_full_size_properties(kf):
    """BASELINE configs[1] size (B=4096, 8 series / 2 factors, T=1000) through size-independent
    properties: (1) a sub-sample equals the oracle, (2) results do not depend on batch position
    or batch size, (3) smoothed == filtered at the last step, (4) smoothed variances <= filtered,
    (5) covariances symmetric, (6) at fully observed steps Z x_f reproduces the observation (R = 0)."""
    import torch

    from metran_amd.synthetic import make_dfm_batch_torch

    B, N, K, T = 4096, 8, 2, 1000
    d = make_dfm_batch_torch(B, N, K, T, seed=123, device=kf.device)
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    r = kf.filter_smooth(d["phi"], d["q"], outputs=("F", "Pf", "S", "Ps"))
    torch.cuda.synchronize()
    assert not r["status"].any()
    # 72 models: the first and last wavefronts whole (every DPP row and every position of a four-model wavefront), then an
    # odd stride through the batch (6 of 4096 were compared before)
    idx = sorted(set(list(range(16)) + list(range(B - 16, B)) + (np.linspace(16, B - 17, 40).astype(np.int64) | 1).tolist()))
    assert len(idx) >= 64 and {i % 4 for i in idx} == {0, 1, 2, 3}
    sub = {k: _np(d[k][idx]) for k in ("obs", "phi", "q", "loadings")}
    ref = oracle.dfm_batch(sub["obs"], sub["phi"], sub["q"], sub["loadings"])
    assert rel_err(_np(r["mle"][idx]), ref["mle"]) < MLE_RTOL
    np.testing.assert_allclose(_np(r["S"][idx]), ref["S"], atol=SMOOTH_ATOL)
    np.testing.assert_allclose(_np(r["Ps"][idx]), ref["Ps"], atol=SMOOTH_ATOL)
    # (2) same models as a smaller, differently aligned batch
    kf2_obs = d["obs"][idx].contiguous()
    kf.set_observations(kf2_obs).set_loadings(d["loadings"][idx].contiguous())
    r2 = kf.filter_smooth(d["phi"][idx].contiguous(), d["q"][idx].contiguous(), outputs=("F", "Pf", "S", "Ps"))
    assert torch.equal(r2["mle"], r["mle"][idx]) and torch.equal(r2["Ps"], r["Ps"][idx])
    # (3)-(6)
    assert torch.equal(r["S"][:, -1], r["F"][:, -1]) and torch.equal(r["Ps"][:, -1], r["Pf"][:, -1])
    dPs = torch.diagonal(r["Ps"], dim1=-2, dim2=-1)
    dPf = torch.diagonal(r["Pf"], dim1=-2, dim2=-1)
    assert bool((dPs <= dPf + 1e-12).all())
    assert float((r["Ps"] - r["Ps"].transpose(-1, -2)).abs().max()) < 1e-12
    zx = r["F"][..., :N] + torch.einsum("bnk,btk->btn", d["loadings"], r["F"][..., N:])
    assert float((zx - d["obs"]).abs().max()) < 1e-9


@pytest.mark.parametrize("B,N,K,T", [(1, 2, 1, 1), (3, 2, 1, 1), (17, 3, 1, 2), (5, 8, 2, 3), (2, 32, 4, 1)])
def test_degenerate_sizes(kf, B, N, K, T):
    """One- and two-step records, a batch that does not fill a workgroup, and a record with no observation
    at all (sigmacount 0, objective 0): every output equals the oracle's."""
    d = make_dfm_batch(B, N, K, T, seed=9)
    d["obs"][0, :, :] = np.nan
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    r = kf.filter_smooth(d["phi"], d["q"])
    np.testing.assert_array_equal(_np(r["sigmacount"]), ref["sigmacount"])
    np.testing.assert_allclose(_np(r["mle"]), ref["mle"], rtol=MLE_RTOL, atol=1e-12)
    assert float(_np(r["mle"])[0]) == 0.0
    for k, tol in (("F", FILT_ATOL), ("Pf", FILT_ATOL), ("Xp", FILT_ATOL), ("Pp", FILT_ATOL), ("S", SMOOTH_ATOL),
                   ("Ps", SMOOTH_ATOL)):
        np.testing.assert_allclose(_np(r[k]), ref[k], rtol=0, atol=tol)
    np.testing.assert_allclose(_np(kf.loglik(d["phi"], d["q"])), ref["mle"], rtol=MLE_RTOL, atol=1e-12)


def test_minus_1e10_is_dropped_like_the_reference(kf):
    """set_observations' "+1e10 then nonzero()" (kalmanfilter.py:666-667) drops a finite -1e10; the oracle
    restates it and the engine maps the value to missing on upload."""
    d = make_dfm_batch(3, 5, 1, 40, seed=10)
    d["obs"][1, 7, 2] = -1e10
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    r = kf.filter_smooth(d["phi"], d["q"])
    np.testing.assert_allclose(_np(r["mle"]), ref["mle"], rtol=MLE_RTOL)
    np.testing.assert_allclose(_np(r["S"]), ref["S"], rtol=0, atol=SMOOTH_ATOL)


def test_full_size_c4_projection(kf):
    """BASELINE configs[3] size (B=4096, 32 series / 4 factors, 30 % missing, T=2000).  Three full state
    records per step would be 264 GB; the projection path keeps one tape (84 GB) and emits what
    get_simulation consumes.  Checks: 72 models across the batch and across wavefront positions equal the oracle
    (loglik, projected means/variances), results are bit-for-bit independent of the batch position,
    every model's status is clean, variances are non-negative and, where a series is observed, the
    smoothed projection reproduces the observation (R = 0) with zero variance."""
    import torch

    from metran_amd.params import observation_matrix
    from metran_amd.synthetic import make_dfm_batch_torch

    B, N, K, T = 4096, 32, 4, 2000
    free, _ = torch.cuda.mem_get_info(kf.device)
    if free < 120e9:
        # on the hardware this library is written for the test must RUN: a silent skip would read as green
        total = torch.cuda.get_device_properties(kf.device).total_memory
        assert total < 250e9, "an MI355X (%.0f GB) with only %.0f GB free: something holds HBM; this test needs ~100 GB" % (total / 1e9, free / 1e9)
        pytest.skip("needs ~100 GB of free HBM (device has %.0f GB)" % (total / 1e9))
    d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=kf.device, missing=0.3)
    kf.set_observations(d["obs"]).set_loadings(d["loadings"]).set_scaling(None, None)
    r = kf.simulate_smoothed(d["phi"], d["q"])
    torch.cuda.synchronize()
    assert not r["status"].any()
    # 72 models spread over the batch and over the positions inside a wavefront (the split filter serves two models per
    # wavefront): both models of the first and last wavefronts, then pairs (even, odd) through the rest
    idx = sorted(set([0, 1, 2, 3, B - 4, B - 3, B - 2, B - 1] + [i + (k & 1) for k, i in enumerate(range(4, B - 4, 64))]))
    assert len(idx) >= 64
    sub = {k: _np(d[k][idx]) for k in ("obs", "phi", "q", "loadings")}
    ref = oracle.dfm_batch(sub["obs"], sub["phi"], sub["q"], sub["loadings"])
    assert rel_err(_np(r["mle"][idx]), ref["mle"]) < MLE_RTOL
    for i, b in enumerate(idx):
        sm, sv = oracle.simulate(observation_matrix(sub["loadings"][i]), ref["S"][i], ref["Ps"][i])
        np.testing.assert_allclose(_np(r["sim_means"][b]), sm, rtol=0, atol=1e-9)
        np.testing.assert_allclose(_np(r["sim_vars"][b]), sv, rtol=0, atol=1e-9)
    # the same models as a small, differently aligned batch: bit for bit (no result depends on the position in the batch
    # or in a wavefront -- model 1 moves from the second to the first slot of its wavefront, ...)
    sel = idx[1:34]
    kf.set_observations(d["obs"][sel].contiguous()).set_loadings(d["loadings"][sel].contiguous())
    r2 = kf.simulate_smoothed(d["phi"][sel].contiguous(), d["q"][sel].contiguous())
    assert torch.equal(r2["mle"], r["mle"][sel])
    assert torch.equal(r2["sim_means"], r["sim_means"][sel]) and torch.equal(r2["sim_vars"], r["sim_vars"][sel])
    del r2
    assert bool((r["sim_vars"] >= 0).all())
    seen = torch.isfinite(d["obs"])
    assert float((r["sim_means"] - torch.nan_to_num(d["obs"]))[seen].abs().max()) < 1e-8
    assert float(r["sim_vars"][seen].abs().max()) < 1e-8
    del r, d
    torch.cuda.empty_cache()


@pytest.mark.parametrize("N,K", [(7, 2), (11, 3), (14, 2), (20, 2), (16, 2), (32, 1), (48, 3)])
# n = 9; 14 and 16 (tiled sweeps); 22 (split filter); the last three: the smoother's structured products (series tiles + factor
# border) with 1, 2 and 3 tile rows and fewer than four factors
def test_runtime_specialised_shapes(kf, N, K, tmp_path_factory, monkeypatch):
    """Shapes outside the ahead-of-time list get kernels built at run time (metran_amd/jit.py:
    hipcc + DPP hazard check + mk_register_shape_module); same parity bar."""
    import os

    monkeypatch.setenv("METRAN_HIP_CACHE", os.environ.get("METRAN_HIP_CACHE", str(tmp_path_factory.getbasetemp() / "mkjit")))
    B, T = 9, 60
    d = make_dfm_batch(B, N, K, T, seed=300 + N, missing=0.2, first_step="random")
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    assert (N, K) in kf.supported_shapes()
    r = kf.filter_smooth(d["phi"], d["q"])
    assert rel_err(_np(r["mle"]), ref["mle"]) < MLE_RTOL
    for k, tol in (("F", FILT_ATOL), ("Pf", FILT_ATOL), ("Xp", FILT_ATOL), ("Pp", FILT_ATOL), ("S", SMOOTH_ATOL),
                   ("Ps", SMOOTH_ATOL), ("sigmas", 1e-10), ("detfs", 1e-10)):
        np.testing.assert_allclose(_np(r[k]), ref[k], rtol=0, atol=tol, err_msg=k)
    if N + K > 16:
        # wide shapes, (48,3) included (round 5: more than 32 series): the projection and the state moments over the tape
        assert kf.tape_path() and kf.state_tape_path()
        p = kf.simulate_smoothed(d["phi"], d["q"])
        assert p.get("_tape")
        Z = np.concatenate([np.broadcast_to(np.eye(N), (B, N, N)), d["loadings"]], axis=2)
        np.testing.assert_allclose(_np(p["sim_means"]), np.einsum("bjn,btn->btj", Z, ref["S"]), rtol=0, atol=SMOOTH_ATOL)
        np.testing.assert_allclose(_np(p["sim_vars"]), np.maximum(np.einsum("bjn,btnm,bjm->btj", Z, ref["Ps"], Z), 0.0), rtol=0, atol=SMOOTH_ATOL)
        v = kf.smooth_state_variances(d["phi"], d["q"])
        assert v.get("_tape")
        np.testing.assert_allclose(_np(v["S"]), ref["S"], rtol=0, atol=SMOOTH_ATOL)
        np.testing.assert_allclose(_np(v["var"]), np.diagonal(ref["Ps"], axis1=2, axis2=3), rtol=0, atol=SMOOTH_ATOL)


def test_fused_projection_epilogue(kf, g1):
    """Row f2: the smoother's fused simulate() epilogue (no smoothed states materialised) equals the
    reference's get_simulated_means/variances on examples/data and mk_simulate on the full states."""
    kf.set_observations(g1["obs"][None]).set_loadings(g1["loadings"][None])
    kf.set_scaling(g1["oseries_std"], g1["oseries_mean"])
    r = kf.simulate_smoothed(g1["phi"][None], g1["q"][None])
    assert abs(float(_np(r["mle"])[0]) - 2332.327069381027) <= MLE_RTOL * 2332.0
    np.testing.assert_allclose(_np(r["sim_means"])[0], g1["sim_means"] + g1["oseries_mean"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(_np(r["sim_vars"])[0], g1["sim_vars"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(_np(r["F"])[0], g1["F"], rtol=0, atol=FILT_ATOL)
    kf.set_scaling(None, None)
    # batched, missing data, vs oracle states projected by the oracle
    from metran_amd.params import observation_matrix

    for (N, K, B, T) in [(8, 2, 21, 80), (32, 4, 3, 30)]:
        d = make_dfm_batch(B, N, K, T, seed=77 + N, missing=0.25, first_step="random")
        rng = np.random.default_rng(N)
        scale, offset = rng.uniform(0.5, 3.0, (B, N)), rng.normal(size=(B, N))
        kf.set_observations(d["obs"]).set_loadings(d["loadings"]).set_scaling(scale, offset)
        r = kf.simulate_smoothed(d["phi"], d["q"])
        ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
        for b in range(B):
            Zs = observation_matrix(d["loadings"][b]) * scale[b][:, None]
            sm, sv = oracle.simulate(Zs, ref["S"][b], ref["Ps"][b])
            np.testing.assert_allclose(_np(r["sim_means"])[b], sm + offset[b], rtol=0, atol=1e-9)
            np.testing.assert_allclose(_np(r["sim_vars"])[b], sv, rtol=0, atol=1e-9)
        kf.set_scaling(None, None)
