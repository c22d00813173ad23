"""GPU: the size-generic kernels (metran_amd/csrc/mk_generic.hip) -- any model shape up to 128 states without specialisation
(missing 1: the reference's loops are size-generic, kalmanfilter.py:315-390, 453-474; the library stopped at
N + K <= 64 and needed hipcc at run time for every shape outside its ahead-of-time list).  Same parity bar as the specialised
kernels: -2 log L 1e-9 relative, filtered / predicted moments 1e-10, smoothed moments 1e-9, against the oracle."""
import numpy as np
import pytest

import oracle
from metran_amd.synthetic import make_dfm_batch

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture
def no_jit(monkeypatch, tmp_path):
    """A machine without hipcc: shapes outside the ahead-of-time list fall back to the generic kernels."""
    monkeypatch.setenv("METRAN_HIP_JIT", "0")
    monkeypatch.setenv("METRAN_HIP_CACHE", str(tmp_path))
    # ... and without the modules build() prebuilds next to the library (round 6: the grid N = 2..16 x K = 1..3 ships there)
    from metran_amd import jit

    monkeypatch.setattr(jit, "PREBUILT_DIR", str(tmp_path / "no_prebuilt_modules"))


def _check_all(kf, d, ref, obsvar=None, x0=None, P0=None):
    B, T, N = d["obs"].shape
    r = kf.filter_smooth(d["phi"], d["q"], x0=x0, P0=P0)             # full output set: packed records
    assert r.get("_rs")
    np.testing.assert_allclose(_np(r["mle"]), ref["mle"], rtol=1e-9)
    assert np.array_equal(_np(r["sigmacount"]), ref["sigmacount"])
    for k, tol in (("F", 1e-10), ("Pf", 1e-10), ("Xp", 1e-10), ("Pp", 1e-10), ("S", 1e-9), ("Ps", 1e-9), ("sigmas", 1e-10), ("detfs", 1e-10)):
        np.testing.assert_allclose(_np(r[k]), ref[k], rtol=0, atol=tol, err_msg=k)
    assert int(_np(r["status"]).sum()) == 0
    r2 = kf.filter_smooth(d["phi"], d["q"], x0=x0, P0=P0, outputs=("F", "Pf", "S", "Ps"))   # a subset: dense arrays
    assert not r2.get("_rs")
    for k, tol in (("F", 1e-10), ("Pf", 1e-10), ("S", 1e-9), ("Ps", 1e-9)):
        np.testing.assert_allclose(_np(r2[k]), ref[k], rtol=0, atol=tol, err_msg="dense " + k)
    np.testing.assert_allclose(_np(kf.loglik(d["phi"], d["q"], x0=x0, P0=P0)), ref["mle"], rtol=1e-9)
    rng = np.random.default_rng(5)
    scale, offset = rng.uniform(0.5, 2.0, (B, N)), rng.normal(size=(B, N))
    kf.set_scaling(scale, offset)
    p = kf.simulate_smoothed(d["phi"], d["q"], x0=x0, P0=P0)
    assert not p.get("_tape")
    Z = np.concatenate([np.broadcast_to(np.eye(N), (B, N, N)), d["loadings"]], axis=2) * scale[:, :, None]
    np.testing.assert_allclose(_np(p["sim_means"]), np.einsum("bjn,btn->btj", Z, ref["S"]) + offset[:, None, :], atol=1e-9)
    np.testing.assert_allclose(_np(p["sim_vars"]), np.maximum(np.einsum("bjn,btnm,bjm->btj", Z, ref["Ps"], Z), 0.0), atol=1e-9)
    s = kf.smooth_state_variances(d["phi"], d["q"], x0=x0, P0=P0)
    np.testing.assert_allclose(_np(s["S"]), ref["S"], atol=1e-9)
    np.testing.assert_allclose(_np(s["var"]), np.diagonal(ref["Ps"], axis1=2, axis2=3), atol=1e-9)


@pytest.mark.parametrize("layout", ["model_major", "time_major"])
@pytest.mark.parametrize("N,K,T,B,missing", [(70, 3, 30, 3, 0.3), (100, 28, 12, 2, 0.5), (61, 4, 25, 2, 0.0), (64, 64, 8, 2, 0.3), (92, 3, 10, 5, 0.2)])
def test_models_beyond_64_states(N, K, T, B, missing, layout):
    """n = 73, 128 (the limit) and 65: refused before round 5.  Round 6 (the family's second form): (64, 64) is the limit with a
    loadings table too large for LDS next to the covariance (the kernels' GL = false instantiation), (92, 3) the band where the
    smoother keeps ONE of its work matrices in LDS (LW = 1; 73 and 65 states: two, LW = 2; 128: one); the small shapes below and
    the property sweep's generic pass cover the one- and four-wavefront forms with every matrix in LDS."""
    from metran_amd.engine import BatchedKalman

    d = make_dfm_batch(B, N, K, T, seed=900 + N, missing=missing, first_step="random")
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    kf = BatchedKalman(layout=layout)
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    assert not kf.specialised() and not kf.has_adjoint() and not kf.tape_path()
    _check_all(kf, d, ref)
    kf.close()


def test_limit_and_refusals():
    from metran_amd.engine import BatchedKalman, MetranHipError

    d = make_dfm_batch(1, 126, 3, 4, seed=1)
    kf = BatchedKalman()
    kf.set_observations(d["obs"])
    with pytest.raises(MetranHipError, match="N \\+ K <= 128"):
        kf.set_loadings(d["loadings"])
    kf.close()
    d = make_dfm_batch(2, 70, 3, 6, seed=2)
    kf = BatchedKalman(packed_sym=True)
    kf.set_observations(d["obs"])
    with pytest.raises(MetranHipError, match="packed-symmetric"):
        kf.set_loadings(d["loadings"])
    kf.close()
    kf = BatchedKalman()
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    with pytest.raises(MetranHipError, match="adjoint"):
        kf.loglik_grad(d["phi"], d["q"])
    kf.close()


@pytest.mark.parametrize("N,K", [(9, 1), (13, 2), (21, 3)])
def test_without_a_compiler_small_shapes_run_generic(N, K, no_jit):
    """A shape outside the ahead-of-time list on a machine that cannot build a module: served, not refused -- with observation
    variances, initial moments, an empty first step and a never-observed series."""
    from metran_amd.engine import BatchedKalman
    from metran_amd.params import observation_matrix

    B, T = 5, 40
    d = make_dfm_batch(B, N, K, T, seed=700 + N, missing=0.35, first_step="empty")
    d["obs"][:, :, 1] = np.nan
    rng = np.random.default_rng(N)
    n = N + K
    R = rng.uniform(0.0, 0.3, (B, N)) * (rng.random((B, N)) < 0.5)
    x0 = rng.normal(size=(B, n))
    A = rng.normal(size=(B, n, n))
    P0 = A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n)
    kf = BatchedKalman()
    kf.set_observations(d["obs"]).set_loadings(d["loadings"], R)
    assert not kf.specialised()
    ref = {k: [] for k in ("mle", "sigmacount", "F", "Pf", "Xp", "Pp", "S", "Ps", "sigmas", "detfs")}
    for b in range(B):
        o, oi, oc = oracle.set_observations(d["obs"][b])
        sg, df, sc, F, Pf, Xp, Pp = oracle.seqkalmanfilter(o, np.diag(d["phi"][b]), np.diag(d["q"][b]), observation_matrix(d["loadings"][b]),
                                                           R[b], oi, oc, x0[b], P0[b])
        S, Ps = oracle.kalmansmoother(F, Pf, Xp, Pp, np.diag(d["phi"][b]))
        sg[sc:] = 0.0
        df[sc:] = 0.0
        for k, v in (("mle", oracle.get_mle(sg[:sc], df[:sc], oc)), ("sigmacount", sc), ("F", F), ("Pf", Pf), ("Xp", Xp), ("Pp", Pp),
                     ("S", S), ("Ps", Ps), ("sigmas", sg), ("detfs", df)):
            ref[k].append(v)
    ref = {k: np.array(v) for k, v in ref.items()}
    _check_all(kf, d, ref, obsvar=R, x0=x0, P0=P0)
    kf.close()


def test_generic_equals_specialised_on_an_aot_shape():
    """The two kernel families on the same models: the generic filter / smoother through the literal 5-argument smoother entry
    (mk_smooth_dense) against the specialised (8,2) kernels -- and a predicted covariance that is NOT the filter's (perturbed):
    kalmansmoother's answer for the arrays as given (kalmanfilter.py:453-474), which the fast path cannot produce."""
    from metran_amd.engine import BatchedKalman

    B, N, K, T = 6, 8, 2, 50
    d = make_dfm_batch(B, N, K, T, seed=4, missing=0.2)
    kf = BatchedKalman()
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    assert kf.specialised()
    r = kf.filter_smooth(d["phi"], d["q"])
    g = kf.smooth_dense(d["phi"], r["F"].contiguous(), r["Pf"].contiguous(), r["Xp"].contiguous(), r["Pp"].contiguous())
    np.testing.assert_allclose(_np(g["S"]), _np(r["S"]), atol=1e-10)
    np.testing.assert_allclose(_np(g["Ps"]), _np(r["Ps"]), atol=1e-10)
    F, Pf, Xp, Pp = (_np(r[k]).copy() for k in ("F", "Pf", "Xp", "Pp"))
    Pp2 = Pp * 1.05 + 0.01 * np.eye(N + K)
    Xp2 = Xp + 0.1
    g2 = kf.smooth_dense(d["phi"], F, Pf, Xp2, Pp2)
    for b in range(B):
        S, Ps = oracle.kalmansmoother(F[b], Pf[b], Xp2[b], Pp2[b], np.diag(d["phi"][b]))
        np.testing.assert_allclose(_np(g2["S"])[b], S, atol=1e-9)
        np.testing.assert_allclose(_np(g2["Ps"])[b], Ps, atol=1e-9)
    assert np.abs(_np(g2["S"]) - _np(r["S"])).max() > 1e-3
    kf.close()


def test_smoother_adapter_honours_its_five_arguments():
    """kalmansmoother_hip on the GPU: the filter call's own arrays take the device-resident fast path; a perturbed Pp, a
    shifted Xp or an in-place edit of a returned array give the reference algorithm's answer for THOSE arrays."""
    import metran_amd.kalmanfilter as hip
    from metran_amd.params import observation_matrix
    from metran_amd.synthetic import make_dfm

    y, _, load, phi, q = make_dfm(5, 1, 80, 31, 0, 0.3, "random")
    o, oi, oc = oracle.set_observations(y)
    Phi, Q, Z = np.diag(phi), np.diag(q), observation_matrix(load)
    sg, df, sc, F, Pf, Xp, Pp = hip.seqkalmanfilter_hip(o, Phi, Q, Z, np.zeros(5), oi, oc, np.zeros(6), np.eye(6))
    S, Ps = hip.kalmansmoother_hip(F, Pf, Xp, Pp, Phi)
    S0, Ps0 = oracle.kalmansmoother(F, Pf, Xp, Pp, Phi)
    np.testing.assert_allclose(S, S0, atol=1e-9)
    np.testing.assert_allclose(Ps, Ps0, atol=1e-9)
    Pp2, Xp2 = Pp * 1.05 + 0.01 * np.eye(6), Xp + 0.1
    S2, Ps2 = hip.kalmansmoother_hip(F, Pf, Xp2, Pp2, Phi)
    Sr, Psr = oracle.kalmansmoother(F, Pf, Xp2, Pp2, Phi)
    np.testing.assert_allclose(S2, Sr, atol=1e-9)
    np.testing.assert_allclose(Ps2, Psr, atol=1e-9)
    assert np.abs(S2 - S0).max() > 1e-3
    Pp[7] *= 1.02
    S3, Ps3 = hip.kalmansmoother_hip(F, Pf, Xp, Pp, Phi)
    Sr3, Psr3 = oracle.kalmansmoother(F, Pf, Xp, Pp, Phi)
    np.testing.assert_allclose(S3, Sr3, atol=1e-9)
    np.testing.assert_allclose(Ps3, Psr3, atol=1e-9)


def test_drop_in_engine_on_a_model_of_70_series():
    """The 9-argument engine callable and the 5-argument smoother on a 73-state model (what ``install(metran)`` binds):
    ran into MK_ERR_SHAPE before round 5."""
    import metran_amd.kalmanfilter as hip
    from metran_amd.params import observation_matrix
    from metran_amd.synthetic import make_dfm

    N, K, T = 70, 3, 25
    y, _, load, phi, q = make_dfm(N, K, T, 41, 0, 0.4, "random")
    o, oi, oc = oracle.set_observations(y)
    Phi, Q, Z = np.diag(phi), np.diag(q), observation_matrix(load)
    args = (o, Phi, Q, Z, np.zeros(N), oi, oc, np.zeros(N + K), np.eye(N + K))
    got = hip.seqkalmanfilter_hip(*args)
    ref = oracle.seqkalmanfilter(*args)
    assert got[2] == ref[2]
    for a, b, tol in zip(got[3:], ref[3:], (1e-10,) * 4):
        np.testing.assert_allclose(a, b, atol=tol)
    np.testing.assert_allclose(got[0][:got[2]], ref[0][:ref[2]], rtol=1e-9, atol=1e-10)
    S, Ps = hip.kalmansmoother_hip(*got[3:], Phi)
    S0, Ps0 = oracle.kalmansmoother(*ref[3:], Phi)
    np.testing.assert_allclose(S, S0, atol=1e-9)
    np.testing.assert_allclose(Ps, Ps0, atol=1e-9)


def test_batched_calibration_of_a_model_beyond_64_states():
    """``calibrate_batch`` on a 73-state model: size-generic kernels, differenced gradients (no adjoint kernel beyond 64 states), the
    L-BFGS kernels with 73 parameters per model (their limit is the generic kernels': 128; it was 64 until the end of round 5,
    and this call was refused).  A few iterations next to the SAME driver over the oracle-backed stand-in engine: with
    differenced gradients (step 1e-8) a 1e-13 difference in an objective of ~700 is 7e-3 in a gradient component, so the two
    trajectories agree in the objective they reach to ~1e-3, not digit for digit (measured: 6.7e-4 after four iterations)."""
    import torch

    from metran_amd.calibrate import calibrate_batch
    from metran_amd.engine import BatchedKalman
    from oracle_engine import OracleEngine

    N, K, T, B = 70, 3, 24, 2
    d = make_dfm_batch(B, N, K, T, seed=77, missing=0.2)
    kf = BatchedKalman()
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    assert not kf.specialised() and not kf.has_adjoint()
    res = calibrate_batch(kf, maxiter=4, compact=0)
    ref = calibrate_batch(OracleEngine(d["obs"], d["loadings"], adjoint=False), maxiter=4, compact=0)
    assert res.nit == ref.nit == 4
    f0 = kf.loglik(*kf.params_from_alpha(torch.full((B, N + K), 10.0, dtype=torch.float64, device=kf.device)))
    assert bool((res.obj < f0).all()) and bool(torch.isfinite(res.alpha).all())
    np.testing.assert_allclose(_np(res.obj), ref.obj.numpy(), rtol=5e-3)
    assert float((f0.cpu() - res.obj.cpu()).min()) > 10.0 * float((res.obj.cpu() - ref.obj).abs().max())   # both went the same way down
    kf.close()
