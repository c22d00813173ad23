"""CPU tier: the sweep of tests/hard_models.py (the inputs of the GPU property tests) through the oracle and the numpy
restatements -- every tolerance the GPU tier applies is first shown to hold between two independent CPU implementations of
the same formulas, and the generator is shown to produce the input classes it promises."""
import numpy as np

import adjoint_ref
import dk_ref
import hard_models
import oracle


def test_generator_covers_the_promised_input_classes():
    seen, nmodels, extremes, withR, withP0 = set(), 0, 0, 0, 0
    for (N, K, T, B), g in hard_models.groups():
        nmodels += B
        seen |= set(g["patterns"])
        extremes += int((g["phi"].max(1) > 1.0 - 1e-6).sum())
        withR += B if g["obsvar"] is not None else 0
        withP0 += B if g["P0"] is not None else 0
        for b in range(B):
            y = g["obs"][b]
            if g["patterns"][b] in ("first", "both"):
                assert np.isnan(y[0]).all()
            if g["patterns"][b] in ("last", "both"):
                assert np.isnan(y[-1]).all()
            assert (g["q"][b] > 0).all() and (g["phi"][b] < 1).all()
    assert nmodels >= 300 and seen == set(hard_models.PATTERNS)
    assert extremes >= 30 and withR >= 40 and withP0 >= 40


def test_sweep_restatements_equal_the_oracle():
    """(1) the inverse-free recursion (projection and state outputs) against the oracle's smoother on every R = 0,
    default-initial-moments model of the wide shapes; (2) the adjoint restatement's objective against the oracle's on a
    sample of all models (observation variances, initial moments, empty first steps: the warm-up quirk)."""
    checked_dk = checked_mle = 0
    for (N, K, T, B), g in hard_models.groups():
        for b in range(B):
            if b % 4 == 0:
                ref = hard_models.oracle_model(oracle, g, b, smooth=False)
                R = None if g["obsvar"] is None else g["obsvar"][b]
                mle = adjoint_ref.forward(g["obs"][b], g["phi"][b], g["q"][b], g["loadings"][b], 1,
                                          None if g["x0"] is None else g["x0"][b], None if g["P0"] is None else g["P0"][b], R)[0]
                assert abs(mle - ref["mle"]) <= 1e-9 * max(1.0, abs(ref["mle"]))
                checked_mle += 1
            if N + K > 16 and g["obsvar"] is None and g["P0"] is None and b % 2 == 0:
                ref = hard_models.oracle_model(oracle, g, b)
                tol = hard_models.smoother_tolerance(g, b, ref)
                tape = dk_ref.filter_tape(g["obs"][b], g["phi"][b], g["q"][b], g["loadings"][b], state=True)
                S, var, m, v = dk_ref.dk_smooth_state(tape, g["phi"][b], g["loadings"][b])
                np.testing.assert_allclose(S, ref["S"], atol=tol)
                np.testing.assert_allclose(var, np.diagonal(ref["Ps"], axis1=1, axis2=2), atol=tol)
                np.testing.assert_allclose(m, ref["S"] @ ref["Z"].T, atol=tol)
                np.testing.assert_allclose(v, np.einsum("jn,tnm,jm->tj", ref["Z"], ref["Ps"], ref["Z"]), atol=tol)
                checked_dk += 1
    assert checked_dk >= 10 and checked_mle >= 75


def test_reference_algorithm_conditioning():
    """Where the flat bars (1e-9 relative on the per-step sigmas, 1e-10 on the filtered moments) cannot hold for ANY fp64
    implementation: on the sweep's extreme models (persistence up to 1 - 1e-9, i.e. q down to ~2e-9) the oracle -- the
    reference's algorithm in fp64 -- is itself up to ~1e-8 away from an extended-precision run of the same recursion, inside
    hard_models.conditioning's bound 2 eps scale / min(q) and (on the worst models) outside the flat bars.  The GPU tier adds
    that bound to its tolerances (tests/test_gpu_property.py); nine sweeps: the default seed, the three the bound was
    found with, seed 23, where -2 log L itself (a sum of sigmas of ~1e7-1e8: noisy data on a model with q ~ 1e-9) is
    1.9e-9 from the extended-precision sum -- the 1e-9 bar on the objective carries the same bound (hard_models.mle_tolerance)
    -- and seeds 62, 68, 88, 122 (round 6: sweeps 41 .. 70 on the GPU left both kernel families 1.1e-6 / 1.5e-7 from the
    oracle on ONE sigma each), where a sigma that is tiny beside the model's largest (5e-6 beside 8e6) is up to 29 times the
    bound from the truth RELATIVE TO ITSELF in the oracle's own arithmetic and inside a quarter of it on the scale of the
    largest: the conditioning term of the sigmas is absolute on that scale (hard_models.filter_tolerances), which is what
    their only consumer, the sum, sees."""
    worst_sig = worst_mom = worst_obj = worst_rel = 0.0
    checked = 0
    for seed in (None, 7, 11, 2024, 23, 62, 68, 88, 122):
        for (N, K, T, B), g in hard_models.groups(seed=seed):
            if N + K > 8:
                continue            # (plain Python loops in extended precision: the small shapes hold the same extremes)
            for b in range(B):
                if g["q"][b].min() > 1e-6:
                    continue
                ref = hard_models.oracle_model(oracle, g, b, smooth=False)
                sc = ref["sigmacount"]
                if sc == 0:
                    continue
                sig, F = hard_models.extended_precision_filter(g, b)
                bound = hard_models.conditioning(g, b, ref)
                truth = sig.astype(float)
                diff = np.abs(ref["sigmas"][:sc] - truth)
                atol_sig, atol_mom = hard_models.filter_tolerances(g, b, ref)
                assert (diff <= 1e-9 * np.abs(truth) + atol_sig).all(), (seed, N, K, T, b)
                top = max(1.0, float(np.abs(truth).max()))
                rel = float(diff.max()) / top
                mom = float(np.abs(ref["F"] - F.astype(float)).max())
                assert rel <= 1e-12 + 0.5 * bound and mom <= 1e-12 + bound, (seed, N, K, T, b, rel, mom, bound)
                obj = abs(float(ref["sigmas"][:sc].sum() - float(sig.sum()))) / max(1.0, abs(float(sig.sum())))
                assert obj <= 1e-12 + bound, (seed, N, K, T, b, obj, bound)
                worst_rel = max(worst_rel, float(np.max(diff / np.maximum(np.abs(truth), 1e-300))) / bound)
                worst_sig, worst_mom, worst_obj = max(worst_sig, rel), max(worst_mom, mom), max(worst_obj, obj)
                checked += 1
    assert checked >= 40
    assert worst_sig > 1e-9 and worst_mom > 1e-10 and worst_obj > 1e-9     # the flat bars are not attainable on these models
    assert worst_rel > 10.0          # ... and neither is "relative to the sigma itself + the bound": the oracle misses it by > 10 x
