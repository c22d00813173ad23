"""Seeded generator of "hard" dynamic-factor models for the property sweeps (test infrastructure; used by the GPU tier's
tests/test_gpu_property.py and, on the CPU, by tests/test_property_generator.py, which runs the SAME sweep through the oracle
and the numpy restatements so that every tolerance below is known to hold between two independent implementations before
a GPU minute is spent on it).

A group = B models of one shape (N, K) and one length T (a launch needs a common T), each model drawn independently:
  persistence   phi = exp(-1/alpha), alpha ~ U(2, 60) (Metran's parametrisation, metran.py:246-263); in 30 % of the models one
                or two states (series or factor) get phi = 1 - 10^U(-9, -3) instead -- up to 1 - 1e-9.  (ALL states that
                persistent would make every innovation variance a 1e-9 difference of O(1) covariances: the objective
                is then conditioned like 1e-16 / 1e-9 and no two implementations agree to 1e-9 on it.)
  loadings      random signs, rows scaled to a communality drawn from {0.2, 0.9, 0.999} x U(0.3, 1)
  q             Metran's: (1 - phi^2)(1 - communality) for the series, 1 - phi^2 for the factors (metran.py:310-322)
  data          simulated from the model (prior N(0, I)), for half of the typical models plus noise; then one missingness pattern:
                none | iid 30 % | heavy 95 % | whole steps | never-observed series | a single observation |
                empty first step | empty last step | empty first and last
  group-wide    observation variances R >= 0 (zero for about half of the series) or none; x0 / P0 (SPD) or the defaults
"""
import numpy as np

PATTERNS = ("none", "iid", "heavy", "steps", "series", "single", "first", "last", "both")
AOT_SHAPES = [(8, 2), (5, 1), (2, 1), (3, 1), (4, 1), (6, 2), (14, 3), (32, 4)]
JIT_SHAPES = [(7, 2), (20, 2), (48, 3)]   # all built anyway by tests/test_hip_parity.py::test_runtime_specialised_shapes; (48,3): the
# tape of a model with more than 32 series (round 5) -- appended, so that the groups drawn before it stay what they were


def draw_model(rng, N, K, T, pattern):
    n = N + K
    phi = np.exp(-1.0 / rng.uniform(2.0, 60.0, n))
    if rng.random() < 0.3:   # one or two states (series or factor) with a persistence up to 1 - 1e-9
        idx = rng.choice(n, size=min(n, int(rng.integers(1, 3))), replace=False)
        phi[idx] = 1.0 - 10.0 ** rng.uniform(-9.0, -3.0, idx.size)
    load = rng.uniform(-1.0, 1.0, (N, K))
    comm = rng.choice([0.2, 0.9, 0.999])
    load *= np.sqrt(comm / np.maximum((load ** 2).sum(1), 1e-12))[:, None] * rng.uniform(0.3, 1.0, N)[:, None]
    q = 1.0 - phi ** 2
    q[:N] *= 1.0 - (load ** 2).sum(1)
    # data from the model itself, started from its prior x_{-1} ~ N(0, I) (run_filter's P0, kalmanfilter.py:747-750): with a
    # persistence of 1 - 1e-9 the innovation variances are ~1e-9, and data the model could not have produced would put
    # terms v^2 / f ~ 1e9 into the objective -- a test of cancellation, not of the filter.  A typical-persistence model also
    # gets plain noise on top (off the model's manifold, like real residual series).
    x = rng.standard_normal(n)
    y = np.empty((T, N))
    for t in range(T):
        x = phi * x + np.sqrt(q) * rng.standard_normal(n)
        y[t] = x[:N] + load @ x[N:]
    if rng.random() < 0.5:
        y += 0.3 * rng.standard_normal((T, N))
    if pattern == "iid":
        y[rng.random((T, N)) < 0.3] = np.nan
    elif pattern == "heavy":
        y[rng.random((T, N)) < 0.95] = np.nan
    elif pattern == "steps":
        y[rng.random(T) < 0.5] = np.nan
    elif pattern == "series":
        y[:, rng.random(N) < 0.5] = np.nan
        y[rng.random((T, N)) < 0.2] = np.nan
    elif pattern == "single":
        keep = (rng.integers(T), rng.integers(N))
        v = y[keep]
        y[:] = np.nan
        y[keep] = v
    else:
        if pattern in ("first", "both"):
            y[rng.random((T, N)) < 0.2] = np.nan
            y[0] = np.nan
        if pattern in ("last", "both"):
            y[rng.random((T, N)) < 0.2] = np.nan
            y[-1] = np.nan
    return y, phi, q, load


def draw_group(seed, N, K, T, B):
    """dict: obs [B,T,N], phi / q [B,n], loadings [B,N,K], obsvar [B,N] or None, x0 [B,n] / P0 [B,n,n] or None, patterns."""
    rng = np.random.default_rng([int(seed), N, K, T, B])
    n = N + K
    g = dict(obs=np.empty((B, T, N)), phi=np.empty((B, n)), q=np.empty((B, n)), loadings=np.empty((B, N, K)), patterns=[])
    for b in range(B):
        pat = PATTERNS[(b + int(rng.integers(len(PATTERNS)))) % len(PATTERNS)] if B < len(PATTERNS) else PATTERNS[b % len(PATTERNS)]
        g["obs"][b], g["phi"][b], g["q"][b], g["loadings"][b] = draw_model(rng, N, K, T, pat)
        g["patterns"].append(pat)
    g["obsvar"] = rng.uniform(0.0, 0.5, (B, N)) * (rng.random((B, N)) < 0.5) if rng.random() < 0.4 else None
    if rng.random() < 0.4:
        g["x0"] = rng.normal(size=(B, n))
        A = rng.normal(size=(B, n, n))
        g["P0"] = A @ A.transpose(0, 2, 1) / n + 0.5 * np.eye(n)
    else:
        g["x0"] = g["P0"] = None
    return g


def groups(seed=None, per_shape=32, shapes=None):
    """The sweep: for every shape two groups, a short one (T in 1..12, 12 models) and a longer one (T in 20..56, the rest).
    The default seed is fixed (the tier is deterministic); ``METRAN_SWEEP_SEED=<int>`` draws another sweep of the same classes
    (profiles/r05/gpu_property_other_seeds.log: three more sweeps, 1 056 more models, run once on the final kernels)."""
    import os

    if seed is None:
        seed = int(os.environ.get("METRAN_SWEEP_SEED", "20250922"))
    rng = np.random.default_rng(seed)
    for (N, K) in (shapes or (AOT_SHAPES + JIT_SHAPES)):
        for (tlo, thi, B) in ((1, 13, 12), (20, 57, per_shape - 12)):
            T = int(rng.integers(tlo, thi))
            yield (N, K, T, B), draw_group(seed, N, K, T, B)


def oracle_model(oracle, g, b, smooth=True):
    """The reference algorithm on model b of a group (oracle/: C restatement of kalmanfilter.py:236-476, 550-567) with its
    observation variances and initial moments.  dict: sigmas, detfs, sigmacount, mle, F, Pf, Xp, Pp (, S, Ps)."""
    N, K = g["loadings"].shape[1:]
    n = N + K
    Z = np.concatenate([np.eye(N), g["loadings"][b]], axis=1)
    R = np.zeros(N) if g["obsvar"] is None else g["obsvar"][b]
    x0 = np.zeros(n) if g["x0"] is None else g["x0"][b]
    P0 = np.eye(n) if g["P0"] is None else g["P0"][b]
    o, oi, oc = oracle.set_observations(g["obs"][b])
    sg, df, sc, F, Pf, Xp, Pp = oracle.seqkalmanfilter(o, np.diag(g["phi"][b]), np.diag(g["q"][b]), Z, R, oi, oc, x0, P0)
    out = dict(sigmas=sg, detfs=df, sigmacount=sc, F=F, Pf=Pf, Xp=Xp, Pp=Pp, mle=oracle.get_mle(sg[:sc], df[:sc], oc), Z=Z)
    if smooth:
        out["S"], out["Ps"] = oracle.kalmansmoother(F, Pf, Xp, Pp, np.diag(g["phi"][b]))
    return out


def conditioning(g, b, ref):
    """What the REFERENCE's fp64 filter itself loses on model b: an innovation variance f is a difference of O(scale)
    covariance entries and cannot be smaller than ~min(q), so its relative rounding error is up to eps * scale / min(q) --
    and so is that of everything divided by it (the gain, sigma = v^2 / f).  With a persistence of 1 - 1e-9 (q ~ 2e-9) that is
    ~1e-7: the oracle is 7e-9 (sigmas) / 3e-9 (filtered means) from an extended-precision run of the same recursion on such
    models (tests/test_property_generator.py::test_reference_algorithm_conditioning), and two correct fp64 implementations
    differ from each other by as much.  Returned: 2 eps * scale / min(q); ~1e-15 for an ordinary model."""
    scale = max(1.0, float(np.abs(ref["Pp"]).max()), float(np.abs(ref["F"]).max()))
    return 2.0 * np.finfo(float).eps * scale / float(g["q"][b].min())


def mle_tolerance(g, b, ref, value=None):
    """Absolute tolerance on -2 log L of model b (``value``: the objective under another warm-up, default ref["mle"]): the
    north-star bar (1e-9 relative) plus the reference algorithm's own conditioning -- the objective is a sum of sigmas, so
    it inherits their eps * scale / min(q) (seed 23 of the sweep holds two models whose fp64 oracle is 0.7e-9 / 1.9e-9 from
    the extended-precision objective)."""
    value = ref["mle"] if value is None else value
    return (1e-9 + conditioning(g, b, ref)) * max(1.0, abs(float(value)))


def filter_tolerances(g, b, ref):
    """(atol of the per-step sigmas -- on top of the flat 1e-9 relative bar --, atol of the filtered / predicted moments) for
    model b: the repository's bars (1e-9 relative / 1e-10 for the sigmas, 1e-10 on the scale of the moments) plus the reference
    algorithm's own conditioning (``conditioning``).  The conditioning term of a sigma is ABSOLUTE, on the scale of the model's
    largest sigma: sigma_t = v^2 / f carries the accumulated error of the state mean through v, so a step whose innovation
    happens to be small next to the model's others (sigma_t 5e-6 beside 8e6: seed 88 of the sweep) has NO relative accuracy in
    the reference's own fp64 arithmetic -- the oracle is up to 29 eps scale / min(q) from the extended-precision run on such
    entries relative to themselves (seeds 62, 88, 122, 125: found by METRAN_SWEEP_SEED = 41 .. 70 on the GPU, where both kernel
    families sat 1.1e-6 / 1.5e-7 from the oracle on ONE sigma each, and the oracle 1.4e-6 from the truth on the first) but never
    more than 0.52 of it on the scale of the largest sigma (104 sweeps, 1 800 extreme models:
    tests/test_property_generator.py::test_reference_algorithm_conditioning).  Sigmas are consumed through their sum only."""
    scale = max(1.0, float(np.abs(ref["Pp"]).max()), float(np.abs(ref["F"]).max()))
    c = conditioning(g, b, ref)
    sc = ref["sigmacount"]
    top = max(1.0, float(np.abs(ref["sigmas"][:sc]).max())) if sc else 1.0
    return 1e-10 + c * top, 1e-10 * scale + c


def extended_precision_filter(g, b):
    """The sequential filter of model b in numpy's extended precision (80-bit on x86: eps 1e-19): per-step sigmas of the
    observed steps and the filtered means.  Plain loops; for the handful of extreme models of the sweep."""
    LD = np.longdouble
    y, G = g["obs"][b], g["loadings"][b]
    T, N = y.shape
    K = G.shape[1]
    n = N + K
    Z = np.concatenate([np.eye(N), G], axis=1).astype(LD)
    R = np.zeros(N) if g["obsvar"] is None else g["obsvar"][b]
    x = (np.zeros(n) if g["x0"] is None else g["x0"][b]).astype(LD)
    P = (np.eye(n) if g["P0"] is None else g["P0"][b]).astype(LD)
    phi, q = g["phi"][b].astype(LD), g["q"][b].astype(LD)
    sig, Fs = [], []
    for t in range(T):
        x = phi * x
        P = P * np.outer(phi, phi) + np.diag(q)
        s, seen = LD(0), False
        for j in range(N):
            if np.isfinite(y[t, j]):
                seen = True
                z = Z[j]
                v = LD(y[t, j]) - z @ x
                d = P @ z
                f = LD(R[j]) + z @ d
                k = d / f
                x = x + k * v
                P = P - np.outer(k, k) * f
                s += v * v / f
        if seen:
            sig.append(s)
        Fs.append(x.copy())
    return np.array(sig, dtype=LD), np.array(Fs, dtype=LD)


def smoother_tolerance(g, b, ref):
    """Absolute tolerance on smoothed moments of model b: the repository's smoother bar (1e-9) on the scale of the moments
    in play, plus what the REFERENCE algorithm itself loses in its explicit inverse of the predicted covariance
    (kalmanfilter.py:455-462): eps * cond(Pp) ~ 1e-15 / min(q) -- with persistence close to 1 or a communality close to 1 two
    correct implementations of the reference's formulas differ by that much (tests/test_dk_tape.py, property test)."""
    big = max(1.0, float(np.abs(ref["Pp"]).max()), float(np.abs(ref["F"]).max()))
    return 1e-9 * big + 4e-15 * big * big / float(g["q"][b].min())
