"""CPU oracle for the Metran Kalman hot path -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``oracle/kalman_oracle.c`` (a plain-C restatement of
/root/reference/metran/kalmanfilter.py:236-400, 403-476, 550-567, 569-674; each C
function cites the reference lines it follows).  Parity status: PINNED against
fixtures generated from the reference itself (tests/golden/make_golden.py,
tests/test_oracle_golden.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package; ``metran_amd`` never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = ctypes.POINTER(ctypes.c_double)
c_ip = ctypes.POINTER(ctypes.c_int64)
i64 = ctypes.c_int64


def usable_cpus():
    """CPUs this process may use: its affinity mask, capped by the cgroup CPU quota (a container on a 256-thread host with a
    quota of 16 runs 128 OpenMP threads slower than 16: profiles/r05/cpu_leg_threads.log)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                n = min(n, max(1, int(round(float(q) / float(p)))))
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(round(q / p))))
    except (OSError, ValueError):
        pass
    return n


def build(native=False):
    subprocess.check_call(["make", "-C", _HERE] + (["native"] if native else []),
                          stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, "libkalman_oracle_native.so" if native else "libkalman_oracle.so")


def load(native=False):
    global _LIB
    if _LIB is not None and not native:
        return _LIB
    path = os.path.join(_HERE, "libkalman_oracle_native.so" if native else "libkalman_oracle.so")
    if not os.path.exists(path):
        build(native)
    lib = ctypes.CDLL(path)
    lib.oracle_get_mle.restype = ctypes.c_double
    lib.oracle_num_threads.restype = ctypes.c_int
    if lib.oracle_num_threads() > usable_cpus():   # (the checker's OpenMP loops: no more threads than CPUs it may use)
        lib.oracle_set_num_threads(ctypes.c_int(usable_cpus()))
    if not native:
        _LIB = lib
    return lib


def _d(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(c_ip)


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


def set_observations(oseries):
    """metran/kalmanfilter.py:646-674 -> (observations, observation_indices, observation_count)."""
    y = _c(oseries)
    T, N = y.shape
    o, oi, oc = np.empty((T, N)), np.empty((T, N)), np.empty(T, dtype=np.int64)
    load().oracle_set_observations(i64(T), i64(N), _d(y), _d(o), _d(oi), _i(oc))
    return o, oi, oc


def seqkalmanfilter(observations, transition_matrix, transition_covariance, observation_matrix,
                    observation_variance, observation_indices, observation_count,
                    filtered_state_mean, filtered_state_covariance):
    """Same 9 arguments -> same 7-tuple as metran/kalmanfilter.py:243-400."""
    o = _c(observations)
    T, N = o.shape
    x0 = _c(filtered_state_mean)
    n = x0.shape[0]
    Phi, Q, Z = _c(transition_matrix), _c(transition_covariance), _c(observation_matrix)
    R, oi = _c(observation_variance), _c(observation_indices)
    oc = _c(observation_count, np.int64)
    P0 = _c(filtered_state_covariance)
    sg, df = np.empty(T), np.empty(T)
    F, Pf, Xp, Pp = np.empty((T, n)), np.empty((T, n, n)), np.empty((T, n)), np.empty((T, n, n))
    sc = ctypes.c_int64(0)
    load().oracle_seqkalmanfilter(i64(T), i64(N), i64(n), _d(o), _d(Phi), _d(Q), _d(Z), _d(R), _d(oi),
                                  _i(oc), _d(x0), _d(P0), _d(sg), _d(df), ctypes.byref(sc),
                                  _d(F), _d(Pf), _d(Xp), _d(Pp))
    return sg, df, int(sc.value), F, Pf, Xp, Pp


def get_mle(sigmas, detfs, observation_count, warmup=1):
    """metran/kalmanfilter.py:550-567 (sigmas/detfs already sliced to sigmacount)."""
    sg, df, oc = _c(sigmas), _c(detfs), _c(observation_count, np.int64)
    return float(load().oracle_get_mle(i64(oc.shape[0]), i64(sg.shape[0]), _d(sg), _d(df), _i(oc),
                                       i64(warmup)))


def kalmansmoother(filtered_state_means, filtered_state_covariances, predicted_state_means,
                   predicted_state_covariances, transition_matrix):
    """metran/kalmanfilter.py:403-476 -> (smoothed_state_means, smoothed_state_covariances)."""
    F, Pf = _c(filtered_state_means), _c(filtered_state_covariances)
    Xp, Pp, Phi = _c(predicted_state_means), _c(predicted_state_covariances), _c(transition_matrix)
    T, n = F.shape
    S, Ps = np.empty((T, n)), np.empty((T, n, n))
    load().oracle_kalmansmoother(i64(T), i64(n), _d(F), _d(Pf), _d(Xp), _d(Pp), _d(Phi), _d(S), _d(Ps))
    return S, Ps


def simulate(observation_matrix, means, covariances):
    """metran/kalmanfilter.py:569-603 -> (simulated_means [T,N], simulated_variances [T,N])."""
    Z, m, P = _c(observation_matrix), _c(means), _c(covariances)
    N, n = Z.shape
    T = m.shape[0]
    sm, sv = np.empty((T, N)), np.empty((T, N))
    load().oracle_simulate(i64(T), i64(N), i64(n), _d(Z), _d(m), _d(P), _d(sm), _d(sv))
    return sm, sv


def decompose(observation_matrix, means):
    """metran/kalmanfilter.py:605-644 -> (sdf_means [T,N], cdf_means [K,T,N])."""
    Z, m = _c(observation_matrix), _c(means)
    N, n = Z.shape
    T = m.shape[0]
    sdf, cdf = np.empty((T, N)), np.empty((n - N, T, N))
    load().oracle_decompose(i64(T), i64(N), i64(n), _d(Z), _d(m), _d(sdf), _d(cdf))
    return sdf, cdf


def dfm_batch(obs, phi, q, loadings, obsvar=None, warmup=1, smooth=True, outputs="all", native=False, out=None):
    """B independent Metran DFMs through the reference algorithm (see C ``oracle_dfm_batch``).

    outputs: "all" | "means" | "mle".  Returns a dict.  ``out``: the dict of an earlier call of the same size, written again
    (bench.py's CPU leg times a pass that does not pay the page faults of fresh output memory).
    """
    obs, phi, q, loadings = _c(obs), _c(phi), _c(q), _c(loadings)
    B, T, N = obs.shape
    K = loadings.shape[2]
    n = N + K
    shapes = dict(mle=(B,), sigmas=(B, T), detfs=(B, T), sigmacount=(B,))
    big = outputs == "all"
    mid = outputs in ("all", "means")
    if mid:
        shapes.update(F=(B, T, n), Xp=(B, T, n))
        if smooth:
            shapes["S"] = (B, T, n)
    if big:
        shapes.update(Pf=(B, T, n, n), Pp=(B, T, n, n))
        if smooth:
            shapes["Ps"] = (B, T, n, n)
    if out is not None:
        if any(k not in out or out[k].shape != sh or not out[k].flags.c_contiguous for k, sh in shapes.items()):
            raise ValueError("out does not hold the arrays of this call")
        res = {k: out[k] for k in shapes}
    else:
        res = {k: np.empty(sh, dtype=np.int64 if k == "sigmacount" else np.float64) for k, sh in shapes.items()}
    ov = None if obsvar is None else _c(obsvar)
    load(native).oracle_dfm_batch(
        i64(B), i64(T), i64(N), i64(K), _d(obs), _d(phi), _d(q), _d(loadings), _d(ov), i64(warmup),
        ctypes.c_int(1 if smooth else 0), _d(res["mle"]), _d(res["sigmas"]), _d(res["detfs"]),
        _i(res["sigmacount"]), _d(res.get("F")), _d(res.get("Pf")), _d(res.get("Xp")),
        _d(res.get("Pp")), _d(res.get("S")), _d(res.get("Ps")))
    return res


def num_threads(native=False):
    return int(load(native).oracle_num_threads())


def set_num_threads(t, native=False):
    load(native).oracle_set_num_threads(ctypes.c_int(int(t)))


# ---------------------------------------------------------------------------------------------------------------
# The OPTIMISED CPU leg (oracle/kalman_fast.c): same recursions written for speed (diagonal Phi / Q and Z = [I | G]
# exploited, Cholesky instead of pinv, -ffp-contract=fast); bench.py's ``cpu_baseline_optimised``.  Not the checker:
# tests/test_oracle_golden.py checks IT against the checker.
_FAST = {}


def load_fast(native=False):
    if native not in _FAST:
        path = os.path.join(_HERE, "libkalman_fast_native.so" if native else "libkalman_fast.so")
        if not os.path.exists(path):
            build(native)
        lib = ctypes.CDLL(path)
        lib.fast_dfm_batch.restype = ctypes.c_int64
        lib.fast_num_threads.restype = ctypes.c_int
        if lib.fast_num_threads() > usable_cpus():
            lib.fast_set_num_threads(ctypes.c_int(usable_cpus()))
        _FAST[native] = lib
    return _FAST[native]


def fast_dfm_batch(obs, phi, q, loadings, warmup=1, outputs="all", native=False, out=None):
    """B models through oracle/kalman_fast.c.  outputs: "all" (six state arrays) | "means" (projected smoothed means /
    variances) | "mle".  Returns a dict (with ``bad`` = number of models the fast path cannot serve: a non-positive
    innovation variance or a predicted covariance that is not positive definite).  ``out``: the dict of an earlier call of
    the same size, whose arrays are written again (a timing that should not include the page faults of fresh output memory)."""
    obs, phi, q, loadings = _c(obs), _c(phi), _c(q), _c(loadings)
    B, T, N = obs.shape
    K = loadings.shape[2]
    n = N + K
    mode = {"mle": 0, "means": 1, "all": 2}[outputs]
    shapes = dict(mle=(B,))
    if mode == 2:
        shapes.update(F=(B, T, n), Xp=(B, T, n), S=(B, T, n), Pf=(B, T, n, n), Pp=(B, T, n, n), Ps=(B, T, n, n))
    if mode == 1:
        shapes.update(sim_means=(B, T, N), sim_vars=(B, T, N))
    if out is not None:
        if any(k not in out or out[k].shape != sh or not out[k].flags.c_contiguous for k, sh in shapes.items()):
            raise ValueError("out does not hold the arrays of this call")
        res = {k: out[k] for k in shapes}
    else:
        res = {k: np.empty(sh) for k, sh in shapes.items()}
    res["bad"] = int(load_fast(native).fast_dfm_batch(
        i64(B), i64(T), i64(N), i64(K), _d(obs), _d(phi), _d(q), _d(loadings), i64(warmup), ctypes.c_int(mode), _d(res["mle"]),
        _d(res.get("F")), _d(res.get("Pf")), _d(res.get("Xp")), _d(res.get("Pp")), _d(res.get("S")), _d(res.get("Ps")),
        _d(res.get("sim_means")), _d(res.get("sim_vars"))))
    return res


def fast_num_threads(native=False):
    return int(load_fast(native).fast_num_threads())


def fast_set_num_threads(t, native=False):
    load_fast(native).fast_set_num_threads(ctypes.c_int(int(t)))
