"""GPU: the output layouts of the C-ABI contract -- PACKED_SYM records
(n + n(n+1)/2 doubles per moment set instead of n + n^2) and VAR_ONLY smoothing (state means + variances) -- give
the same numbers as the full-square records / the oracle, for the 16-lane kernels (n <= 16) and the wide ones; and a
torch-free walk through the ABI with mk_malloc / mk_memcpy_* / mk_memset."""
import ctypes

import numpy as np
import pytest

import oracle
from metran_amd.synthetic import make_dfm_batch

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("layout", ["model_major", "time_major"])
# (11, 6): run-time-specialised, n = 17 on the split filter's 16-lane groups -- its factor block has K K = 36 elements
# (K (K + 1) / 2 = 21 packed) for 16 lanes: the block is stored in more than one pass (round-4 advice: one element per lane
# left part of it unwritten for K >= 5)
@pytest.mark.parametrize("N,K,T,B,missing", [(8, 2, 70, 9, 0.15), (5, 1, 40, 3, 0.3), (32, 4, 30, 3, 0.3), (14, 3, 25, 2, 0.0),
                                             (11, 6, 25, 3, 0.2)])
def test_packed_symmetric_records(layout, N, K, T, B, missing):
    from metran_amd.engine import BatchedKalman

    n = N + K
    d = make_dfm_batch(B, N, K, T, seed=808 + N, missing=missing, first_step="random")
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    kf = BatchedKalman(layout=layout, packed_sym=True)
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    assert kf.record_stride() == ((n + n * (n + 1) // 2 + 2 + 15) // 16) * 16 < ((n + n * n + 2 + 15) // 16) * 16
    r = kf.filter_smooth(d["phi"], d["q"])
    assert r["Pf"].shape == (B, T, n * (n + 1) // 2)                     # the packed triangle, not a square
    np.testing.assert_allclose(_np(r["mle"]), ref["mle"], rtol=1e-9)
    for k in ("F", "Xp", "S"):
        np.testing.assert_allclose(_np(r[k]), ref[k], atol=1e-9)
    for k in ("Pf", "Pp", "Ps"):
        full = _np(kf.unpack_sym(r[k]))
        np.testing.assert_allclose(full, ref[k], atol=1e-9)
        np.testing.assert_array_equal(full, np.swapaxes(full, -1, -2))  # exactly symmetric by construction
    sc = _np(r["sigmacount"])
    for b in range(B):
        np.testing.assert_allclose(_np(r["sigmas"])[b, : sc[b]], ref["sigmas"][b, : sc[b]], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(_np(r["detfs"])[b, : sc[b]], ref["detfs"][b, : sc[b]], atol=1e-10)
    # projection path on packed-symmetric filtered records == the projection of the oracle's smoothed moments
    p = kf.simulate_smoothed(d["phi"], d["q"])
    Z = np.concatenate([np.broadcast_to(np.eye(N), (B, N, N)), d["loadings"]], axis=2)
    np.testing.assert_allclose(_np(p["sim_means"]), np.einsum("bjn,btn->btj", Z, ref["S"]), atol=1e-9)
    np.testing.assert_allclose(_np(p["sim_vars"]), np.maximum(np.einsum("bjn,btnm,bjm->btj", Z, ref["Ps"], Z), 0), atol=1e-9)
    # full-square engine on the same data: the same values up to which of the two rounding-level different copies
    # (r,c) / (c,r) of a covariance element each layout keeps
    kq = BatchedKalman(layout=layout)
    kq.set_observations(d["obs"]).set_loadings(d["loadings"])
    rq = kq.filter_smooth(d["phi"], d["q"])
    iu = np.triu_indices(n)
    np.testing.assert_allclose(_np(r["Ps"]), _np(rq["Ps"])[..., iu[0], iu[1]], rtol=0, atol=1e-13)
    np.testing.assert_allclose(_np(r["S"]), _np(rq["S"]), rtol=0, atol=1e-13)


@pytest.mark.parametrize("packed_sym", [False, True])
@pytest.mark.parametrize("N,K,T,B", [(8, 2, 60, 7), (32, 4, 24, 2)])
def test_var_only_smoothing(N, K, T, B, packed_sym):
    from metran_amd.engine import BatchedKalman

    d = make_dfm_batch(B, N, K, T, seed=515 + N, missing=0.2)
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    kf = BatchedKalman(layout="time_major", packed_sym=packed_sym)
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    r = kf.smooth_state_variances(d["phi"], d["q"])
    np.testing.assert_allclose(_np(r["S"]), ref["S"], atol=1e-9)
    np.testing.assert_allclose(_np(r["var"]), np.diagonal(ref["Ps"], axis1=2, axis2=3), atol=1e-9)
    np.testing.assert_allclose(_np(r["mle"]), ref["mle"], rtol=1e-9)
    assert int(_np(r["status"]).sum()) == 0


def test_abi_rejects_inconsistent_layout_flags():
    from metran_amd import _lib
    from metran_amd.engine import BatchedKalman, MetranHipError

    d = make_dfm_batch(2, 8, 2, 10, seed=1)
    kf = BatchedKalman()
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    res = kf._alloc_outputs(2, ["F", "Pf", "Xp", "Pp"])
    o = kf._outputs_struct(res)
    o.flags = 1  # PACKED_SYM with the full-square stride
    prob, keep, B = kf._problem(d["phi"], d["q"], 1, None, None)
    with pytest.raises(MetranHipError, match="mk_record_stride_sym"):
        _lib.check(kf._L.mk_filter(kf._ctx, ctypes.byref(prob), ctypes.byref(o)))


def test_raw_c_abi_without_torch():
    """INTEGRATION.md section 1 made executable: a host that has no torch drives the library with mk_malloc /
    mk_memcpy_h2d / mk_memset / mk_memcpy_d2h; here with packed-symmetric records, checked against the oracle."""
    from metran_amd import _lib
    from metran_amd._lib import Outputs, Problem, check

    L = _lib.lib()
    B, N, K, T = 3, 8, 2, 50
    n = N + K
    d = make_dfm_batch(B, N, K, T, seed=77, missing=0.1)
    ref = oracle.dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"])
    ctx = ctypes.c_void_p()
    check(L.mk_create(0, ctypes.byref(ctx)))
    bufs = []

    def dev(nbytes, host=None):
        p = ctypes.c_void_p()
        check(L.mk_malloc(ctx, nbytes, ctypes.byref(p)))
        bufs.append(p)
        if host is not None:
            h = np.ascontiguousarray(host, dtype=np.float64)
            check(L.mk_memcpy_h2d(ctx, p, h.ctypes.data_as(ctypes.c_void_p), h.nbytes))
        else:
            check(L.mk_memset(ctx, p, 0, nbytes))
        return p

    RS = int(L.mk_record_stride_sym(n))
    nv = n + n * (n + 1) // 2
    d_obs, d_phi, d_q, d_ld = (dev(d[k].nbytes, d[k]) for k in ("obs", "phi", "q", "loadings"))
    rec = {k: dev(B * T * RS * 8) for k in ("pred", "filt", "smooth")}
    d_mle, d_sc, d_st = dev(B * 8), dev(B * 8), dev(B * 4)
    prob = Problem(B, B, T, N, K, 1, d_obs, d_phi, d_q, d_ld, None, None, None, 0, None, None)
    at = lambda p, k: ctypes.c_void_p(p.value + 8 * k)  # noqa: E731
    out = Outputs(d_mle, at(rec["filt"], nv), at(rec["filt"], nv + 1), d_sc, rec["filt"], at(rec["filt"], n), rec["pred"],
                  at(rec["pred"], n), rec["smooth"], at(rec["smooth"], n), d_st, 0, None, None, RS, 1)
    check(L.mk_filter_smooth(ctx, ctypes.byref(prob), ctypes.byref(out)))
    check(L.mk_sync(ctx))
    host = {k: np.empty((B, T, RS)) for k in rec}
    for k in rec:
        check(L.mk_memcpy_d2h(ctx, host[k].ctypes.data_as(ctypes.c_void_p), rec[k], host[k].nbytes))
    mle = np.empty(B)
    check(L.mk_memcpy_d2h(ctx, mle.ctypes.data_as(ctypes.c_void_p), d_mle, mle.nbytes))
    st = np.empty(B, dtype=np.uint32)
    check(L.mk_memcpy_d2h(ctx, st.ctypes.data_as(ctypes.c_void_p), d_st, st.nbytes))
    for p in bufs:
        check(L.mk_free(ctx, p))
    check(L.mk_destroy(ctx))
    iu = np.triu_indices(n)
    np.testing.assert_allclose(mle, ref["mle"], rtol=1e-9)
    assert not st.any()
    np.testing.assert_allclose(host["filt"][..., :n], ref["F"], atol=1e-10)
    np.testing.assert_allclose(host["pred"][..., n:nv], ref["Pp"][..., iu[0], iu[1]], atol=1e-10)
    np.testing.assert_allclose(host["smooth"][..., :n], ref["S"], atol=1e-9)
    np.testing.assert_allclose(host["smooth"][..., n:nv], ref["Ps"][..., iu[0], iu[1]], atol=1e-9)
    assert not host["smooth"][..., nv:].any()    # pad doubles are written as zeros (whole cache lines)


def test_accumulating_kernel_timing():
    """mk_enable_timing(ctx, 2) + mk_kernel_ms_totals: one hipEvent pair per launch, summed when asked (what bench.py
    reads after its timed loop), next to the mode-1 'most recent launch' query."""
    from metran_amd.engine import BatchedKalman

    d = make_dfm_batch(64, 8, 2, 200, seed=5, missing=0.1)
    kf = BatchedKalman()
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    kf.enable_timing(True, accumulate=True)
    for _ in range(3):
        kf.filter_smooth(d["phi"], d["q"])
    kf.loglik(d["phi"], d["q"])
    f_ms, f_n, s_ms, s_n = kf.kernel_ms_totals()
    assert (f_n, s_n) == (4, 3) and 0.0 < f_ms < 100.0 and 0.0 < s_ms < 100.0
    assert kf.kernel_ms_totals() == (0.0, 0, 0.0, 0)          # collected pairs are recycled
    kf.enable_timing(True)                                    # mode 1: the most recent launch of each kind
    kf.filter_smooth(d["phi"], d["q"])
    f1, s1 = kf.last_kernel_ms()
    assert 0.0 < f1 < 100.0 and 0.0 < s1 < 100.0
    kf.enable_timing(False)


def test_wide_filter_auto_rule():
    """MK_VARIANT_WIDE_FILTER = 0 ("auto", the library default): one state per lane up to two models per SIMD, the split
    layout above; both equal the oracle, and "auto" is bit-for-bit one of the two forced variants on either side of the
    threshold."""
    import torch

    from metran_amd.engine import BatchedKalman
    from metran_amd.synthetic import make_dfm_batch_torch

    N, K, T = 32, 4, 12
    simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    small = make_dfm_batch(5, N, K, T, seed=71, missing=0.3, first_step="random")
    ref = oracle.dfm_batch(small["obs"], small["phi"], small["q"], small["loadings"], smooth=False)
    res = {}
    for name in ("auto", "lane_per_state", "split"):
        kf = BatchedKalman()
        kf.set_variant("wide_filter", name)
        assert kf.get_variant("wide_filter") == name
        kf.set_observations(small["obs"]).set_loadings(small["loadings"])
        r = kf.filter(small["phi"], small["q"])
        np.testing.assert_allclose(_np(r["Pf"]), ref["Pf"], atol=1e-10)
        np.testing.assert_allclose(_np(r["mle"]), ref["mle"], rtol=1e-9)
        res[name] = (r["F"].clone(), r["Pf"].clone(), r["mle"].clone())
        kf.close()
    assert all(torch.equal(a, b) for a, b in zip(res["auto"], res["lane_per_state"]))      # 5 models: one state per lane
    B = 2 * simds + 64
    big = make_dfm_batch_torch(B, N, K, T, seed=72, device=torch.device("cuda", 0), missing=0.3)
    out = {}
    for name in ("auto", "lane_per_state", "split"):
        kf = BatchedKalman(layout="time_major")
        kf.set_variant("wide_filter", name)
        kf.set_observations(big["obs"]).set_loadings(big["loadings"])
        out[name] = kf.loglik(big["phi"], big["q"]).clone()
        kf.close()
    assert torch.equal(out["auto"], out["split"])                                          # > 2 models per SIMD: split layout
    assert float(((out["split"] - out["lane_per_state"]) / out["split"]).abs().max()) < 1e-12
