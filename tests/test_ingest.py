"""Row f3 (observation ingestion).  CPU: the pandas-side mirror of Metran.__init__'s handling of
`oseries` (metran/metran.py:102-199, 508-579) on the real-data golden; GPU: mk_standardize,
mk_mask_observations and mk_pack_observations against pandas / the oracle's set_observations."""
import numpy as np
import pandas as pd
import pytest

from conftest import load_golden


def _g1_series(g1):
    """The five unstandardised example series rebuilt from the golden (obs * std + mean on its index),
    each reduced to its own observation dates as the CSV files are."""
    idx = pd.DatetimeIndex(g1["index_ns"].astype("datetime64[ns]"))
    raw = g1["obs"] * g1["oseries_std"] + g1["oseries_mean"]
    out = []
    for j in range(raw.shape[1]):
        s = pd.Series(raw[:, j], index=idx, name="B21B021400%d" % (j + 1)).dropna()
        out.append(s)
    return out, idx


def test_combine_standardize_matches_reference_pipeline(g1):
    from metran_amd import ingest

    series, idx = _g1_series(g1)
    frame, names = ingest.combine_series(series)
    assert names == ["B21B021400%d" % i for i in range(1, 6)]
    assert frame.shape == g1["obs"].shape and (frame.index == idx).all()    # daily grid, same span
    pairs = ingest.cross_section_pairs(frame)
    assert list(pairs.values) == [343, 332, 332, 332, 331]
    std_frame, std, mean = ingest.standardize(frame)
    np.testing.assert_allclose(std, g1["oseries_std"], rtol=1e-12)
    np.testing.assert_allclose(mean, g1["oseries_mean"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(std_frame.values, g1["obs"], rtol=0, atol=1e-12, equal_nan=True)


def test_combine_series_errors():
    from metran_amd import ingest

    idx = pd.date_range("2000-01-01", periods=30, freq="D")
    a = pd.Series(np.arange(30.0), index=idx, name="a")
    with pytest.raises(Exception, match="at least 2 series"):
        ingest.combine_series([a])
    with pytest.raises(TypeError, match="list, tuple, or pandas.DataFrame"):
        ingest.combine_series(a)
    with pytest.raises(Exception, match="multiple columns"):
        ingest.combine_series([pd.DataFrame({"x": a, "y": a}), a])
    with pytest.raises(TypeError, match="DatetimeIndex"):
        ingest.combine_series(pd.DataFrame({"x": np.arange(5.0), "y": np.arange(5.0)}))
    b = pd.Series(np.arange(30.0), index=idx)             # unnamed -> "Series2" (metran.py:553-554)
    frame, names = ingest.combine_series([a, b], tmin="2000-01-05", tmax="2000-01-20")
    assert names == ["a", "Series2"] and frame.shape == (16, 2)
    short = pd.Series([1.0, 2.0], index=idx[:2], name="short")
    frame, _ = ingest.combine_series([a, short])
    with pytest.raises(Exception, match="less than 20 for series short"):
        ingest.cross_section_pairs(frame)


def test_observation_batch_padding(g1):
    from metran_amd import ingest

    series, _ = _g1_series(g1)
    idx = pd.date_range("2001-03-01", periods=90, freq="D")
    rng = np.random.default_rng(5)
    small = [pd.Series(rng.normal(size=90), index=idx, name="s%d" % i) for i in range(5)]
    small[2].iloc[10:40] = np.nan
    batch = ingest.ObservationBatch([series, small])
    assert batch.shape == (2, 6255, 5) and list(batch.lengths) == [6255, 90]
    assert np.isnan(batch.obs[1, 90:]).all() and not np.isnan(batch.obs[1, :90, 0]).any()
    fr = batch.frame(1, np.zeros((6255, 5)))
    assert fr.shape == (90, 5) and list(fr.columns) == ["s0", "s1", "s2", "s3", "s4"]


# --------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["model_major", "time_major"])
def test_device_standardize_mask_pack(g1, layout):
    import torch

    import oracle
    from metran_amd import ingest
    from metran_amd.engine import BatchedKalman

    series, _ = _g1_series(g1)
    rng = np.random.default_rng(11)
    idx = pd.date_range("2001-03-01", periods=400, freq="D")
    others = []
    for m in range(3):
        ss = []
        for i in range(5):
            v = rng.normal(loc=3.0 * i, scale=1.0 + i, size=400)
            v[rng.random(400) < 0.3] = np.nan
            ss.append(pd.Series(v, index=idx, name="m%d_s%d" % (m, i)))
        others.append(ss)
    batch = ingest.ObservationBatch([series] + others)
    kf = BatchedKalman(0, layout=layout)
    batch.upload(kf)
    # (1) standardisation == pandas per model
    np.testing.assert_allclose(batch.std[0], g1["oseries_std"], rtol=1e-12)
    np.testing.assert_allclose(batch.mean[0], g1["oseries_mean"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(kf.obs[0].cpu().numpy(), g1["obs"], rtol=0, atol=1e-12, equal_nan=True)
    for m in range(3):
        fr, _ = ingest.combine_series(others[m])
        sf, sd, mu = ingest.standardize(fr)
        np.testing.assert_allclose(batch.std[m + 1], sd, rtol=1e-12)
        np.testing.assert_allclose(kf.obs[m + 1, :400].cpu().numpy(), sf.values, rtol=0, atol=1e-12, equal_nan=True)
        assert bool(torch.isnan(kf.obs[m + 1, 400:]).all())
    # the standardised real-data record gives the reference objective
    kf.set_loadings(np.repeat(g1["loadings"][None], 4, 0))
    phi, q = kf.params_from_alpha(np.repeat(g1["alpha_star"][None], 4, 0))
    assert abs(float(kf.loglik(phi, q)[0]) - 2332.327069381027) < 1e-8
    # (2) packing == the oracle's restatement of set_observations, incl. the -1e10 quirk
    obs_h = kf.obs.cpu().numpy().copy()
    obs_h[1, 3, 2] = -1e10
    kf2 = BatchedKalman(0, layout=layout).set_observations(obs_h)
    o, ix, cnt = (t.cpu().numpy() for t in kf2.pack_observations())
    for r in range(4):
        ro, ri, rc = oracle.set_observations(obs_h[r])
        np.testing.assert_array_equal(o[r], ro)
        np.testing.assert_array_equal(ix[r], ri)
        np.testing.assert_array_equal(cnt[r], rc)
    assert o[1, 3, 2] == 0.0
    # (3) mask / unmask: the golden masked objective of the reference (make_golden.g1_real)
    kf.set_observations(np.repeat(g1["obs"][None], 2, 0)).set_loadings(np.repeat(g1["loadings"][None], 2, 0))
    phi, q = kf.params_from_alpha(np.repeat(g1["alpha_star"][None], 2, 0))
    mask = np.zeros((2, 6255, 5), dtype=bool)
    mask[1, int(g1["mask_t"]), 4] = True  # series B21B0214005
    kf.mask_observations(mask)
    mle = kf.loglik(phi, q).cpu().numpy()
    assert abs(mle[0] - 2332.327069381027) < 1e-8 and abs(mle[1] - float(g1["masked_mle_star"])) < 1e-8
    kf.unmask_observations()
    mle = kf.loglik(phi, q).cpu().numpy()
    assert abs(mle[1] - 2332.327069381027) < 1e-8
    with pytest.raises(ValueError, match="Dimensions of mask"):
        kf.mask_observations(np.zeros((2, 10, 5)))
