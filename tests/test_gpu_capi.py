"""GPU (`-m gpu`): parity of the HIP path against the CPU oracle, through the C ABI.

Bar: integer outputs (bucket order, ids) bit-exact; float outputs (logits, similarities,
distances) bit-exact too, because both sides evaluate the same k-ordered binary32 fmaf chain
(north_star asks for 1e-4 relative; the tests assert equality and would report the first
differing element).
"""
import numpy as np
import pytest

from helpers import inputs_for, layers_from, load_golden, compare_modulo_near_ties

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from learnedmetricindex_amd import _capi

    _capi.lib()
    return _capi


def test_knn_ip_matches_oracle(capi, oracle):
    rs = np.random.RandomState(11)
    for (nq, nb, d) in ((37, 1000, 64), (5, 7, 45), (130, 4500, 768), (1, 33, 8), (64, 129, 100)):
        xq = rs.randn(nq, d).astype(np.float32)
        xb = rs.randn(nb, d).astype(np.float32)
        D, I = capi.knn_ip(xq, xb, 10)
        Do, Io = oracle.knn_ip(xq, xb, 10, nthreads=4)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)


def test_knn_ip_duplicates_and_padding(capi, oracle):
    rs = np.random.RandomState(12)
    xb = rs.randn(300, 32).astype(np.float32)
    xb[100:140] = xb[7]  # 41 identical rows: ties -> lowest rows first
    xq = np.concatenate([xb[7:8] * 2, rs.randn(4, 32).astype(np.float32)])
    D, I = capi.knn_ip(xq, xb, 10)
    Do, Io = oracle.knn_ip(xq, xb, 10)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert list(I[0]) == [7] + list(range(100, 109))
    D, I = capi.knn_ip(xq, xb[:4], 10)  # nb < k: faiss padding
    assert np.all(I[:, 4:] == -1) and np.all(D[:, 4:] == -np.finfo(np.float32).max)
    Do, Io = oracle.knn_ip(xq, xb[:4], 10)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


@pytest.mark.parametrize("name", ["G1", "G3", "G5", "G6"])
def test_mlp_logits_and_bucket_order(capi, oracle, name):
    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    layers = layers_from(g)
    nb = int(g["n_buckets"])
    idx = capi.Index(0)
    idx.set_mlp(layers)
    order, logits = idx.mlp_topk(Qn, nb, want_logits=True)
    np.testing.assert_array_equal(logits, oracle.forward_logits(layers, Qn, nthreads=4))
    np.testing.assert_array_equal(order, oracle.rank_classes(logits, nb))
    np.testing.assert_array_equal(order, g["ref_bucket_order"][:, :, 0])  # == the reference's torch path
    full = idx.mlp_topk(Qn, logits.shape[1])
    np.testing.assert_array_equal(full, oracle.rank_classes(logits, logits.shape[1]))
    idx.close()


@pytest.mark.parametrize("name", ["G1", "G3", "G4", "G5", "G6"])
def test_search_matches_oracle_and_reference(capi, oracle, name):
    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    layers = layers_from(g)
    nb, k = int(g["n_buckets"]), int(g["k"])
    dp = g["data_prediction"]
    L = layers[-1][0].shape[0]
    idx = capi.Index(0, chunk_rows=256)  # small chunks: several chunks per bucket even at test sizes
    idx.set_mlp(layers)
    idx.set_buckets(Xs, dp[:, 0], L)
    d, i, bo = idx.search(Qn, Qs, nb, k)
    do, io, boo = oracle.search(layers, Qn, Xs, Qs, dp, nb, k, nthreads=4)
    np.testing.assert_array_equal(bo, boo[:, :, 0])
    np.testing.assert_array_equal(i, io)
    np.testing.assert_array_equal(d.astype(np.float64), do)
    # and against the reference-generated fixture (different fp32 summation order inside BLAS)
    compare_modulo_near_ties(g["ref_dists"], g["ref_nns"], d.astype(np.float64), i)
    # scan alone, fed with the reference's bucket order
    d2, i2 = idx.scan_topk(Qs, g["ref_bucket_order"][:, :, 0], k)
    np.testing.assert_array_equal(i2, io)
    idx.close()


def test_k_variants_G4(capi, oracle):
    g = load_golden("G4")
    Xn, Qn, Xs, Qs = inputs_for("G4", g)
    layers = layers_from(g)
    dp = g["data_prediction"]
    idx = capi.Index(0, chunk_rows=256)
    idx.set_mlp(layers)
    idx.set_buckets(Xs, dp[:, 0], 12)
    for tag, nb, k in (("nb1_k5", 1, 5), ("nb3_k5", 3, 5), ("nb3_k15", 3, 15), ("nb12_k10", 12, 10)):
        d, i, bo = idx.search(Qn, Qs, nb, k)
        do, io, _ = oracle.search(layers, Qn, Xs, Qs, dp, nb, k)
        assert d.shape == g[f"{tag}_dists"].shape
        np.testing.assert_array_equal(i, io)
        np.testing.assert_array_equal(d.astype(np.float64), do)
        compare_modulo_near_ties(g[f"{tag}_dists"], g[f"{tag}_nns"], d.astype(np.float64), i)
    with pytest.raises(capi.LmiError):
        idx.search(Qn, Qs, 2, 25)
    idx.close()


def test_error_paths(capi):
    idx = capi.Index(0)
    with pytest.raises(capi.LmiError, match="not built"):
        idx.scan_topk(np.zeros((2, 8), np.float32), np.zeros((2, 1), np.int32))
    with pytest.raises(capi.LmiError, match="no MLP"):
        idx.n_classes = 3
        idx.mlp_topk(np.zeros((2, 8), np.float32), 1)
    with pytest.raises(capi.LmiError, match="outside"):
        idx.buckets_begin(np.array([0, 5]), 8, 3)
    idx.close()


def test_timings_ring_mean(capi):
    """lmi_timings_reset / lmi_timings_mean: the phase times of the calls since the reset, averaged with one
    stream synchronisation (bench.py reads them once after its timed loop)."""
    g = load_golden("G3")
    Xn, Qn, Xs, Qs = inputs_for("G3", g)
    layers = layers_from(g)
    nb, k = int(g["n_buckets"]), int(g["k"])
    L = layers[-1][0].shape[0]
    idx = capi.Index(0)
    idx.set_mlp(layers)
    idx.set_buckets(Xs, g["data_prediction"][:, 0], L)
    idx.search(Qn, Qs, nb, k)
    idx.timings_reset()
    ms0, n0 = idx.timings_mean()
    assert n0 == 0 and not ms0.any()
    for _ in range(3):
        idx.search(Qn, Qs, nb, k)
    ms, n = idx.timings_mean()
    assert n == 3
    last = idx.timings()
    assert ms[capi.T_TOTAL] > 0 and ms[capi.T_SCAN] > 0 and ms[capi.T_INFERENCE] > 0
    assert ms[capi.T_TOTAL] >= ms[capi.T_SCAN]
    assert 0.2 * last[capi.T_TOTAL] < ms[capi.T_TOTAL] < 5 * last[capi.T_TOTAL]
    for _ in range(140):  # more calls than the ring holds: the newest 128 are averaged
        idx.mlp_topk(Qn[:8], nb)
    _, n = idx.timings_mean()
    assert n == 128
    # timing levels: 1 keeps the call's first and last event, 0 records nothing (each event is a bubble)
    idx.set_timing(1)
    idx.search(Qn, Qs, nb, k)
    t1 = idx.timings()
    assert t1[capi.T_TOTAL] > 0 and t1[capi.T_SCAN] == 0 and t1[capi.T_PF_EMIT] == 0
    idx.set_timing(0)
    d0, i0, _ = idx.search(Qn, Qs, nb, k)
    assert not idx.timings().any()
    idx.set_timing(2)
    d2, i2, _ = idx.search(Qn, Qs, nb, k)
    assert idx.timings()[capi.T_SCAN] > 0
    np.testing.assert_array_equal(i0, i2)
    idx.set_timing(3)   # every phase from hipEvents between the kernels (the round-1..4 form of level 2)
    d3, i3, _ = idx.search(Qn, Qs, nb, k)
    t3 = idx.timings()
    assert t3[capi.T_SCAN] > 0 and t3[capi.T_PF_EMIT] >= 0
    np.testing.assert_array_equal(i0, i3)
    with pytest.raises(capi.LmiError):
        idx.set_timing(4)
    idx.close()
