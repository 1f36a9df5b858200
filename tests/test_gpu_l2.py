"""GPU (`-m gpu`): the L2 metric (lmi_set_metric(LMI_METRIC_L2); BASELINE north_star names an "exact L2/IP" scan, the
reference itself has only 1 - ip).  Both scan modes against the oracle's L2 restatement: ids and squared distances
bit-identical; distances within 1e-4 relative of float64 brute force (north_star's tolerance); short and empty
buckets, k variants, un-normalised vectors; on unit-norm vectors the neighbours equal the inner-product ones."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(seed, n, d, L, nq, unit):
    rs = np.random.RandomState(seed)
    cen = rs.randn(L, d).astype(np.float32)
    lab = rs.randint(0, L, size=n)
    X = cen[lab] + 0.8 * rs.randn(n, d).astype(np.float32)
    Q = cen[rs.randint(0, L, size=nq)] + 0.8 * rs.randn(nq, d).astype(np.float32)
    if unit:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    else:
        X *= rs.uniform(0.3, 3.0, size=(n, 1)).astype(np.float32)
    lab[lab == 2] = 3                      # bucket 2 empty
    few = np.flatnonzero(lab == 5)
    lab[few[4:]] = 6                       # bucket 5 holds 4 objects
    order = np.argsort(((Q[:, None, :] - cen[None, :, :]) ** 2).sum(-1), axis=1)[:, :4].astype(np.int32)
    order[:8, 0] = 5
    order[8:16, 1] = 2
    return X.astype(np.float32), Q.astype(np.float32), lab.astype(np.int64), np.ascontiguousarray(order)


@pytest.mark.parametrize("d,unit", [(64, False), (768, True), (45, False), (100, False)])
def test_l2_matches_oracle(oracle, d, unit):
    from learnedmetricindex_amd import _capi

    L = 12
    X, Q, lab, order = make(d, 6000, d, L, 160, unit)
    res = {}
    for pf in (True, False):
        idx = _capi.Index(0, prefilter=pf, metric="l2")
        idx.set_buckets(X, lab, L)
        res[pf] = idx.scan_topk(Q, order, 10)
        res[pf, "k5"] = idx.scan_topk(Q, order[:, :3], 5)
        rows, ids = idx.read_bucket(7)                      # the norm column is internal
        sel = np.flatnonzero(lab == 7)
        np.testing.assert_array_equal(rows, X[sel])
        np.testing.assert_array_equal(ids, sel + 1)
        idx.close()
    for key in (True, (True, "k5")):
        other = False if key is True else (False, "k5")
        np.testing.assert_array_equal(res[key][1], res[other][1])
        np.testing.assert_array_equal(res[key][0], res[other][0])
    do, io, _ = oracle.search(None, None, X, Q, lab[:, None], 4, 10, nthreads=8, bucket_order=order[:, :, None], metric="l2")
    np.testing.assert_array_equal(res[True][1], io)
    np.testing.assert_array_equal(res[True][0].astype(np.float64), do)
    do5, io5, _ = oracle.search(None, None, X, Q, lab[:, None], 3, 5, nthreads=8, bucket_order=order[:, :3, None], metric="l2")
    np.testing.assert_array_equal(res[True, "k5"][1], io5)
    np.testing.assert_array_equal(res[True, "k5"][0].astype(np.float64), do5)
    # float64 brute force over the visited buckets: distances within 1e-4 relative (atol for near-zero distances)
    d32, i32 = res[True]
    ref = ((Q.astype(np.float64)[:, None, :] - X.astype(np.float64)[None, :, :]) ** 2).sum(-1)
    real = (i32 > 0) & (d32 < 1e30)
    qq, jj = np.nonzero(real)
    np.testing.assert_allclose(d32[qq, jj], ref[qq, i32[qq, jj].astype(np.int64) - 1], rtol=1e-4, atol=1e-5 * float(ref.mean()))
    assert np.all(d32[:8, :][i32[:8, :] > 0] >= 0)
    if unit:  # L2^2 = 2 (1 - ip): same neighbours as the inner-product scan
        ip = _capi.Index(0, metric="ip")
        ip.set_buckets(X, lab, L)
        dip, iip = ip.scan_topk(Q, order, 10)
        ip.close()
        assert (iip == i32).mean() > 0.98


def test_l2_through_the_li_api(oracle):
    """LearnedIndex.search(..., metric="l2") on fixture G1's index: MLP routing as usual, squared-L2 scan == oracle."""
    import pandas as pd

    from helpers import inputs_for, layers_from, load_golden
    from learnedmetricindex_amd.li.LearnedIndex import LearnedIndex
    from learnedmetricindex_amd.li.model import network_from_layers

    g = load_golden("G1")
    Xn, Qn, Xs, Qs = inputs_for("G1", g)
    layers = layers_from(g)
    li = LearnedIndex(network_from_layers(layers), {}, [(i,) for i in range(12)])
    df = pd.DataFrame(Xs * np.float32(1.7))        # un-normalised: L2 and ip order differ
    df.index += 1
    dp = g["data_prediction"].astype(np.int64)
    d, n, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10, metric="l2")
    do, no, _ = oracle.search(layers, Qn, Xs * np.float32(1.7), Qs, dp, 3, 10, nthreads=4, metric="l2")
    np.testing.assert_array_equal(n, no)
    np.testing.assert_array_equal(d, do)
    d_ip, n_ip, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10)     # the metric is part of the resident index's key
    do_ip, no_ip, _ = oracle.search(layers, Qn, Xs * np.float32(1.7), Qs, dp, 3, 10, nthreads=4)
    np.testing.assert_array_equal(n_ip, no_ip)
    li.close()


def test_l2_index_survives_save_and_load(oracle, tmp_path):
    """the metric is part of the on-disk index -- an L2 index reloads as an L2 index (meta.json "metric"),
    and a version-1 directory (no such key) still loads as inner product."""
    import json

    import pandas as pd

    from helpers import inputs_for, layers_from, load_golden
    from learnedmetricindex_amd.index_io import load_index, save_index
    from learnedmetricindex_amd.li.LearnedIndex import LearnedIndex
    from learnedmetricindex_amd.li.model import network_from_layers

    g = load_golden("G1")
    Xn, Qn, Xs, Qs = inputs_for("G1", g)
    layers = layers_from(g)
    li = LearnedIndex(network_from_layers(layers), {}, [(i,) for i in range(12)])
    X = Xs * np.float32(1.7)
    df = pd.DataFrame(X)
    df.index += 1
    dp = g["data_prediction"].astype(np.int64)
    d, n, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10, metric="l2")
    save_index(str(tmp_path / "idx"), li, [12])
    li.close()
    meta = json.load(open(tmp_path / "idx" / "meta.json"))
    assert meta["metric"] == "l2" and meta["version"] == 2
    li2, ncat = load_index(str(tmp_path / "idx"))
    assert li2._engine.metric == "l2" and ncat == [12]
    d2, n2, _ = li2.search_resident(Qn, Qs, [12], 3, 10)
    np.testing.assert_array_equal(n2, n)
    np.testing.assert_array_equal(d2, d)
    do, no, _ = oracle.search(layers, Qn, X, Qs, dp, 3, 10, nthreads=4, metric="l2")
    np.testing.assert_array_equal(n2, no)
    li2.close()
    # a version-1 directory: no "metric" key -> inner product
    del meta["metric"]
    meta["version"] = 1
    json.dump(meta, open(tmp_path / "idx" / "meta.json", "w"))
    li3, _ = load_index(str(tmp_path / "idx"))
    assert li3._engine.metric == "ip"
    d3, n3, _ = li3.search_resident(Qn, Qs, [12], 3, 10)
    do_ip, no_ip, _ = oracle.search(layers, Qn, X, Qs, dp, 3, 10, nthreads=4)
    np.testing.assert_array_equal(n3, no_ip)
    li3.close()
