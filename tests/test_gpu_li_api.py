"""GPU (`-m gpu`): the drop-in `li` API (LearnedIndex.search, NeuralNetwork.predict*) against the
reference-generated fixtures and the oracle.  Reads like a test of the reference would: build the
objects the way search.py does (DataFrames with a 1-based index) and call `search`."""
import pickle

import numpy as np
import pandas as pd
import pytest
import torch

from helpers import compare_modulo_near_ties, inputs_for, layers_from, load_golden

pytestmark = pytest.mark.gpu

MODEL_OF = {"G1": "MLP", "G3": "MLP-4", "G4": "MLP", "G5": "MLP-4", "G6": "MLP-5"}


def frame(X):
    df = pd.DataFrame(X)
    df.index += 1  # search.py:190-191
    return df


def make_index(name, g):
    from learnedmetricindex_amd.li.LearnedIndex import LearnedIndex
    from learnedmetricindex_amd.li.model import NeuralNetwork

    layers = layers_from(g)
    net = NeuralNetwork(input_dim=layers[0][0].shape[1], output_dim=layers[-1][0].shape[0], model_type=MODEL_OF[name])
    lin = [m for m in net.model.layers if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for m, (W, b) in zip(lin, layers):
            m.weight.copy_(torch.from_numpy(W))
            m.bias.copy_(torch.from_numpy(b))
    L = layers[-1][0].shape[0]
    return LearnedIndex(net, {}, [(i,) for i in range(L)]), net


@pytest.mark.parametrize("name", ["G1", "G3", "G4", "G5", "G6"])
def test_search_dropin(oracle, name):
    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    li, net = make_index(name, g)
    nb, k = int(g["n_buckets"]), int(g["k"])
    dp = g["data_prediction"].astype(np.int64)
    nav, srch = frame(Xn), frame(Xs)
    cols_before = list(nav.columns)
    dists, nns, mt = li.search(data_navigation=nav, queries_navigation=Qn, data_search=srch, queries_search=Qs,
                               data_prediction=dp, n_categories=[int(g["n_categories"][0])], n_buckets=nb, k=k)
    assert dists.dtype == np.float64 and nns.dtype == np.uint32 and dists.shape == nns.shape == (Qs.shape[0], k)
    assert list(nav.columns) == cols_before  # never mutated
    for key in ("inference", "search_within_buckets", "seq_search", "sort", "search"):
        assert key in mt and mt[key] >= 0.0
    compare_modulo_near_ties(g["ref_dists"], g["ref_nns"], dists, nns)
    do, no, _ = oracle.search(layers_from(g), Qn, Xs, Qs, dp, nb, k, nthreads=4)
    np.testing.assert_array_equal(nns, no)
    np.testing.assert_array_equal(dists, do)
    # second call reuses the HBM-resident index; results are reproducible
    d2, n2, _ = li.search(nav, Qn, srch, Qs, dp, [int(g["n_categories"][0])], nb, k)
    np.testing.assert_array_equal(n2, nns)
    np.testing.assert_array_equal(d2, dists)
    bo, t_inf = li._precompute_bucket_order(Qn, nb, [int(g["n_categories"][0])])
    np.testing.assert_array_equal(bo, g["ref_bucket_order"])
    li.close()


def test_aliased_frames_and_custom_labels(oracle):
    """the reference cannot run with data_search is data_navigation; the drop-in can.
    Ids are the DataFrame's index labels (Q2), whatever they are."""
    g = load_golden("G1")
    Xn, Qn, Xs, Qs = inputs_for("G1", g)
    li, _ = make_index("G1", g)
    dp = g["data_prediction"].astype(np.int64)
    df = frame(Xs)
    d, n, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10)
    compare_modulo_near_ties(g["ref_dists"], g["ref_nns"], d, n)
    df2 = pd.DataFrame(Xs, index=np.arange(Xs.shape[0]) * 3 + 1000)
    d2, n2, _ = li.search(df2, Qn, df2, Qs, dp, [12], 3, 10)
    np.testing.assert_array_equal(n2, (n.astype(np.int64) - 1) * 3 + 1000)
    np.testing.assert_array_equal(d2, d)
    li.close()


def test_predict_and_predict_proba(oracle):
    for name in ("G1", "G6"):
        g = load_golden(name)
        Xn, Qn, Xs, Qs = inputs_for(name, g)
        li, net = make_index(name, g)
        layers = layers_from(g)
        probs, classes = net.predict_proba(torch.from_numpy(Qn))
        po, co = oracle.predict_proba(layers, Qn)
        assert probs.dtype == np.float32 and classes.dtype == np.int64
        np.testing.assert_array_equal(classes, co)
        np.testing.assert_array_equal(probs, po)  # same binary32 operation sequence on both sides
        nb = int(g["n_buckets"])
        np.testing.assert_array_equal(classes[:, :nb], g["ref_classes_top"])
        np.testing.assert_allclose(probs[:, :nb], g["ref_probs_top"], rtol=2e-5, atol=1e-7)
        pred = net.predict(torch.from_numpy(Xn[:3000]))
        np.testing.assert_array_equal(pred, g["data_prediction"][:3000, 0] if name != "G4" else pred)
        np.testing.assert_array_equal(pred, oracle.predict(layers, Xn[:3000]))


def test_pickle_roundtrip_and_k_quirks():
    g = load_golden("G4")
    Xn, Qn, Xs, Qs = inputs_for("G4", g)
    li, _ = make_index("G4", g)
    dp = g["data_prediction"].astype(np.int64)
    df = frame(Xs)
    d, n, _ = li.search(df, Qn, df, Qs, dp, [12], 1, 5)     # Q3: single bucket ignores k
    assert d.shape == (Qs.shape[0], 10)
    compare_modulo_near_ties(g["nb1_k5_dists"], g["nb1_k5_nns"], d, n)
    with pytest.raises(AssertionError):
        li.search(df, Qn, df, Qs, dp, [12], 2, 25)           # LearnedIndex.py:142-146
    li2 = pickle.loads(pickle.dumps(li))                      # search.py:234-241 pickles the index
    d2, n2, _ = li2.search(df, Qn, df, Qs, dp, [12], 3, 15)
    compare_modulo_near_ties(g["nb3_k15_dists"], g["nb3_k15_nns"], d2, n2)
    li.close()
    li2.close()


def test_bucket_read_roundtrip():
    from learnedmetricindex_amd import _capi

    rs = np.random.RandomState(3)
    X = rs.randn(1000, 45).astype(np.float32)
    lab = rs.randint(0, 7, size=1000)
    lab[lab == 4] = 5  # bucket 4 empty
    idx = _capi.Index(0, chunk_rows=256)
    idx.set_buckets(X, lab, 7)
    np.testing.assert_array_equal(idx.bucket_sizes(), np.bincount(lab, minlength=7))
    for b in range(7):
        rows, ids = idx.read_bucket(b)
        sel = np.flatnonzero(lab == b)
        np.testing.assert_array_equal(rows, X[sel])
        np.testing.assert_array_equal(ids, sel + 1)
    idx.close()


@pytest.mark.parametrize("name", ["G2", "G7", "G8"])
def test_multi_level_index(oracle, name):
    """len(n_categories) > 1 -- priority-queue navigation with the HIP MLP for every node,
    one scan call for all ranks.  Bucket order and results identical to the oracle's restatement;
    against the reference fixture modulo near-equal priorities / distances.  G2 `[4,3]`, G7 `[10,10]` (the reference's
    default tree, 10 buckets visited), G8 `[4,3,2]` (three levels); G7 a second time with the navigation chunked
    (`_NAV_QUEUE_BYTES` small), which must not change anything."""
    from learnedmetricindex_amd.li.LearnedIndex import LearnedIndex
    from learnedmetricindex_amd.li.model import NeuralNetwork
    from test_oracle_multilevel import internal_of

    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    ncat = [int(v) for v in g["n_categories"]]
    nb, k = int(g["n_buckets"]), int(g["k"])

    def net_from(layers):
        net = NeuralNetwork(input_dim=layers[0][0].shape[1], output_dim=layers[-1][0].shape[0], model_type="MLP")
        lin = [m for m in net.model.layers if isinstance(m, torch.nn.Linear)]
        with torch.no_grad():
            for m, (W, b) in zip(lin, layers):
                m.weight.copy_(torch.from_numpy(W))
                m.bias.copy_(torch.from_numpy(b))
        return net

    internal = internal_of(g)
    bucket_paths = [tuple(int(v) for v in p) for p in g["bucket_paths"]]
    li = LearnedIndex(net_from(layers_from(g)), {p: net_from(l) for p, l in internal}, bucket_paths)
    dp = g["data_prediction"].astype(np.int64)
    nav, srch = frame(Xn), frame(Xs)
    dists, nns, mt = li.search(nav, Qn, srch, Qs, dp, ncat, nb, k)
    bo, _ = li._precompute_bucket_order(Qn, nb, ncat)
    bo_o = oracle.precompute_bucket_order_multilevel(layers_from(g), internal, bucket_paths, Qn, nb, ncat)
    np.testing.assert_array_equal(bo, bo_o)
    do, no, _ = oracle.search(layers_from(g), Qn, Xs, Qs, dp, nb, k, bucket_order=bo_o)
    np.testing.assert_array_equal(nns, no)
    np.testing.assert_array_equal(dists, do)
    same = (bo == g["ref_bucket_order"]).all(axis=(1, 2))
    assert same.all()   # pinned: 0 rows differ on G2, G7, G8 (tests/test_oracle_multilevel.py::PINNED_ORDER_DIFFS)
    compare_modulo_near_ties(g["ref_dists"], g["ref_nns"], dists, nns)
    assert mt["inference"] > 0 and mt["seq_search"] > 0
    # lmi_search_tree (one call, scan vectors uploaded beside the walk) against lmi_nav_order + lmi_scan_topk, host and device buffers
    eng = li._engine
    slab, ent = eng.nav_order(Qn, nb)
    d_a, i_a = eng.scan_topk(Qs, slab, k)
    d_b, i_b, slab_b, ent_b = eng.search_tree(Qn, Qs, nb, k, want_order=True)
    for a, b in ((d_a, d_b), (i_a, i_b), (slab, slab_b), (ent, ent_b)):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(i_b, nns)
    dev = torch.device("cuda", 0)
    qn_t, qs_t = torch.from_numpy(np.ascontiguousarray(Qn)).to(dev), torch.from_numpy(np.ascontiguousarray(Qs)).to(dev)
    ko = eng.kout(nb, k)
    d_t = torch.empty((Qn.shape[0], ko), dtype=torch.float32, device=dev)
    i_t = torch.empty((Qn.shape[0], ko), dtype=torch.int32, device=dev)
    slab_t = torch.empty((Qn.shape[0], nb), dtype=torch.int32, device=dev)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):   # (twice: the walk's per-call state is re-initialised by the call itself)
        eng.search_tree_device(qn_t, qs_t, nb, k, d_t, i_t, None, slab_t, None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(i_t.cpu().numpy().view(np.uint32), i_b)
    np.testing.assert_array_equal(d_t.cpu().numpy(), d_b)
    np.testing.assert_array_equal(slab_t.cpu().numpy(), slab)
    eng.set_stream(0)
    if name == "G7":   # the walk in query chunks (a tree whose queues would not fit lmi_nav_order's limit)
        li._NAV_QUEUE_BYTES = 8 * 110 * 37
        d2, n2, _ = li.search(nav, Qn, srch, Qs, dp, ncat, nb, k)
        bo2, _ = li._precompute_bucket_order(Qn, nb, ncat)
        np.testing.assert_array_equal(bo2, bo)
        np.testing.assert_array_equal(n2, nns)
        np.testing.assert_array_equal(d2, dists)
    li.close()


@pytest.mark.parametrize("ncat,what", [([20, 3], "21 models: steps in batches with the count read back; queues in LDS"),
                                       ([12, 12], "13 models: every step enqueued up front; 156-entry queues stay in global memory"),
                                       ([5, 4], "small: everything up front, queues in LDS")])
def test_walk_forms_against_the_oracle(oracle, ncat, what):
    """The walk's launch forms (lmi_hip.hip nav_enqueue: NAV_ENQUEUE_ALL, NAV_LDS_CAP) on synthetic two-level trees with random models:
    bucket order and results equal the oracle's restatement of LearnedIndex.py:216-325."""
    from learnedmetricindex_amd.li.LearnedIndex import LearnedIndex
    from learnedmetricindex_amd.li.model import NeuralNetwork

    rs = np.random.RandomState(sum(ncat))
    d_nav, d_s, N, nq, nb, k = 16, 24, 4000, 300, 7, 10

    def layers_of(net):   # whatever hidden width the li model has: take the oracle's layers from the module itself
        return [(m.weight.detach().cpu().numpy().copy(), m.bias.detach().cpu().numpy().copy()) for m in net.model.layers if isinstance(m, torch.nn.Linear)]

    root = NeuralNetwork(input_dim=d_nav, output_dim=ncat[0], model_type="MLP")
    nets = {(i, -1): NeuralNetwork(input_dim=d_nav, output_dim=ncat[1], model_type="MLP") for i in range(ncat[0])}
    for net in [root] + list(nets.values()):
        with torch.no_grad():
            for m in net.model.layers:
                if isinstance(m, torch.nn.Linear):
                    m.weight.copy_(torch.from_numpy((rs.randn(*m.weight.shape) * 0.5).astype(np.float32)))
                    m.bias.copy_(torch.from_numpy((rs.randn(*m.bias.shape) * 0.1).astype(np.float32)))
    bucket_paths = [(i, j) for i in range(ncat[0]) for j in range(ncat[1])]
    dp = np.stack([rs.randint(0, ncat[0], N), rs.randint(0, ncat[1], N)], axis=1).astype(np.int64)
    dp[dp[:, 0] == 1] = (1, 0)   # node 1: one bucket with rows, its other children listed and empty
    Xn, Xs = rs.randn(N, d_nav).astype(np.float32), rs.randn(N, d_s).astype(np.float32)
    Qn, Qs = rs.randn(nq, d_nav).astype(np.float32), rs.randn(nq, d_s).astype(np.float32)
    li = LearnedIndex(root, nets, bucket_paths)
    dists, nns, _ = li.search(frame(Xn), Qn, frame(Xs), Qs, dp, ncat, nb, k)
    bo, _ = li._precompute_bucket_order(Qn, nb, ncat)
    internal = [(p, layers_of(n)) for p, n in nets.items()]
    bo_o = oracle.precompute_bucket_order_multilevel(layers_of(root), internal, bucket_paths, Qn, nb, ncat)
    np.testing.assert_array_equal(bo, bo_o)
    do, no, _ = oracle.search(layers_of(root), Qn, Xs, Qs, dp, nb, k, bucket_order=bo_o)
    np.testing.assert_array_equal(nns, no)
    np.testing.assert_array_equal(dists, do)
    li.close()


def test_resident_cache_key_and_query_chunks(oracle):
    """The HBM-resident copy is keyed by a content fingerprint: an in-place edit of a sampled row, new labels or
    `invalidate()` rebuild it, an equal-content copy of `data_prediction` does not; and a search answered in
    several query chunks (small workspace budget) equals the one-call answer."""
    g = load_golden("G1")
    Xn, Qn, Xs, Qs = inputs_for("G1", g)
    li, _ = make_index("G1", g)
    dp = g["data_prediction"].astype(np.int64)
    df = frame(Xs.copy())
    d, n, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10)
    eng = li._engine
    d1, n1, _ = li.search(df, Qn, df, Qs, dp.copy(), [12], 3, 10)      # same contents, another array: cache hit
    assert li._engine is eng
    np.testing.assert_array_equal(n1, n)
    li._WORKSPACE_BYTES = eng.workspace_bytes(37, 3)                      # ~37 queries per chunk (lmi_workspace_bytes)
    assert eng.workspace_bytes(200, 3) > li._WORKSPACE_BYTES > eng.workspace_bytes(0, 3)
    d2, n2, mt = li.search(df, Qn, df, Qs, dp, [12], 3, 10)
    np.testing.assert_array_equal(n2, n)
    np.testing.assert_array_equal(d2, d)
    assert mt["seq_search"] > 0 and mt["inference"] > 0
    df.iloc[0, :] = Qs[0]                                                  # in-place edit of a sampled row
    d3, n3, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10)
    assert li._engine is not eng
    X2 = Xs.copy()
    X2[0] = Qs[0]
    do, no, _ = oracle.search(layers_from(g), Qn, X2, Qs, dp, 3, 10, nthreads=4)
    np.testing.assert_array_equal(n3, no)
    np.testing.assert_array_equal(d3, do)
    eng = li._engine
    li.invalidate()
    assert li._engine is None
    d4, n4, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10)
    np.testing.assert_array_equal(n4, no)
    # the key is order-sensitive and covers every weight -- a placement permuted between two objects of different
    # buckets, or a changed LAST-layer weight, misses the cache (a sum of samples / the first layer alone did not see either)
    eng = li._engine
    i, j = 0, int(np.flatnonzero(dp[:, 0] != dp[0, 0])[0])
    dp_sw = dp.copy()
    dp_sw[[i, j]] = dp_sw[[j, i]]
    li.search(df, Qn, df, Qs, dp_sw, [12], 3, 10)
    assert li._engine is not eng
    eng = li._engine
    last = [m for m in li.root_model.model.layers if isinstance(m, torch.nn.Linear)][-1]
    with torch.no_grad():
        last.bias[0] += 0.5
    li.root_model._engine = None
    li.search(df, Qn, df, Qs, dp_sw, [12], 3, 10)
    assert li._engine is not eng
    li.close()


def test_in_place_edit_of_any_row_is_never_answered_from_the_stale_copy(oracle):
    """the reference re-reads its frames on every call (LearnedIndex.py:350-357).  By default the resident copy is
    keyed on EVERY byte of the scan frame: an in-place edit of a row no sample would see rebuilds it; a frame with equal content
    but another identity / memory layout does not; `assume_unchanged=True` and `strict_cache=False` are the documented opt-outs."""
    g = load_golden("G1")
    Xn, Qn, Xs, Qs = inputs_for("G1", g)
    li, _ = make_index("G1", g)
    dp = g["data_prediction"].astype(np.int64)
    df = frame(Xs.copy())
    d, n, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10)
    eng = li._engine
    li.search(df.copy(), Qn, df[list(df.columns)], Qs, dp, [12], 3, 10)      # fresh objects, equal content: no rebuild
    assert li._engine is eng
    row = int(np.setdiff1d(np.arange(Xs.shape[0]), np.unique(np.linspace(0, Xs.shape[0] - 1, num=4096, dtype=np.int64)))[100])
    bo = oracle.precompute_bucket_order(layers_from(g), Qn, 3, nthreads=4)[:, :, 0]
    qv = int(np.flatnonzero(bo[:, 0] == dp[row, 0])[0])                       # a query whose first bucket holds that row
    df.iloc[row, :] = Qs[qv]                                                  # a row the sampled fingerprint does not look at
    d_stale, n_stale, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10, assume_unchanged=True)
    assert li._engine is eng                                                  # the caller vouched: answered from the resident copy
    np.testing.assert_array_equal(n_stale, n)
    d2, n2, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10)                    # default: every byte fingerprinted -> rebuilt
    assert li._engine is not eng
    X2 = Xs.copy()
    X2[row] = Qs[qv]
    do, no, _ = oracle.search(layers_from(g), Qn, X2, Qs, dp, 3, 10, nthreads=4)
    np.testing.assert_array_equal(n2, no)
    np.testing.assert_array_equal(d2, do)
    assert n2[qv, 0] == df.index[row] and n[qv, 0] != df.index[row]          # (that query now finds its own copy first)
    # the sampled mode (what frames above 1 GiB get unless strict_cache is True) does not see such an edit: documented, opt-in
    li.strict_cache = False
    li.search(df, Qn, df, Qs, dp, [12], 3, 10)
    eng = li._engine
    df.iloc[row, :] = Xs[row]
    li.search(df, Qn, df, Qs, dp, [12], 3, 10)
    assert li._engine is eng
    li.strict_cache = True
    d3, n3, _ = li.search(df, Qn, df, Qs, dp, [12], 3, 10)
    assert li._engine is not eng
    np.testing.assert_array_equal(n3, n)
    li.close()
