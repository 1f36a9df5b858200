"""GPU (`-m gpu`): the fp16 prefilter + exact re-rank against the all-f32 scan and the oracle.

The two modes must agree bit for bit on everything -- including inputs built to stress the
candidate logic: clouds of near-duplicates inside the 2-eps window (forces the exact fallback),
un-normalised vectors with norms spread over four orders of magnitude, tiny and empty buckets."""
import numpy as np
import pytest

from helpers import inputs_for, layers_from, load_golden

pytestmark = pytest.mark.gpu


def both_modes(capi, X, labels, L, Q, order, k=10, chunk_rows=256):
    out = []
    for pf in (True, False):
        idx = capi.Index(0, chunk_rows=chunk_rows, prefilter=pf)
        idx.set_buckets(X, labels, L)
        d, i = idx.scan_topk(Q, order, k)
        active, survivors, fallbacks = idx.prefilter_stats()
        assert active == pf
        out.append((d, i, survivors, fallbacks))
        idx.close()
    return out


@pytest.fixture(scope="module")
def capi():
    from learnedmetricindex_amd import _capi

    _capi.lib()
    return _capi


@pytest.mark.parametrize("name", ["G3", "G4", "G5"])
def test_modes_agree_on_fixtures(capi, oracle, name):
    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    layers = layers_from(g)
    dp = g["data_prediction"]
    L = layers[-1][0].shape[0]
    nb = int(g["n_buckets"])
    order = oracle.rank_classes(oracle.forward_logits(layers, Qn, nthreads=4), nb)
    (d1, i1, sv, fb), (d0, i0, _, _) = both_modes(capi, Xs, dp[:, 0], L, Qs, order, chunk_rows=2048)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    # other chunking: different items, same answers
    (d2, i2, _, fb2), _ = both_modes(capi, Xs, dp[:, 0], L, Qs, order, chunk_rows=256)
    np.testing.assert_array_equal(i2, i0)
    np.testing.assert_array_equal(d2, d0)
    do, io, _ = oracle.search(layers, Qn, Xs, Qs, dp, nb, 10, nthreads=4)
    np.testing.assert_array_equal(i1, io)
    np.testing.assert_array_equal(d1.astype(np.float64), do)
    assert fb == 0 and sv >= 10  # ordinary data never needs the fallback
    slots = order.size
    assert sv / slots < 40, f"{sv / slots:.1f} survivors per slot: the bound is looser than expected"


def test_near_duplicate_cloud_forces_exact_fallback(capi, oracle):
    """300 vectors within 1e-5 of each other, all near the query: far more than PF_KEEP survivors
    inside the 2-eps window -> the slot must fall back to the exact kernel and still be right."""
    rs = np.random.RandomState(4)
    d = 96
    base = rs.randn(d).astype(np.float32)
    base /= np.linalg.norm(base)
    X = rs.randn(6000, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    cloud = base[None, :] + 1e-5 * rs.randn(300, d).astype(np.float32)
    X[1000:1300] = cloud / np.linalg.norm(cloud, axis=1, keepdims=True)
    labels = np.zeros(6000, dtype=np.int64)
    labels[3000:] = 1
    Q = np.stack([base, -base, X[4000]]).astype(np.float32)
    order = np.array([[0, 1], [0, 1], [1, 0]], dtype=np.int32)
    (d1, i1, sv, fb), (d0, i0, _, _) = both_modes(capi, X, labels, 2, Q, order)
    assert fb >= 1
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    D, I = oracle.knn_ip(Q[:1], X[:3000], 10)
    np.testing.assert_array_equal(i1[0], (I[0] + 1).astype(np.uint32))  # ids are 1-based rows


def test_unnormalised_wide_dynamic_range(capi, oracle):
    """Norms from 1e-2 to 1e2 (search.py:198-210: the scan vectors are not normalised when
    kind != clip768v2): the fp16 copy uses one scale for the whole index, the bound stays valid."""
    rs = np.random.RandomState(8)
    N, d, L = 20000, 45, 6
    X = rs.randn(N, d).astype(np.float32) * (10.0 ** rs.uniform(-2, 2, size=(N, 1))).astype(np.float32)
    Q = rs.randn(64, d).astype(np.float32) * (10.0 ** rs.uniform(-1, 1, size=(64, 1))).astype(np.float32)
    labels = rs.randint(0, L, size=N)
    labels[labels == 2] = 3                       # an empty bucket
    labels[np.flatnonzero(labels == 4)[7:]] = 5   # a 7-vector bucket
    order = np.stack([rs.permutation(L)[:4] for _ in range(64)]).astype(np.int32)
    (d1, i1, sv, fb), (d0, i0, _, _) = both_modes(capi, X, labels, L, Q, order)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    # spot-check against the oracle's knn for one (query, bucket) with a populated bucket
    order[0, 0] = 5
    rows = np.flatnonzero(labels == order[0, 0])
    D, I = oracle.knn_ip(Q[:1], X[rows], 10)
    idx = capi.Index(0, chunk_rows=256)
    idx.set_buckets(X, labels, L)
    dd, ii = idx.scan_topk(Q[:1], order[:1, :1], 10)
    np.testing.assert_array_equal(ii[0], (rows[I[0]] + 1).astype(np.uint32))
    np.testing.assert_array_equal(dd[0], np.float32(1) - D[0])
    idx.close()


def test_wide_vectors_beyond_lds_query_staging(capi, oracle):
    """d = 1100 > the 1 024 floats the re-rank kernel stages in LDS: the query is read from L2."""
    rs = np.random.RandomState(15)
    X = rs.randn(3000, 1100).astype(np.float32)
    Q = rs.randn(5, 1100).astype(np.float32)
    labels = rs.randint(0, 3, size=3000)
    order = np.array([[0, 1], [1, 2], [2, 0], [0, 2], [1, 0]], dtype=np.int32)
    (d1, i1, sv, fb), (d0, i0, _, _) = both_modes(capi, X, labels, 3, Q, order, chunk_rows=2048)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    rows = np.flatnonzero(labels == 0)
    D, I = oracle.knn_ip(Q[:1], X[rows], 10)
    idx = capi.Index(0)
    idx.set_buckets(X, labels, 3)
    dd, ii = idx.scan_topk(Q[:1], order[:1, :1], 10)
    np.testing.assert_array_equal(ii[0], (rows[I[0]] + 1).astype(np.uint32))
    np.testing.assert_array_equal(dd[0], np.float32(1) - D[0])
    idx.close()


@pytest.mark.parametrize("d", [96, 80])
def test_query_tile_shapes(capi, oracle, d):
    """Every split of a bucket's queries over 256-query tiles and the two wave groups of a block: buckets that
    receive 1 ... 600 queries (1-8 col-blocks per tile, uneven groups, an idle group, several tiles) and hold
    a ragged number of rows (several 2048-row chunks, a partial last tile); d = 96 is 3 stages of 32 k,
    d = 80 pads a stage.  Prefilter and exact mode must agree bit for bit; one bucket is checked against the oracle."""
    rs = np.random.RandomState(21)
    m_per_bucket = [1, 31, 33, 64, 65, 96, 127, 129, 160, 200, 255, 257, 300, 385, 513, 600]
    L = len(m_per_bucket)
    sizes = [rs.randint(300, 5000) for _ in range(L)]
    sizes[3], sizes[7] = 4096 + 7, 37
    labels = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)])
    rs.shuffle(labels)
    X = rs.randn(labels.size, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    order = np.concatenate([np.full(m, b) for b, m in enumerate(m_per_bucket)]).astype(np.int32)
    rs.shuffle(order)
    order = order[:, None]
    Q = rs.randn(order.shape[0], d).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    (d1, i1, sv, fb), (d0, i0, _, _) = both_modes(capi, X, labels, L, Q, order, chunk_rows=2048)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    b = 12
    rows = np.flatnonzero(labels == b)
    qsel = np.flatnonzero(order[:, 0] == b)
    D, I = oracle.knn_ip(Q[qsel], X[rows], 10)
    np.testing.assert_array_equal(i1[qsel], (rows[I] + 1).astype(np.uint32))
    np.testing.assert_array_equal(d1[qsel], np.float32(1) - D)


@pytest.mark.parametrize("dup,d", [(100, 64), (1000, 64), (1000, 160)])
def test_duplicate_heavy_buckets_recover_without_fallback(capi, oracle, dup, d):
    """Every vector copied 100 / 1 000 times (1e-7 apart): far more than PF_CAP rows pass a column's sampled threshold, and at
    1 000 copies far more than PF_CAP lie within 2 eps' of a slot's top ten.  What does not fit a column's buffer goes to the
    shared overflow log, is sorted by column behind pass 2 and re-scored from there (fallback_kernel's candidate path; d = 64:
    the low-dimensional kernels, d = 160: pass2_kernel): results as always identical to the all-f32 scan."""
    rs = np.random.RandomState(21)
    L, U, nq, nb = 4, 120_000 // dup, 160, 2
    base = rs.randn(U, d).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    X = np.repeat(base, dup, axis=0) + (1e-7 * rs.randn(U * dup, d)).astype(np.float32)
    labels = rs.randint(0, L, size=U).repeat(dup).astype(np.int64)
    Q = base[rs.choice(U, nq)] + 0.05 * rs.randn(nq, d).astype(np.float32)
    Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    order = np.stack([rs.permutation(L)[:nb] for _ in range(nq)]).astype(np.int32)
    (d1, i1, sv, fb), (d0, i0, _, _) = both_modes(capi, X, labels, L, Q, order, chunk_rows=2048)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    # and no slot scanned its whole bucket: the flagged ones were re-scored from their (complete) candidates
    idx = capi.Index(0, chunk_rows=2048)
    idx.set_buckets(X, labels, L)
    d2, i2 = idx.scan_topk(Q, order, 10)
    st = idx.debug_peek("pf_fallback", 32).view(np.uint32)
    idx.close()
    np.testing.assert_array_equal(i2, i0)
    assert st[1] == 0 and st[2] == 0, "the overflow log ran full"
    assert st[5] == 0, f"{st[5]} of {nq * nb} slots brute-forced their bucket ({st[0]} flagged, {st[3]} overflow entries)"
    if dup == 1000:
        assert st[3] > 0 and st[0] > nq   # thousands of rows within 2 eps' of a slot's top ten: past the column buffers, into the log


@pytest.mark.parametrize("d", [5, 29, 45, 64, 77, 96, 109, 128])
def test_low_dimensional_form_every_group_count(capi, oracle, d):
    """d <= 128 runs both prefilter passes in their low-dimensional form (lmi_pass2_small.h: 1 ... 8 k16-groups per row-block,
    K padded to whole groups).  Buckets of 0, 3 and 9 rows (fewer than ten: no bound), an odd number of row-blocks, a ragged
    last row-block, more than 384 queries on one bucket (two query tiles), several items per bucket (256-row chunks) and one
    (auto chunks); top-3 routing so that the query-level bound is in play.  Prefilter, exact mode and the oracle must agree
    bit for bit."""
    rs = np.random.RandomState(100 + d)
    sizes = [0, 3, 9, 33, 64 + 31, 700, 1500, 2048 + 65, 5000]
    L = len(sizes)
    labels = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)]).astype(np.int64)
    rs.shuffle(labels)
    centres = rs.randn(L, d).astype(np.float32)
    X = centres[labels] + rs.randn(labels.size, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    nq, nb = 900, 3
    qc = rs.randint(0, L, nq)
    qc[:450] = 8   # 450 queries whose first bucket is the largest: two query tiles
    Q = centres[qc] + rs.randn(nq, d).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    order = np.stack([np.concatenate([[qc[i]], rs.permutation(np.delete(np.arange(L), qc[i]))[: nb - 1]]) for i in range(nq)]).astype(np.int32)
    res = {}
    for chunk_rows in (256, None):
        for pf in (True, False):
            idx = capi.Index(0, chunk_rows=chunk_rows, prefilter=pf)
            idx.set_buckets(X, labels, L)
            res[(chunk_rows, pf)] = idx.scan_topk(Q, order, 10)
            if pf:
                active, survivors, fallbacks = idx.prefilter_stats()
                assert active and fallbacks == 0
            idx.close()
    d0, i0 = res[(256, False)]
    for key, (dd, ii) in res.items():
        np.testing.assert_array_equal(ii, i0, err_msg=str(key))
        np.testing.assert_array_equal(dd, d0, err_msg=str(key))
    # the oracle on the largest bucket's first-rank queries: rank lists merged by the library, so compare through knn on the union
    rows = np.flatnonzero(np.isin(labels, order[0]))
    D, I = oracle.knn_ip(Q[:1], X[rows], 10)
    np.testing.assert_array_equal(i0[:1], (rows[I] + 1).astype(np.uint32))
    np.testing.assert_array_equal(d0[:1], np.float32(1) - D)


@pytest.mark.parametrize("d", [160, 256])
def test_narrow_query_tiles_in_long_chunks(capi, oracle, d):
    """Most tiles of a real batch are narrow (a bucket receives 0 ... 4 235 queries, median 220, at C2): buckets that receive
    1 ... 192 queries (1 ... 6 col-blocks) and the first wider ones, bucket sizes around the 256 / 512-row boundaries (one
    256-row tile, two, two and a ragged row-block, several + a short rest), 2048-row and 512-row chunks, d = 160 (5 stages: a dead
    ring step per tile) and 256.  Prefilter == exact mode bit for bit; two buckets against the oracle.  (Written for the
    512-vector tile of profiles/r03_pass2_wide_variant.h.txt, which passed it; kept for the shapes.)"""
    rs = np.random.RandomState(300 + d)
    m_per_bucket = [1, 31, 33, 64, 65, 96, 97, 128, 129, 160, 161, 192, 193, 224, 300, 3]
    sizes = [700, 256, 257, 512, 513, 1000, 2048, 2049, 2048 + 300, 4096 + 33, 511, 1025, 3000, 777, 1500, 5000]
    L = len(m_per_bucket)
    labels = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)])
    rs.shuffle(labels)
    centres = rs.randn(L, d).astype(np.float32)
    X = centres[labels] + rs.randn(labels.size, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    first = np.concatenate([np.full(m, b) for b, m in enumerate(m_per_bucket)]).astype(np.int32)
    rs.shuffle(first)
    nq = first.size
    second = (first + 1 + rs.randint(0, L - 1, nq)) % L      # a second, different bucket: the query-level bound is in play
    order = np.stack([first, second], axis=1).astype(np.int32)
    Q = centres[first] + rs.randn(nq, d).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    ref = None
    for chunk_rows in (2048, 512):
        (d1, i1, sv, fb), (d0, i0, _, _) = both_modes(capi, X, labels, L, Q, order, chunk_rows=chunk_rows)
        np.testing.assert_array_equal(i1, i0)
        np.testing.assert_array_equal(d1, d0)
        assert fb == 0
        if ref is None:
            ref = (d0, i0)
        np.testing.assert_array_equal(i0, ref[1])
        np.testing.assert_array_equal(d0, ref[0])
    # one-bucket routing of two narrow buckets against the oracle
    for b in (9, 12):
        qsel = np.flatnonzero(first == b)
        idx = capi.Index(0, chunk_rows=2048, prefilter=True)
        idx.set_buckets(X, labels, L)
        dd, ii = idx.scan_topk(Q[qsel], np.full((qsel.size, 1), b, dtype=np.int32), 10)
        idx.close()
        rows = np.flatnonzero(labels == b)
        D, I = oracle.knn_ip(Q[qsel], X[rows], 10)
        np.testing.assert_array_equal(ii, (rows[I] + 1).astype(np.uint32))
        np.testing.assert_array_equal(dd, np.float32(1) - D)


@pytest.mark.parametrize("d,nb", [(1100, 2), (1536, 1), (3000, 3), (9000, 2), (2048, 4)])
def test_wide_rows_take_the_streamed_rerank(capi, oracle, d, nb):
    """Rows wider than four waves' re-rank buffers (d > 1 126: two waves per block up to ~8 700, one beyond; the small form one wave
    per block from ~2 000) keep select_kernel + rescore_kernel (round 4; before: select_rescore_kernel).  A cluster of 120 near-copies
    puts groups on the `big` list (more survivors than the small form holds).  Prefilter, exact mode and the oracle agree bit for bit."""
    rs = np.random.RandomState(d)
    L, nq = 5, 160
    sizes = [900, 33, 1500, 700, 1200]
    labels = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)]).astype(np.int64)
    rs.shuffle(labels)
    centres = rs.randn(L, d).astype(np.float32)
    X = centres[labels] + rs.randn(labels.size, d).astype(np.float32)
    dup = np.flatnonzero(labels == 2)[:120]
    X[dup] = X[dup[0]] + 1e-4 * rs.randn(120, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    qc = rs.randint(0, L, nq)
    Q = centres[qc] + rs.randn(nq, d).astype(np.float32)
    Q[:20] = X[dup[0]] + 0.05 * rs.randn(20, d).astype(np.float32)   # queries next to the cluster of copies
    qc[:20] = 2
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    order = np.stack([np.concatenate([[qc[i]], rs.permutation(np.delete(np.arange(L), qc[i]))[: nb - 1]]) for i in range(nq)]).astype(np.int32)
    (d1, i1, sv, fb), (d0, i0, _, _) = both_modes(capi, X, labels, L, Q, order)
    assert fb == 0
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    for q in (0, 5, 40, nq - 1):
        rows = np.flatnonzero(np.isin(labels, order[q]))
        D, I = oracle.knn_ip(Q[q:q + 1], X[rows], 10)
        np.testing.assert_array_equal(i1[q:q + 1], (rows[I] + 1).astype(np.uint32))
        np.testing.assert_array_equal(d1[q:q + 1], np.float32(1) - D)


@pytest.mark.parametrize("L", [1500, 7000, 10000, 20011])
def test_many_buckets_route_and_scan(capi, oracle, L):
    """Thousands of leaf buckets (the routing kernels sort the buckets by work in LDS: a bitonic sort padded to 2 048 / 8 192 keys;
    round 3's form needed 24 bytes of LDS per bucket and could not be launched past ~6 800 buckets although 8 000 were accepted;
    past 8 000 -- fan-outs like [100, 100] -- the same sort runs in a global scratch buffer):
    many empty buckets, many with fewer than ten rows, queries spread over all of them.  Prefilter == exact == oracle."""
    rs = np.random.RandomState(L)
    d, nq, nb = 24, 600, 3
    N = 6 * L
    labels = rs.randint(0, L, N).astype(np.int64)
    labels[: N // 10] = rs.randint(0, 8, N // 10)          # a few heavy buckets
    centres = rs.randn(L, d).astype(np.float32)
    X = centres[labels] + 0.3 * rs.randn(N, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    qc = rs.randint(0, L, nq)
    Q = centres[qc] + 0.3 * rs.randn(nq, d).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    order = np.stack([np.concatenate([[qc[i]], rs.choice(np.delete(np.arange(16), qc[i]) if qc[i] < 16 else np.arange(16), nb - 1, replace=False)])
                      for i in range(nq)]).astype(np.int32)
    (d1, i1, sv, fb), (d0, i0, _, _) = both_modes(capi, X, labels, L, Q, order)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    for q in (0, 17, nq - 1):
        rows = np.flatnonzero(np.isin(labels, order[q]))
        if rows.size >= 10:
            D, I = oracle.knn_ip(Q[q:q + 1], X[rows], 10)
            np.testing.assert_array_equal(i1[q:q + 1], (rows[I] + 1).astype(np.uint32))
            np.testing.assert_array_equal(d1[q:q + 1], np.float32(1) - D)
