"""GPU (`-m gpu`): the fused tail (lmi_tail.h: selection + exact re-rank + rank merge in one wave per query, LMI_TAIL=1, the default)
against the five launches it replaces (LMI_TAIL=0) and the all-f32 scan, bit for bit -- every group size (n_buckets 1..4: a query per wave, merged
in the wave; 5 / 6 / 8 / 10: groups of 1 / 3 / 4 / 2 slots per wave + merge_ranks_kernel), the
hand-overs (queries with more survivors than the small ring holds -> batches inside the kernel; slots whose candidates overflow or with
hundreds of survivors -> fallback_kernel, which then merges the query), unvisited and repeated slots, k != 10, the L2 metric."""
import os

import numpy as np
import pytest

from test_gpu_front import make

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from learnedmetricindex_amd import _capi

    _capi.lib()
    return _capi


def run(capi, X, labels, L, Q, order, k, tail, prefilter=True, metric="ip", chunk_rows=256, want_keys=False):
    old = os.environ.get("LMI_TAIL")
    os.environ["LMI_TAIL"] = "2" if tail else "0"   # 2: the fused tail also group-wise for n_buckets > 4 (off by default there: slower)
    try:
        idx = capi.Index(0, chunk_rows=chunk_rows, prefilter=prefilter, metric=metric)
    finally:
        if old is None:
            os.environ.pop("LMI_TAIL", None)
        else:
            os.environ["LMI_TAIL"] = old
    idx.set_buckets(X, labels, L)
    out = idx.scan_topk(Q, order, k, want_keys=want_keys)
    out2 = idx.scan_topk(Q, order, k, want_keys=want_keys)
    for a, b in zip(out, out2):
        np.testing.assert_array_equal(a, b)
    st = idx.prefilter_stats() if prefilter else None
    idx.close()
    return out, st


@pytest.mark.parametrize("nb,k", [(1, 10), (2, 10), (3, 7), (4, 10), (4, 15), (4, 1), (5, 10), (6, 12), (8, 10), (10, 20)])
def test_tail_equals_five_launches_every_group_size(capi, nb, k):
    X, lab, Q, order = make(20 + nb, 40_000, 96, 20, 1500, nb, invalid_frac=0.03, repeat_frac=0.1 if nb > 1 else 0.0, empty=(3,))
    (d1, i1, k1), st1 = run(capi, X, lab, 20, Q, order, k, tail=True, want_keys=True)
    (d0, i0, k0), st0 = run(capi, X, lab, 20, Q, order, k, tail=False, want_keys=True)
    (de, ie, ke), _ = run(capi, X, lab, 20, Q, order, k, tail=True, prefilter=False, want_keys=True)
    for a, b in ((d1, d0), (i1, i0), (k1, k0), (d1, de), (i1, ie), (k1, ke)):
        np.testing.assert_array_equal(a, b)
    assert st1 == st0   # same survivors re-scored, same slots flagged


def dup_data(seed, d, n_dup, n_clusters, spread):
    """Clusters of n_dup near-copies of a vector, each close to a query: many rows inside 2 eps' of a slot's top ten."""
    rs = np.random.RandomState(seed)
    N, L = 24_000, 6
    X = rs.randn(N, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    lab = (np.arange(N) % L).astype(np.int64)
    Qs = []
    for c in range(n_clusters):
        base = rs.randn(d).astype(np.float32)
        base /= np.linalg.norm(base)
        rows = np.flatnonzero(lab == (c % L))[100 + c * n_dup: 100 + (c + 1) * n_dup]
        cl = base[None, :] + spread * rs.randn(rows.size, d).astype(np.float32)
        X[rows] = cl / np.linalg.norm(cl, axis=1, keepdims=True)
        Qs.append(base)
    Q = np.concatenate([np.stack(Qs), X[rs.randint(0, N, 200)] + 0.01 * rs.randn(200, d).astype(np.float32)]).astype(np.float32)
    order = np.stack([np.roll(np.arange(L), -(i % L))[:4] for i in range(Q.shape[0])]).astype(np.int32)
    return X, lab, L, Q, order


@pytest.mark.parametrize("n_dup,spread,what", [(40, 1e-6, "big"), (300, 1e-6, "fallback: survivors"), (1500, 0.0, "fallback: overflow")])
def test_tail_hand_overs(capi, n_dup, spread, what):
    X, lab, L, Q, order = dup_data(7, 64, n_dup, 6 if n_dup < 1000 else 3, spread)
    (d1, i1), st1 = run(capi, X, lab, L, Q, order, 10, tail=True, chunk_rows=2048)
    (d0, i0), st0 = run(capi, X, lab, L, Q, order, 10, tail=False, chunk_rows=2048)
    (de, ie), _ = run(capi, X, lab, L, Q, order, 10, tail=True, prefilter=False, chunk_rows=2048)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    np.testing.assert_array_equal(i1, ie)
    np.testing.assert_array_equal(d1, de)
    assert st1 == st0
    if what.startswith("fallback"):
        assert st1[2] >= 1, "the data was meant to flag slots for fallback_kernel"


def test_tail_two_flagged_slots_of_one_query(capi):
    """A query whose FIRST TWO buckets both hold a cloud of its near-copies: two slots of one query go through fallback_kernel,
    the second to finish merges the query."""
    rs = np.random.RandomState(3)
    d, N = 48, 12_000
    X = rs.randn(N, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    lab = (np.arange(N) % 3).astype(np.int64)
    base = rs.randn(d).astype(np.float32)
    base /= np.linalg.norm(base)
    for b in (0, 1):
        rows = np.flatnonzero(lab == b)[50:50 + 400]
        X[rows] = base   # 400 exact copies in bucket 0 and in bucket 1
    Q = np.stack([base, X[7], -base]).astype(np.float32)
    order = np.array([[0, 1, 2], [1, 0, 2], [2, 1, 0]], dtype=np.int32)
    (d1, i1), st1 = run(capi, X, lab, 3, Q, order, 10, tail=True)
    (d0, i0), st0 = run(capi, X, lab, 3, Q, order, 10, tail=False)
    (de, ie), _ = run(capi, X, lab, 3, Q, order, 10, tail=True, prefilter=False)
    assert st1[2] >= 2
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    np.testing.assert_array_equal(i1, ie)
    np.testing.assert_array_equal(d1, de)


def test_tail_l2_and_raw_knn(capi, oracle):
    X, lab, Q, order = make(31, 20_000, 60, 8, 500, 3, unit=False)
    (d1, i1), _ = run(capi, X, lab, 8, Q, order, 10, tail=True, metric="l2")
    (d0, i0), _ = run(capi, X, lab, 8, Q, order, 10, tail=False, metric="l2")
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    # lmi_knn_ip: one bucket, raw similarities and rows
    D, I = capi.knn_ip(Q[:64], X[:5000], 10)
    Do, Io = oracle.knn_ip(Q[:64], X[:5000], 10)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


def test_overflow_machinery_arms_itself_after_a_batch_that_needed_it(capi):
    """The sort-by-column machinery of the overflow log (overflow_rebound_kernel + pass 2's redo launch) is NOT in the fused tail's launch
    sequence until a batch has put candidates into the log: that batch's flagged slots pick their entries out of the unsorted log, a flag in
    pinned host memory arms the machinery, and the next calls on the handle run with it.  Every call must return the all-f32 answer."""
    X, lab, L, Q, order = dup_data(9, 64, 1500, 3, 0.0)          # 1 500 exact copies near three queries: far past a column's 1 024 buffer entries
    Xn, labn, Qn, ordern = make(41, 20_000, 64, L, 300, 4)         # an ordinary batch on the same handle afterwards
    idx = capi.Index(0, chunk_rows=2048)
    idx.set_buckets(X, lab, L)
    ref = capi.Index(0, chunk_rows=2048, prefilter=False)
    ref.set_buckets(X, lab, L)
    d_ref, i_ref = ref.scan_topk(Q, order, 10)
    seen = []
    for call in range(3):                                           # call 0: unsorted log; calls 1, 2: the armed machinery (sorted log)
        d, i = idx.scan_topk(Q, order, 10)
        st = idx.debug_peek("pf_fallback", 32).view(np.uint32)
        np.testing.assert_array_equal(i, i_ref)
        np.testing.assert_array_equal(d, d_ref)
        assert st[1] == 0 and st[5] == 0 and st[3] > 0, st         # log not full, nobody scanned a whole bucket, entries were logged
        seen.append(int(st[4]))                                     # entries sorted by column: 0 on the unarmed call
    assert seen[0] == 0 and seen[1] > 0 and seen[2] > 0, seen
    ref.close()
    idx.close()
