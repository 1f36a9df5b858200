"""GPU (`-m gpu`): pass 2's graded work items (lmi_kernels.h RouteArrays::chunk_rb_b: the chunk a bucket's rows are handed out in is chosen per
call from the bucket's place in the work-sorted order) change nothing but the schedule -- results and prefilter statistics equal the
static-chunk form's (LMI_P2_GRADED=0) and the all-f32 scan's, bit for bit, for default and extreme level settings, uneven bucket sizes,
d <= 128 (lmi_pass2_small.h) and d > 128 (lmi_pass2.h), both preparation forms, and beyond 1 024 buckets (static chunks: the sort network)."""
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


def run(capi, X, labels, L, Q, order, k, env, prefilter=True, chunk_rows=1024):
    old = {v: os.environ.get(v) for v in env}
    os.environ.update(env)
    try:
        idx = capi.Index(0, chunk_rows=chunk_rows, prefilter=prefilter)
    finally:
        for v, o in old.items():
            if o is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = o
    idx.set_buckets(X, labels, L)
    out = idx.scan_topk(Q, order, k)
    st = idx.prefilter_stats() if prefilter else None
    idx.close()
    return out, st


def uneven(seed, N, d, L, nq, nb):
    """Bucket sizes over two orders of magnitude, queries concentrated on a few buckets: every chunk level is in use."""
    X, _, Q, order = make(seed, N, d, L, nq, nb)
    rs = np.random.RandomState(seed)
    w = rs.rand(L) ** 3 + 0.002
    lab = rs.choice(L, size=N, p=w / w.sum()).astype(np.int64)
    return X, lab, Q, order


SETTINGS = [
    {},                                                                  # the default levels
    {"LMI_P2_CHUNKS": "256,256,256"},
    {"LMI_P2_CHUNKS": "4096,512,256", "LMI_P2_CHUNK_FRAC": "0.5,0.25"},
    {"LMI_P2_CHUNKS": "512,2048,256", "LMI_P2_CHUNK_FRAC": "0.9,0.01"},   # (not monotone: still only a schedule)
]


@pytest.mark.parametrize("d,L,nb,front", [(768, 24, 4, "1"), (45, 40, 4, "1"), (200, 9, 3, "0"), (96, 300, 5, "1")])
def test_graded_items_change_only_the_schedule(capi, d, L, nb, front):
    X, lab, Q, order = uneven(31 + d, 60_000, d, L, 1500, nb)
    (d0, i0), st0 = run(capi, X, lab, L, Q, order, 10, {"LMI_P2_GRADED": "0", "LMI_FRONT": front})
    (de, ie), _ = run(capi, X, lab, L, Q, order, 10, {"LMI_FRONT": front}, prefilter=False)
    np.testing.assert_array_equal(i0, ie)
    np.testing.assert_array_equal(d0.view(np.uint32), de.view(np.uint32))
    for env in SETTINGS:
        (d1, i1), st1 = run(capi, X, lab, L, Q, order, 10, dict(env, LMI_FRONT=front))
        np.testing.assert_array_equal(i1, i0)
        np.testing.assert_array_equal(d1.view(np.uint32), d0.view(np.uint32))
        assert st1 == st0, env


def test_graded_items_beyond_the_counting_rank(capi):
    """1 500 buckets: the queue builder sorts with its network and every bucket keeps the index's static chunk."""
    X, lab, Q, order = make(5, 60_000, 64, 1500, 800, 6)
    (d0, i0), st0 = run(capi, X, lab, 1500, Q, order, 10, {"LMI_P2_GRADED": "0"}, chunk_rows=256)
    (d1, i1), st1 = run(capi, X, lab, 1500, Q, order, 10, {"LMI_P2_CHUNKS": "1024,512,256"}, chunk_rows=256)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1.view(np.uint32), d0.view(np.uint32))
    assert st1 == st0
