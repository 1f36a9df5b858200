"""GPU (`-m gpu`): front_kernel (lmi_front.h: routing + query norms / fp16 packing / bounds in one launch) against the
separate preparation kernels it replaces (LMI_FRONT=0) and the all-f32 scan, bit for bit, over the shapes that steer its code
paths: every lanes-per-row group (d = 45 / 100 / 200 / 768 / 1 536: two k-slices), ragged and repeated bucket orders, invalid
and empty buckets, several parts per bucket, more columns than one positioning window, the L2 metric, k != 10."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from learnedmetricindex_amd import _capi

    _capi.lib()
    return _capi


def run(capi, X, labels, L, Q, order, k, front, prefilter=True, metric=None, chunk_rows=256):
    old = os.environ.get("LMI_FRONT")
    os.environ["LMI_FRONT"] = "1" if front else "0"
    try:
        idx = capi.Index(0, chunk_rows=chunk_rows, prefilter=prefilter, **({} if metric is None else {"metric": metric}))
    finally:
        if old is None:
            os.environ.pop("LMI_FRONT", None)
        else:
            os.environ["LMI_FRONT"] = old
    idx.set_buckets(X, labels, L)
    d, i = idx.scan_topk(Q, order, k)
    d2, i2 = idx.scan_topk(Q, order, k)   # a second call on the same handle: nothing of the first one may linger
    np.testing.assert_array_equal(i, i2)
    np.testing.assert_array_equal(d, d2)
    tm = idx.timings()
    idx.close()
    return d, i, tm


def make(seed, N, d, L, nq, nb, empty=(), invalid_frac=0.0, repeat_frac=0.0, unit=True):
    rs = np.random.RandomState(seed)
    cent = rs.randn(L, d).astype(np.float32)
    lab = rs.randint(0, L, N).astype(np.int64)
    for e in empty:
        lab[lab == e] = (e + 1) % L
    X = cent[lab] + 0.7 * rs.randn(N, d).astype(np.float32)
    Q = cent[rs.randint(0, L, nq)] + 0.7 * rs.randn(nq, d).astype(np.float32)
    if unit:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    order = np.argsort(-(Q @ cent.T), axis=1)[:, :nb].astype(np.int32)
    if invalid_frac:
        bad = rs.rand(nq, nb) < invalid_frac
        order[bad] = rs.choice([-1, L, L + 7], size=int(bad.sum()))
    if repeat_frac and nb > 1:
        rep = rs.rand(nq) < repeat_frac
        order[rep, 1] = order[rep, 0]   # caller data may repeat a bucket in a row
    return X, lab, Q, order


SHAPES = [
    # (seed, N, d, L, nq, nb, k, kwargs)
    (1, 20_000, 45, 16, 700, 4, 10, {}),
    (2, 20_000, 100, 12, 500, 3, 10, dict(invalid_frac=0.05)),
    (3, 12_000, 200, 9, 300, 4, 10, dict(empty=(2, 5), repeat_frac=0.2)),
    (4, 30_000, 768, 24, 1000, 4, 10, {}),
    (5, 6_000, 1536, 6, 200, 2, 7, {}),
    (6, 40_000, 64, 3, 5000, 2, 10, {}),            # 3 300 columns per bucket: parts x several positioning windows
    (7, 9_000, 33, 40, 257, 8, 15, dict(unit=False, invalid_frac=0.02)),
    (8, 5_000, 128, 5, 64, 1, 10, {}),              # a single rank: returned unmerged
    (9, 15_000, 96, 300, 900, 5, 10, dict(empty=(0, 299))),
    (10, 30_000, 32, 2, 12_000, 2, 10, {}),         # 12 000 slots per bucket: a wave's LDS list of its bucket's slots runs over -> the placing walk
]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"s{s[0]}_d{s[2]}_L{s[3]}_nb{s[5]}" for s in SHAPES])
def test_front_kernel_equals_separate_kernels_and_exact(capi, shape):
    seed, N, d, L, nq, nb, k, kw = shape
    X, lab, Q, order = make(seed, N, d, L, nq, nb, **kw)
    d1, i1, tm1 = run(capi, X, lab, L, Q, order, k, front=True)
    d0, i0, tm0 = run(capi, X, lab, L, Q, order, k, front=False)
    de, ie, _ = run(capi, X, lab, L, Q, order, k, front=True, prefilter=False)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    np.testing.assert_array_equal(i1, ie)
    np.testing.assert_array_equal(d1, de)
    # device stamps: every phase of the prefilter path was stamped, the parts add up, nothing is negative or absurd
    from learnedmetricindex_amd import _capi as C
    for tm in (tm1, tm0):
        assert tm[C.T_TOTAL] > 0 and tm[C.T_TOTAL] < 1000
        assert tm[C.T_PF_EMIT] > 0 and tm[C.T_PF_SAMPLE] > 0 and tm[C.T_ROUTE] > 0 and tm[C.T_RESCORE] > 0
        parts = tm[C.T_ROUTE] + tm[C.T_SCAN] + tm[C.T_MERGE]
        assert abs(parts - tm[C.T_TOTAL]) <= 0.02 * tm[C.T_TOTAL] + 0.002, (parts, tm)


def test_front_kernel_l2_metric(capi):
    X, lab, Q, order = make(11, 16_000, 60, 10, 400, 3, unit=False)
    d1, i1, _ = run(capi, X, lab, 10, Q, order, 10, front=True, metric="l2")
    d0, i0, _ = run(capi, X, lab, 10, Q, order, 10, front=False, metric="l2")
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)


def test_hipevent_timing_level_agrees_with_device_stamps(capi):
    """Level 3 (hipEvents between the kernels) and level 2 (device stamps) time the same phases: the dominant kernel's
    duration agrees within the events' own bubbles."""
    X, lab, Q, order = make(12, 400_000, 128, 16, 4000, 4)
    idx = capi.Index(0, chunk_rows=2048)
    idx.set_buckets(X, lab, 16)
    out = {}
    for level in (2, 3):
        idx.set_timing(level)
        for _ in range(3):
            idx.scan_topk(Q, order, 10)
        idx.timings_reset()
        for _ in range(5):
            d, i = idx.scan_topk(Q, order, 10)
        out[level] = idx.timings_mean()[0]
    idx.close()
    from learnedmetricindex_amd import _capi as C
    a, b = out[2][C.T_PF_EMIT], out[3][C.T_PF_EMIT]
    assert a > 0 and b > 0 and abs(a - b) <= 0.15 * max(a, b) + 0.02, (a, b)
