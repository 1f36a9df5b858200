"""Harness-only stand-in for the absent `faiss-cpu==1.7.4` wheel (requirements.txt:8 of the
reference).  It exists so that the reference's own Python (`search/li/*.py`) can be imported in
the build container to generate golden vectors; it is never imported by the product, the tests
or the bench, and it never travels as anything but this file.

Only the two entry points the reference calls are provided, restating faiss 1.7.4's published
behaviour:

* ``knn(xq, xb, k, metric=METRIC_INNER_PRODUCT)`` (call site LearnedIndex.py:360-365):
  fp32 ``xq @ xb.T`` through BLAS sgemm (faiss' ``exhaustive_inner_product_blas`` path calls
  sgemm_ for nq >= 20), then per-row top-k by descending similarity; when ``nb < k`` the tail is
  padded with ``I = -1`` and ``D = -FLT_MAX`` (faiss' min-heap initial value for IP).
* ``Kmeans(d, k, **kw).train(x)`` / ``.index.search(x, 1)`` (clustering/faiss_kmeans.py:18-22):
  Lloyd iterations (niter=20, faiss' default), random-subset init with the given seed.

The exact fp32 summation order of the real wheel's BLAS and its in-bucket tie order are NOT
reproduced (the wheel's source is not under /root/reference): parity at this boundary is
"unpinned" and is stated as such in oracle/ and DESIGN.md.
"""
import numpy as np

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


def knn(xq, xb, k, metric=METRIC_L2):
    xq = np.ascontiguousarray(xq, dtype=np.float32)
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    nq, d = xq.shape
    nb, d2 = xb.shape
    assert d == d2
    if metric == METRIC_INNER_PRODUCT:
        sim = xq @ xb.T  # BLAS sgemm, fp32
        order = np.argsort(-sim, axis=1, kind="stable")[:, :k]
        D = np.take_along_axis(sim, order, axis=1)
        pad = np.float32(-np.finfo(np.float32).max)
    else:
        sim = (
            (xq * xq).sum(1, dtype=np.float32)[:, None]
            + (xb * xb).sum(1, dtype=np.float32)[None, :]
            - np.float32(2) * (xq @ xb.T)
        ).astype(np.float32)
        order = np.argsort(sim, axis=1, kind="stable")[:, :k]
        D = np.take_along_axis(sim, order, axis=1)
        pad = np.float32(np.finfo(np.float32).max)
    I = order.astype(np.int64)
    if nb < k:
        D = np.concatenate([D, np.full((nq, k - nb), pad, dtype=np.float32)], axis=1)
        I = np.concatenate([I, np.full((nq, k - nb), -1, dtype=np.int64)], axis=1)
    return D.astype(np.float32), I


class _FlatIndex:
    def __init__(self, centroids):
        self.centroids = centroids

    def search(self, x, k):
        x = np.ascontiguousarray(x, dtype=np.float32)
        d2 = (
            (x * x).sum(1)[:, None]
            + (self.centroids * self.centroids).sum(1)[None, :]
            - 2.0 * (x @ self.centroids.T)
        )
        I = np.argsort(d2, axis=1, kind="stable")[:, :k]
        return np.take_along_axis(d2, I, axis=1).astype(np.float32), I.astype(np.int64)


class Kmeans:
    def __init__(self, d, k, niter=20, verbose=False, seed=1234, **kw):
        self.d, self.k, self.niter, self.seed = d, k, niter, seed
        self.centroids = None
        self.index = None

    def train(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        rs = np.random.RandomState(self.seed)
        c = x[rs.choice(x.shape[0], self.k, replace=False)].copy()
        for _ in range(self.niter):
            _, lab = _FlatIndex(c).search(x, 1)
            lab = lab[:, 0]
            for j in range(self.k):
                m = lab == j
                if m.any():
                    c[j] = x[m].mean(0)
        self.centroids = c
        self.index = _FlatIndex(c)
        return 0.0
