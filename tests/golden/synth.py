"""Seeded synthetic inputs shared by tests/golden/make_golden.py, the tests and bench.py.

Follows SURVEY.md section 8d: a mixture of `n_centres` Gaussian clusters, rows L2-normalised
(`sklearn.preprocessing.normalize` semantics, reference search.py:142-144), float32, drawn from
`numpy.random.RandomState(seed)` -- the legacy generator whose streams are frozen across numpy
versions, so fixtures only need to store the seed (plus a checksum, verified by the tests).
"""
import numpy as np


def l2_normalize(x: np.ndarray) -> np.ndarray:
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
    n[n == 0] = 1.0
    return (x / n).astype(np.float32)


def mixture(seed: int, n: int, d: int, n_centres: int, nq: int, spread: float = 1.0,
            normalize: bool = True):
    """Returns (X f32[n,d], Q f32[nq,d]).  Queries are fresh draws, not members of X."""
    rs = np.random.RandomState(seed)
    centres = rs.randn(n_centres, d).astype(np.float32)
    X = centres[rs.randint(n_centres, size=n)] + np.float32(spread) * rs.randn(n, d).astype(np.float32)
    Q = centres[rs.randint(n_centres, size=nq)] + np.float32(spread) * rs.randn(nq, d).astype(np.float32)
    if normalize:
        X, Q = l2_normalize(X), l2_normalize(Q)
    return np.ascontiguousarray(X, dtype=np.float32), np.ascontiguousarray(Q, dtype=np.float32)


def checksum(a: np.ndarray) -> np.ndarray:
    """Order-sensitive fingerprint: [sum, sum of i-weighted elements mod prime stride, first, last]."""
    f = a.astype(np.float64).ravel()
    w = (np.arange(f.size) % 8191 + 1).astype(np.float64)
    return np.array([f.sum(), (f * w).sum(), f[0], f[-1]], dtype=np.float64)
