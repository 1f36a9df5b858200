"""Exact k-NN baseline on the MI355X (the role of reference li/Baseline.py:7-21: ground truth for recall).

Cosine distance by brute force: rows are L2-normalised on the host, then ONE `lmi_knn_ip` call (the C ABI's
`faiss.knn(..., METRIC_INNER_PRODUCT)` replacement: fp16-MFMA prefilter + exact binary32 re-rank over a
single-bucket index in HBM) yields the k largest inner products per query; distance = 1 - similarity.
Same call surface as the reference class: `search(queries, data, k) -> (dists, ids 1-based, seconds)`,
`build(data) -> seconds`.  There is no CPU path here; `li.utils.pairwise_cosine` remains for callers that
want the sklearn form."""
import time

import numpy as np

from .Logger import Logger

try:
    from .. import _capi
except ImportError:  # pragma: no cover  (`li` used as a top-level package next to _capi.py)
    import _capi  # type: ignore


def _unit(x) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = np.linalg.norm(x, axis=1, keepdims=True)
    return x / np.where(n == 0, 1, n)


class Baseline(Logger):
    MAX_K = _capi.K_PER_BUCKET  # lmi_knn_ip returns at most 10 neighbours per query (LearnedIndex.py:334)

    def build(self, data) -> float:
        """Nothing to build: the scan is exhaustive."""
        self.logger.info("Baseline: exhaustive GPU scan, no index to build.")
        return 0.0

    def search(self, queries, data, k: int = 10, device: int = 0):
        if not 1 <= k <= self.MAX_K:
            raise ValueError(f"Baseline.search: k must be in [1, {self.MAX_K}]")
        t0 = time.time()
        sims, rows = _capi.knn_ip(_unit(queries), _unit(data), k=k, device=device)
        found = rows >= 0  # fewer than k objects: faiss-style padding (-1)
        dists = np.where(found, np.float32(1) - sims, np.float32(np.inf))
        return dists, np.where(found, rows + 1, 0), time.time() - t0
