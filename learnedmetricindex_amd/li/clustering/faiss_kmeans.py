"""faiss k-means, seed 2023 (reference: li/clustering/faiss_kmeans.py:8-24).

The MI355X image has no faiss.  Clustering is part of the offline build, not of
the search hot path, so when faiss cannot be imported the same interface is served by Lloyd's algorithm in
torch on the GPU (faiss's defaults: 20 iterations, k-means on the raw vectors, random initial centroids
from the data, seed 2023) and a warning is logged -- labels differ from faiss's, the index it leads to is
searched exactly like any other."""
import logging
from typing import Any, Dict, Optional

import numpy as np

LOG = logging.getLogger(__name__)


class TorchKmeans:
    """Minimal stand-in for faiss.Kmeans: `centroids` [k, d] float32 after `train`, `assign` for labels."""

    def __init__(self, d: int, k: int, niter: int = 20, seed: int = 2023, verbose: bool = False, **_):
        self.d, self.k, self.niter, self.seed = d, k, niter, seed
        self.centroids = None

    def train(self, data: np.ndarray) -> None:
        import torch

        dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        x = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(dev)
        g = torch.Generator().manual_seed(self.seed)
        cent = x[torch.randperm(x.shape[0], generator=g)[: self.k].to(dev)].clone()
        x2 = (x * x).sum(1)
        for _ in range(self.niter):
            lab = self._nearest(x, x2, cent)
            sums = torch.zeros_like(cent).index_add_(0, lab, x)
            cnt = torch.bincount(lab, minlength=self.k).to(x.dtype)
            keep = cnt > 0
            cent[keep] = sums[keep] / cnt[keep, None]  # an emptied cluster keeps its centroid
        self.centroids = cent.cpu().numpy()
        self._x, self._x2, self._cent = x, x2, cent

    @staticmethod
    def _nearest(x, x2, cent):
        import torch

        out = torch.empty(x.shape[0], dtype=torch.long, device=x.device)
        c2 = (cent * cent).sum(1)
        for lo in range(0, x.shape[0], 1 << 18):
            hi = min(x.shape[0], lo + (1 << 18))
            d2 = x2[lo:hi, None] - 2.0 * (x[lo:hi] @ cent.T) + c2[None, :]
            out[lo:hi] = d2.argmin(1)
        return out

    def assign(self) -> np.ndarray:
        return self._nearest(self._x, self._x2, self._cent).cpu().numpy()


def cluster(data, n_clusters: int, parameters: Optional[Dict[str, Any]]):
    params = {"verbose": False, "seed": 2023} if parameters is None else parameters
    data = np.ascontiguousarray(data, dtype=np.float32)
    try:
        from faiss import Kmeans
    except ImportError:
        LOG.warning("faiss is not installed: k-means by the torch stand-in (labels differ from faiss's)")
        km = TorchKmeans(d=data.shape[1], k=n_clusters, **params)
        km.train(data)
        return km, km.assign().astype(np.int32)
    km = Kmeans(d=data.shape[1], k=n_clusters, **params)
    km.train(data)
    labels = km.index.search(data, 1)[1][:, 0]
    return km, labels.astype(np.int32)
