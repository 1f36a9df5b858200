"""faiss k-means, seed 2023 (reference: li/clustering/faiss_kmeans.py:8-24).  Requires faiss."""
from typing import Any, Dict, Optional

import numpy as np


def cluster(data, n_clusters: int, parameters: Optional[Dict[str, Any]]):
    from faiss import Kmeans  # not installed in the MI355X image: use "scikit_kmeans" there

    params = {"verbose": False, "seed": 2023} if parameters is None else parameters
    data = np.ascontiguousarray(data, dtype=np.float32)
    km = Kmeans(d=data.shape[1], k=n_clusters, **params)
    km.train(data)
    labels = km.index.search(data, 1)[1][:, 0]
    return km, labels.astype(np.int32)
