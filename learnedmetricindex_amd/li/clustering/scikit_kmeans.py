"""scikit-learn k-means with faiss-like defaults (reference: li/clustering/scikit_kmeans.py:8-29)."""
from typing import Any, Dict, Optional

import numpy as np
from sklearn.cluster import KMeans

_DEFAULTS = {"verbose": 0, "random_state": 2023, "init": "random", "max_iter": 25, "n_init": 1}


def cluster(data, n_clusters: int, parameters: Optional[Dict[str, Any]]):
    km = KMeans(n_clusters=n_clusters, **(dict(_DEFAULTS) if parameters is None else parameters))
    km.fit(data)
    return km, km.labels_
