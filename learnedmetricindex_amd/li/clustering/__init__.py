"""k-means label providers for the (offline) index build (reference: li/clustering/__init__.py:1-17).

`ClusteringAlgorithm`: (data f32[n,d], n_clusters, params|None) -> (fitted object, labels int32[n]).
`faiss_kmeans` needs the faiss wheel (absent in the MI355X image); it is imported lazily so that the
registry itself always imports."""
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np
import numpy.typing as npt

from .scikit_kmeans import cluster as scikit_kmeans

ClusteringAlgorithm = Callable[
    [npt.NDArray[np.float32], int, Optional[Dict[str, Any]]],
    Tuple[Any, npt.NDArray[np.int32]],
]


def faiss_kmeans(data, n_clusters, parameters):
    from .faiss_kmeans import cluster

    return cluster(data, n_clusters, parameters)


algorithms: Dict[str, ClusteringAlgorithm] = {
    "faiss_kmeans": faiss_kmeans,
    "scikit_kmeans": scikit_kmeans,
}
