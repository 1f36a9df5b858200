"""Per-level build hyper-parameters (reference: li/BuildConfiguration.py:8-141).

Every argument is either one value for all levels, a one-element list, or a list with one entry
per level (`n_categories` always lists the fan-out of every level)."""
from dataclasses import dataclass, field
from typing import Any, List, Union

from .clustering import ClusteringAlgorithm
from .model import ModelParameters


def _per_level(value: Any, n_levels: int) -> List[Any]:
    """scalar | [x] | [x0..x_{n-1}]  ->  list of n_levels entries (BuildConfiguration.py:127-141)."""
    if isinstance(value, list):
        return list(value) if len(value) != 1 else value * n_levels
    return [value] * n_levels


@dataclass
class BuildConfiguration:
    clustering_algorithms: List[ClusteringAlgorithm]
    epochs: List[int]
    model_types: List[str]
    lrs: List[float]
    n_categories: List[int]
    level_configurations: List[ModelParameters] = field(init=False)
    n_levels: int = field(init=False)

    def __init__(self, clustering_algorithms: Union[List[ClusteringAlgorithm], ClusteringAlgorithm],
                 epochs: Union[List[int], int], model_types: Union[List[str], str],
                 lrs: Union[List[float], float], n_categories: List[int]):
        self._validate(clustering_algorithms, epochs, model_types, lrs, n_categories)
        n = len(n_categories)
        self.clustering_algorithms = _per_level(clustering_algorithms, n)
        self.epochs = _per_level(epochs, n)
        self.model_types = _per_level(model_types, n)
        self.lrs = _per_level(lrs, n)
        self.n_categories = n_categories
        self.n_levels = n
        self.level_configurations = [
            ModelParameters(clustering_algorithm=c, model_type=m, epochs=e, lr=lr, n_categories=k)
            for c, m, e, lr, k in zip(self.clustering_algorithms, self.model_types, self.epochs, self.lrs,
                                      self.n_categories)
        ]

    @staticmethod
    def _validate(clustering_algorithms, epochs, model_types, lrs, n_categories) -> None:
        """AssertionError on inconsistent arguments (BuildConfiguration.py:93-125)."""
        assert len(n_categories) > 0, "n_categories must specify at least one level"
        per_level = [clustering_algorithms, epochs, model_types, lrs]
        all_lists = all(isinstance(a, list) for a in per_level)
        all_scalars = (callable(clustering_algorithms) and isinstance(epochs, int)
                       and isinstance(model_types, str) and isinstance(lrs, float))
        assert all_lists or all_scalars, (
            "clustering_algorithms, epochs, model_types, and lrs must be lists or single values")
        for a in per_level:
            if isinstance(a, list):
                assert len(a) in (1, len(n_categories)), (
                    "clustering_algorithms, epochs, model_types, and lrs must "
                    "be lists of size 1 or the same size as n_categories")

    @staticmethod
    def _expand(arg, n_categories: int) -> List[Any]:
        return _per_level(arg, n_categories)
