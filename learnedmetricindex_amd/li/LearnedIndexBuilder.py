"""Index construction (reference: li/LearnedIndexBuilder.py:21-352) -- offline, out of the query
hot path, kept API-compatible: `LearnedIndexBuilder(data, config).build()`
returns `(LearnedIndex, data_prediction int64[N, n_levels], n_buckets, build_t, cluster_t)`.

Per node: k-means labels -> MLP trained until it predicts every category (the reference's rule,
:176-194) -> objects are placed by argmax MLP(x) (NOT by their k-means label, :76, :270-274).  The
training loop is torch autograd on the GPU; object placement (`NeuralNetwork.predict` over all N) runs
through the HIP MLP kernel."""
import time
from itertools import product, takewhile
from logging import DEBUG
from typing import Dict, List, Optional, Tuple

import numpy as np
import numpy.typing as npt
import pandas as pd
import torch
import torch.utils.data

from .BuildConfiguration import BuildConfiguration
from .LearnedIndex import LearnedIndex
from .Logger import Logger
from .model import LIDataset, ModelParameters, NeuralNetwork, data_X_to_torch
from .PriorityQueue import EMPTY_VALUE
from .utils import filter_path_idxs, log_runtime

MAX_TRAINING_ROUNDS = 1_000  # LearnedIndexBuilder.py:191


class LearnedIndexBuilder(Logger):
    def __init__(self, data: pd.DataFrame, config: BuildConfiguration):
        self.data = data
        self.config = config
        self.root_model: Optional[NeuralNetwork] = None
        self.internal_models: Dict[Tuple, NeuralNetwork] = {}
        self.bucket_paths: List[Tuple] = []

    # ------------------------------------------------------------------------------------------
    def build(self) -> Tuple[LearnedIndex, npt.NDArray[np.int64], int, float, float]:
        started = time.time()
        n_levels = self.config.n_levels
        data_prediction = np.full((self.data.shape[0], n_levels), EMPTY_VALUE, dtype=np.int64)
        self.logger.debug("Training the root model.")
        self.root_model, cluster_t = self._train_model(self.data, self.config.level_configurations[0])
        data_prediction[:, 0] = self.root_model.predict(data_X_to_torch(self.data))
        if n_levels == 1:
            # one bucket per predicted category (LearnedIndexBuilder.py:78-88)
            self.bucket_paths = [(i,) for i in range(len(np.unique(data_prediction[:, 0])))]
        else:
            self.logger.debug(f"Training {self.config.n_categories[:-1]} internal models.")
            cluster_t += self._train_internal_models(self.data, data_prediction, self.config)
        return self._create_index(), data_prediction, len(self.bucket_paths), time.time() - started, cluster_t

    def _create_index(self) -> LearnedIndex:
        assert self.root_model is not None, "The root model is not trained."
        return LearnedIndex(self.root_model, self.internal_models, self.bucket_paths)

    # ------------------------------------------------------------------------------------------
    @log_runtime(DEBUG, "Trained the model in: {}")
    def _train_model(self, data: pd.DataFrame, model_parameters: ModelParameters) -> Tuple[NeuralNetwork, float]:
        """One node: cluster, then train in rounds of `epochs` until every category is predicted for
        at least one object; RuntimeError after 1000 rounds (LearnedIndexBuilder.py:120-201)."""
        clustering_algorithm, model_type, epochs, lr, n_categories = model_parameters
        _, labels, cluster_t = self._cluster(data, clustering_algorithm, n_categories)
        found = len(np.unique(labels))
        if found != n_categories:
            self.logger.debug("Clustering algorithm did not return %d clusters, got %d.", n_categories, found)
            n_categories = found
        loader = torch.utils.data.DataLoader(
            dataset=LIDataset(data, labels), batch_size=256,
            sampler=torch.utils.data.SubsetRandomSampler(data.index.values.tolist()))  # 1-based labels
        everything = data_X_to_torch(data)
        model = NeuralNetwork(input_dim=data.shape[1], output_dim=n_categories, lr=lr, model_type=model_type)
        for rounds in range(1, MAX_TRAINING_ROUNDS + 2):
            if rounds > MAX_TRAINING_ROUNDS:
                raise RuntimeError("The model did not converge after 1000 iterations.")
            model.train_batch(loader, epochs=epochs, logger=self.logger)
            if len(np.unique(model.predict(everything))) == n_categories:
                break
        if rounds > 1:
            self.logger.debug(f"Trained for {rounds * epochs} epochs instead of {epochs}.")
        return model, cluster_t

    def _train_internal_models(self, data: pd.DataFrame, data_prediction: npt.NDArray[np.int64],
                               config: BuildConfiguration) -> float:
        """Levels 1..n-1, one model per internal path; fills data_prediction in place and appends
        the bucket paths of the last level (LearnedIndexBuilder.py:203-280)."""
        assert self.root_model is not None, "The root model is not trained, call `_train_root_model` first."
        cluster_t = 0.0
        for level in range(1, config.n_levels):
            self.logger.debug(f"Training level {level}.")
            for path in self._generate_internal_node_paths(level, config.n_levels, config.n_categories):
                rows = filter_path_idxs(data_prediction, path)
                assert rows.shape[0] != 0, "There are no data points associated with the given path."
                subset = data.loc[rows + 1]  # DataFrame labels are 1-based
                labels_backup = subset.index.values
                # the node's objects are re-labelled 1..m for the sampler, then restored
                model, t = self._train_model(subset.set_index(pd.Index(range(1, subset.shape[0] + 1))),
                                             config.level_configurations[level])
                self.internal_models[path] = model
                cluster_t += t
                predictions = model.predict(data_X_to_torch(subset))
                data_prediction[labels_backup - 1, level] = predictions
                if level == config.n_levels - 1:
                    self.bucket_paths.extend(path[:-1] + (i,) for i in range(len(np.unique(predictions))))
        return cluster_t

    def _cluster(self, data: pd.DataFrame, clustering_algorithm, n_clusters: int):
        """(fitted object, labels, seconds); tiny nodes get fewer clusters (LearnedIndexBuilder.py:282-304)."""
        s = time.time()
        if data.shape[0] < 2:
            return None, np.array([0] * data.shape[0]), time.time() - s
        if data.shape[0] < n_clusters:
            n_clusters = max(data.shape[0] // 5, 2)
        fitted, labels = clustering_algorithm(np.array(data), n_clusters, None)
        return fitted, labels, time.time() - s

    # ------------------------------------------------------------------------------------------
    def _serialize_path(self, path: Tuple) -> str:
        """(1, 2, -1, -1) -> "1.2" (LearnedIndexBuilder.py:306-316)."""
        return ".".join(str(v) for v in takewhile(lambda v: v != EMPTY_VALUE, path))

    def _deserialize_path(self, path: str, n_levels: int) -> Tuple:
        """"1.2", 4 -> (1, 2, -1, -1) (LearnedIndexBuilder.py:318-328)."""
        levels = [int(v) for v in path.split(".")]
        return tuple(levels + [EMPTY_VALUE] * (n_levels - len(levels)))

    def _generate_internal_node_paths(self, level: int, n_levels: int, n_categories: List[int]) -> List[Tuple]:
        """All paths of internal nodes at `level`, padded with EMPTY_VALUE (LearnedIndexBuilder.py:330-352)."""
        heads = product(*(range(n_categories[lvl]) for lvl in range(level)))
        pad = (EMPTY_VALUE,) * (n_levels - level)
        return [h + pad for h in heads]
