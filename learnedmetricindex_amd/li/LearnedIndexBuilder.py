"""Index construction -- offline, out of the query hot path; API as the reference's
(li/LearnedIndexBuilder.py:21-107): `LearnedIndexBuilder(data, config).build()` returns
`(LearnedIndex, data_prediction int64[N, n_levels], n_buckets, build_t, cluster_t)`.

Structure (this build's own): the tree is grown level by level from a work-list of `_Node`s.  A node is a path
prefix plus the positions of the objects placed under it; fitting a node = cluster its vectors, train its MLP in
rounds until every category is predicted for at least one object (the reference's stopping rule, :176-194), then
place the node's objects by argmax MLP(x) -- NOT by their k-means label (:76, :270-274).  Placement of level l
produces the work-list of level l + 1.  Training is torch autograd on the GPU (mini-batches drawn without DataFrame
label games: positions, not labels, index the tensors); placement runs through the HIP MLP kernel
(`NeuralNetwork.predict`).  What callers of the reference see is unchanged: `root_model`, `internal_models`
(path padded with EMPTY_VALUE -> model, in breadth-first lexicographic order), `bucket_paths`."""
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import numpy.typing as npt
import pandas as pd
import torch
import torch.utils.data

from .BuildConfiguration import BuildConfiguration
from .LearnedIndex import LearnedIndex
from .Logger import Logger
from .model import ModelParameters, NeuralNetwork
from .PriorityQueue import EMPTY_VALUE

TRAINING_ROUND_LIMIT = 1_000   # rounds of `epochs` epochs before a node is declared not to converge (:186-194)
MINI_BATCH = 256               # the reference's DataLoader batch size (:170-174)


@dataclass
class _Node:
    prefix: Tuple[int, ...]       # categories chosen at levels 0 .. level-1; () is the root
    rows: np.ndarray              # positions (0-based) of the objects under this node

    @property
    def level(self) -> int:
        return len(self.prefix)


class _PositionBatches(torch.utils.data.Dataset):
    """(x, y) pairs of one node, indexed by position; the loader below shuffles positions each epoch."""

    def __init__(self, x: np.ndarray, y: np.ndarray):
        self.x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        self.y = torch.from_numpy(np.ascontiguousarray(y)).long()

    def __len__(self) -> int:
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i]


class LearnedIndexBuilder(Logger):
    def __init__(self, data: pd.DataFrame, config: BuildConfiguration):
        self.data = data
        self.config = config
        self.root_model: Optional[NeuralNetwork] = None
        self.internal_models: Dict[Tuple, NeuralNetwork] = {}
        self.bucket_paths: List[Tuple] = []

    # ------------------------------------------------------------------------------------------
    def build(self) -> Tuple[LearnedIndex, npt.NDArray[np.int64], int, float, float]:
        t0 = time.time()
        cfg = self.config
        depth = cfg.n_levels
        vectors = np.ascontiguousarray(self.data.to_numpy(dtype=np.float32))
        placement = np.full((vectors.shape[0], depth), EMPTY_VALUE, dtype=np.int64)
        self.internal_models, self.bucket_paths = {}, []
        clustering_seconds = 0.0
        frontier = [_Node((), np.arange(vectors.shape[0]))]
        for level in range(depth):
            params = cfg.level_configurations[level]
            self.logger.debug("level %d: %d node(s) to fit", level, len(frontier))
            next_frontier: List[_Node] = []
            for node in frontier:
                assert node.rows.size, f"node {self._dotted(node.prefix)} received no objects: nothing to train on"
                net, chosen, secs = self._fit(vectors[node.rows], params)
                clustering_seconds += secs
                placement[node.rows, level] = chosen
                if level == 0:
                    self.root_model = net
                else:
                    self.internal_models[self._padded(node.prefix, depth)] = net
                if level == depth - 1:
                    # a bucket per category the node's objects were actually placed in (:78-88, :276-278)
                    n_used = len(np.unique(chosen))
                    self.bucket_paths.extend(node.prefix + (c,) for c in range(n_used))
                else:
                    # every category of the configured fan-out becomes a node of the next level (:330-352)
                    next_frontier.extend(_Node(node.prefix + (c,), node.rows[chosen == c])
                                         for c in range(cfg.n_categories[level]))
            frontier = next_frontier
        index = LearnedIndex(self.root_model, self.internal_models, self.bucket_paths)
        return index, placement, len(self.bucket_paths), time.time() - t0, clustering_seconds

    # ------------------------------------------------------------------------------------------
    def _fit(self, x: np.ndarray, params: ModelParameters) -> Tuple[NeuralNetwork, np.ndarray, float]:
        """One node: (trained classifier, argmax category of each of its objects, clustering seconds)."""
        cluster, model_type, epochs, lr, fan_out = params
        t = time.time()
        if x.shape[0] < 2:
            labels = np.zeros(x.shape[0], dtype=np.int64)
        else:
            # a node with fewer objects than categories asks for fewer clusters (:282-304)
            want = fan_out if x.shape[0] >= fan_out else max(x.shape[0] // 5, 2)
            _, labels = cluster(x, want, None)
        secs = time.time() - t
        n_classes = len(np.unique(labels))
        if n_classes != fan_out:
            self.logger.debug("clustering produced %d of %d categories; training on %d", n_classes, fan_out, n_classes)
        batches = torch.utils.data.DataLoader(_PositionBatches(x, labels), batch_size=MINI_BATCH, shuffle=True)
        net = NeuralNetwork(input_dim=x.shape[1], output_dim=n_classes, lr=lr, model_type=model_type)
        for round_no in range(1, TRAINING_ROUND_LIMIT + 1):
            net.train_batch(batches, epochs=epochs, logger=self.logger)
            chosen = net.predict(x)
            if len(np.unique(chosen)) == n_classes:
                if round_no > 1:
                    self.logger.debug("needed %d epochs (%d rounds of %d)", round_no * epochs, round_no, epochs)
                return net, chosen, secs
        raise RuntimeError(f"a node's model still leaves categories empty after {TRAINING_ROUND_LIMIT} training rounds")

    # ---- path helpers ---------------------------------------------------------------------------
    @staticmethod
    def _padded(prefix: Tuple[int, ...], depth: int) -> Tuple[int, ...]:
        return tuple(prefix) + (EMPTY_VALUE,) * (depth - len(prefix))

    @staticmethod
    def _dotted(path: Tuple[int, ...]) -> str:
        """(1, 2, -1, -1) -> "1.2" (the reference's serialised form, :306-316)."""
        return ".".join(str(v) for v in path if v != EMPTY_VALUE) or "<root>"

    @staticmethod
    def _from_dotted(text: str, depth: int) -> Tuple[int, ...]:
        """"1.2", 4 -> (1, 2, -1, -1) (:318-328)."""
        head = tuple(int(v) for v in text.split("."))
        return head + (EMPTY_VALUE,) * (depth - len(head))
