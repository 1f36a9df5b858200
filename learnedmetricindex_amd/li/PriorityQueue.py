"""Batched per-query priority queue (reference: li/PriorityQueue.py:1-94).

Three dense arrays hold, for every query, the probabilities and paths of the nodes still to
visit; queues are kept sorted ASCENDING by probability so that `pop` takes from the tail.
Used by multi-level navigation only (LearnedIndex.py:216-250)."""
import numpy as np
import numpy.typing as npt

EMPTY_VALUE = -1


class PriorityQueue:
    def __init__(self, n_queries: int, queue_length_upper_bound: int, n_levels: int):
        self.n_levels = n_levels
        self.probability: npt.NDArray[np.float32] = np.full(
            (n_queries, queue_length_upper_bound), EMPTY_VALUE, dtype=np.float32)
        self.path: npt.NDArray[np.int32] = np.full(
            (n_queries, queue_length_upper_bound, n_levels), EMPTY_VALUE, dtype=np.int32)
        self.length: npt.NDArray[np.int32] = np.zeros(n_queries, dtype=np.int32)
        self.should_sort: npt.NDArray[np.bool_] = np.zeros(n_queries, dtype=np.bool_)

    def add(self, indices, path, probabilities) -> None:
        """Appends one (path, probability) entry to the queues of `indices` (PriorityQueue.py:37-50)."""
        slot = self.length[indices]
        self.probability[indices, slot] = probabilities
        self.path[indices, slot, :] = path
        self.should_sort[indices] = True
        self.length[indices] = slot + 1

    def pop(self, indices):
        """Removes and returns the tail (most probable) path of each queue in `indices` (:52-56)."""
        self.length[indices] -= 1
        return self.path[indices, self.length[indices], :]

    def sort(self) -> None:
        """Re-sorts (ascending probability) every queue flagged by `add` (:58-94); queues of equal
        length are handled together, like the reference does."""
        for qlen in np.unique(self.length):
            if qlen < 2:
                continue
            rows = np.flatnonzero((self.length == qlen) & self.should_sort)
            if rows.size == 0:
                continue
            order = self.probability[rows, :qlen].argsort()
            self.probability[rows, :qlen] = np.take_along_axis(self.probability[rows, :qlen], order, axis=1)
            self.path[rows, :qlen, :] = np.take_along_axis(self.path[rows, :qlen, :], order[:, :, None], axis=1)
            self.should_sort[rows] = False
