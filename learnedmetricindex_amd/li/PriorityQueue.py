"""`li.PriorityQueue` of the reference API surface (reference: li/PriorityQueue.py).

The per-query priority queues of the multi-level walk live on the device in this build (csrc/lmi_mlp_fused.h:
`nav_pop_kernel` pops, `mlp_fused_kernel<FM_NAV>` pushes; entries are never moved, a popped entry is marked dead).
This module keeps the names callers import -- `EMPTY_VALUE`, the padding of paths, and a small host-side
`PriorityQueue` with the device queue's semantics (max probability first, the LATER push wins a tie: what the
reference's ascending sort + pop-from-the-tail yields), used to cross-check the kernels in tests."""
import numpy as np

EMPTY_VALUE = -1


class PriorityQueue:
    def __init__(self, n_queries: int, queue_length_upper_bound: int, n_levels: int):
        self.n_levels = n_levels
        self.probability = np.full((n_queries, queue_length_upper_bound), np.nan, dtype=np.float32)
        self.path = np.full((n_queries, queue_length_upper_bound, n_levels), EMPTY_VALUE, dtype=np.int32)
        self.alive = np.zeros((n_queries, queue_length_upper_bound), dtype=bool)
        self.length = np.zeros(n_queries, dtype=np.int32)  # entries ever pushed

    def add(self, indices, path, probabilities) -> None:
        at = self.length[indices]
        self.probability[indices, at] = probabilities
        self.path[indices, at] = path
        self.alive[indices, at] = True
        self.length[indices] = at + 1

    def pop(self, indices):
        """Paths of the most probable live entry of each queue in `indices` (ties: the latest push); EMPTY rows
        for exhausted queues."""
        indices = np.asarray(indices)
        out = np.full((indices.shape[0], self.n_levels), EMPTY_VALUE, dtype=np.int32)
        for r, q in enumerate(indices):
            live = np.flatnonzero(self.alive[q, : self.length[q]])
            if live.size:
                p = self.probability[q, live]
                best = live[np.flatnonzero(p == p.max())[-1]]
                self.alive[q, best] = False
                out[r] = self.path[q, best]
        return out

    def sort(self) -> None:
        """Nothing to do: `pop` selects by priority (kept for callers written against the reference)."""
