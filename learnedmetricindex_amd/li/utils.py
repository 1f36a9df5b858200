"""Helpers with the reference's names and semantics (reference: li/utils.py:10-65)."""
import functools
import pickle
import time
from typing import Any, List, Tuple, Union

import numpy as np
import numpy.typing as npt


def pairwise_cosine(x, y):
    """1 - cosine similarity of every row of x against every row of y (utils.py:10-11)."""
    from sklearn.metrics.pairwise import cosine_similarity

    return 1 - cosine_similarity(x, y)


def save_as_pickle(filename: str, obj) -> None:
    """Pickles `obj` into `filename`; the directory must exist (utils.py:14-29)."""
    with open(filename, "wb") as fh:
        pickle.dump(obj, fh)


def log_runtime(level: int, message: str):
    """Decorator: logs `message.format(seconds)` on the owner's `self.logger` (utils.py:32-58)."""

    def wrap(method):
        @functools.wraps(method)
        def timed(self, *args, **kwargs):
            t0 = time.time()
            try:
                return method(self, *args, **kwargs)
            finally:
                self.logger.log(level, message.format(time.time() - t0))

        return timed

    return wrap


def serialize(lst: List[Any]) -> str:
    """`[1, 2] -> "1,2"` (utils.py:56-58)."""
    return ",".join(str(v) for v in lst)


def filter_path_idxs(paths: npt.NDArray[Union[np.int32, np.int64]], path: Tuple) -> npt.NDArray[np.int64]:
    """Indexes of the rows of `paths` equal to `path` (utils.py:61-65).

    Host-side helper kept for API compatibility; on the query hot path the same grouping is done on
    the GPU by the routing kernels (lmi_kernels.h: route_*_kernel)."""
    target = np.asarray(path)
    return np.flatnonzero((np.asarray(paths) == target).all(axis=1))
