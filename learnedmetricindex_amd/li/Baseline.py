"""Brute-force cosine baseline (reference: li/Baseline.py:7-21): ids are 1-based."""
import time

import numpy as np

from .Logger import Logger
from .utils import pairwise_cosine


class Baseline(Logger):
    def __init__(self):
        pass

    def search(self, queries, data, k=10):
        s = time.time()
        anns = pairwise_cosine(data, queries).T
        order = anns.argsort()[:, :k]
        return np.take_along_axis(anns, order, axis=1), order + 1, time.time() - s

    def build(self, data):
        s = time.time()
        self.logger.info("No build method implemented for baseline.")
        return time.time() - s
