"""GPU (`-m gpu`): random small shapes, the prefilter mode against the all-f32 mode, bit for bit (tools/fuzz_modes.py draws the cases:
d 1..2100 around the kernels' boundaries, 1..2 500 buckets of empty / tiny / ragged / heavy sizes, top-1..8, k 1..20, batches of
1..3 000 queries routed evenly or onto a few buckets, clusters of near-copies, unvisited slots; the low-dimensional kernels' wide form
forced on, off and automatic in turn; a quarter of the cases with the L2 metric), five queries of every case also against the CPU
oracle.  150 cases here (LMI_FUZZ_CASES / LMI_FUZZ_SEED for more); 5 700 ran clean at the end of round 4 (profiles/r04_fuzz_modes.txt)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_shapes_prefilter_equals_exact(oracle):
    from fuzz_modes import one_case
    from learnedmetricindex_amd import _capi

    saved = os.environ.get("LMI_PS_WIDE")
    try:
        seed = int(os.environ.get("LMI_FUZZ_SEED", "2024"))
        for case in range(int(os.environ.get("LMI_FUZZ_CASES", "150"))):
            ok, desc = one_case(_capi, np.random.RandomState(seed * 100003 + case), case, oracle)   # (+ 5 queries per case against the oracle)
            assert ok, desc
    finally:
        if saved is None:
            os.environ.pop("LMI_PS_WIDE", None)
        else:
            os.environ["LMI_PS_WIDE"] = saved
