"""CPU (`-m "not gpu"`): the two CPU baselines bench.py times (oracle/cpu_baseline.py) return the reference's
answers: compared with the reference-generated fixtures and the oracle, modulo float32 near-ties (BLAS
summation order vs the canonical chain)."""
import numpy as np
import pandas as pd
import pytest
import torch

from helpers import compare_modulo_near_ties, inputs_for, layers_from, load_golden


@pytest.mark.parametrize("name", ["G1", "G3", "G4"])
def test_cpu_baselines_match_reference_fixture(oracle, name):
    from oracle import cpu_baseline as cb

    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    layers = layers_from(g)
    nb, k = int(g["n_buckets"]), int(g["k"])
    dp = g["data_prediction"][:, 0].astype(np.int64)
    L = layers[-1][0].shape[0]
    order = cb.mlp_order_numpy(layers, Qn, nb)
    np.testing.assert_array_equal(order, g["ref_bucket_order"][:, :, 0])
    # reference-structured: two distinct frames, 1-based labels
    nav = pd.DataFrame(Xn.copy()); nav.index += 1
    srch = pd.DataFrame(Xs.copy()); srch.index += 1
    d, n, t = cb.reference_structured(nav, srch, Qs, order, dp, k)
    assert "category_L1" not in nav.columns and n.dtype == np.uint32 and d.dtype == np.float64
    compare_modulo_near_ties(g["ref_dists"], g["ref_nns"], d, n)
    # best effort: bucket-contiguous slab
    perm = np.argsort(dp, kind="stable")
    offsets = np.concatenate([[0], np.cumsum(np.bincount(dp, minlength=L))])
    slab = torch.from_numpy(np.ascontiguousarray(Xs[perm]))
    ids = (perm + 1).astype(np.uint32)
    d2, n2, o2, secs = cb.best_effort(slab, offsets, ids, layers, torch.from_numpy(Qs), nb, k, threads=2)
    np.testing.assert_array_equal(o2, order)
    compare_modulo_near_ties(g["ref_dists"], g["ref_nns"], d2, n2)
    assert cb.id_agreement(n2, g["ref_nns"]) > 0.995  # G4 holds 21 duplicate vectors: exact ties
    # the bucket-parallel form of the same baseline: same answers (BLAS blocks a product differently on one thread: last-bit
    # differences of a distance, ids equal outside float32 near-ties)
    d3, n3, o3, _ = cb.best_effort_bucket_parallel(slab, offsets, ids, layers, torch.from_numpy(Qs), nb, k, workers=3)
    np.testing.assert_array_equal(o3, order)
    compare_modulo_near_ties(g["ref_dists"], g["ref_nns"], d3, n3)
    np.testing.assert_allclose(d3, d2, rtol=0, atol=1e-6)
    # a deadline that has already passed stops the reference-structured loop after its first rank
    nav2 = pd.DataFrame(Xn.copy()); nav2.index += 1
    _, _, t2 = cb.reference_structured(nav2, srch, Qs, order, dp, k, deadline_s=0.0)
    assert t2["ranks_done"] == 1 and t["ranks_done"] == nb and cb.usable_cpus() >= 1
