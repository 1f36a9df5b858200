"""GPU (`-m gpu`): the "next" rows N2-N4 of SURVEY section 8f -- index build, on-disk format, driver."""
import os
import sys

import numpy as np
import pandas as pd
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def frame(X):
    df = pd.DataFrame(X)
    df.index += 1
    return df


@pytest.mark.parametrize("ncat", [[12], [4, 3]])
def test_builder_then_search_and_roundtrip(oracle, tmp_path, ncat):
    from learnedmetricindex_amd import index_io
    from learnedmetricindex_amd.li.BuildConfiguration import BuildConfiguration
    from learnedmetricindex_amd.li.clustering import algorithms
    from learnedmetricindex_amd.li.LearnedIndexBuilder import LearnedIndexBuilder
    from learnedmetricindex_amd.li.model import linear_layers

    torch.manual_seed(2023)
    X, Q = synth.mixture(2023, 5000, 64, 12, 200)
    df = frame(X)
    cfg = BuildConfiguration([algorithms["scikit_kmeans"]], [20], ["MLP"], [0.01], ncat)
    li, dp, n_buckets, build_t, cluster_t = LearnedIndexBuilder(df, cfg).build()
    assert dp.shape == (5000, len(ncat)) and dp.dtype == np.int64 and n_buckets == len(li.bucket_paths)
    # placement == argmax of the trained model, evaluated independently by the oracle (:76)
    root = linear_layers(li.root_model.model)
    np.testing.assert_array_equal(dp[:, 0], oracle.predict(root, X))
    assert len(np.unique(dp[:, 0])) == ncat[0]  # "train until every category is predicted"
    nb = 3
    dists, nns, mt = li.search(df, Q, df, Q, dp, ncat, nb, 10)
    if len(ncat) == 1:
        do, no, _ = oracle.search(root, Q, X, Q, dp, nb, 10)
    else:
        internal = [(p, linear_layers(m.model)) for p, m in li.internal_models.items()]
        bo = oracle.precompute_bucket_order_multilevel(root, internal, li.bucket_paths, Q, nb, ncat)
        do, no, _ = oracle.search(root, Q, X, Q, dp, nb, 10, bucket_order=bo)
    np.testing.assert_array_equal(nns, no)
    np.testing.assert_array_equal(dists, do)
    gt = np.argsort(-(Q.astype(np.float64) @ X.astype(np.float64).T), axis=1)[:, :10] + 1
    assert oracle.recall_at_k(nns, gt) > 0.5
    # on-disk round trip: same answers from a freshly loaded index, no DataFrames
    d = str(tmp_path / "idx")
    index_io.save_index(d, li, ncat)
    li2, ncat2 = index_io.load_index(d)
    assert ncat2 == ncat
    d2, n2, _ = li2.search_resident(Q, Q, ncat2, nb, 10)
    np.testing.assert_array_equal(n2, nns)
    np.testing.assert_array_equal(d2, dists)
    li.close()
    li2.close()


def test_driver_cli_smoke(tmp_path, monkeypatch):
    """The reference's CI smoke run (ci.yml:116-122: `python3 search/search.py`) with this build's driver, on
    generated vectors; also: missing files without --synthetic must fail loudly, results are stamped synthetic."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lmi_search_driver", os.path.join(root, "learnedmetricindex_amd", "search.py"))
    drv = importlib.util.module_from_spec(spec)
    sys.modules["lmi_search_driver"] = drv  # dataclasses resolves the module of Experiment
    spec.loader.exec_module(drv)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setitem(drv.CARDINALITY, "100K", 6000)     # keep the smoke run small
    monkeypatch.setattr(drv, "QUERY_COUNT", 100)
    argv = ["--dataset", "pca32v2", "--emb", "pca32", "--size", "100K", "--n-categories", "6",
            "--epochs", "10", "--model-type", "MLP", "--lr", "0.01", "-bp", "50",
            "--clustering-algorithm", "scikit_kmeans", "--eval"]
    with pytest.raises(FileNotFoundError):
        drv.main(argv)
    out = drv.main(argv + ["--synthetic"])
    (bucket, (dists, nns, mt)), = [(k, v) for k, v in out.items() if not str(k).startswith("recall")]
    assert bucket == 3 and dists.shape == nns.shape == (100, 10) and nns.dtype == np.uint32
    assert out["recall_3"] > 0.3 and mt["search"] > 0
    files = [f for f in os.listdir(tmp_path / "result" / "pca32v2" / "100K")]
    assert len(files) == 1 and (tmp_path / "models").exists()
    if files[0].endswith(".npz"):
        z = np.load(tmp_path / "result" / "pca32v2" / "100K" / files[0])
        assert str(z["data"]) == "synthetic-pca32v2" and z["knns"].dtype == np.uint32 and z["dists"].dtype == np.float64
    assert drv.bucket_budgets([10, 50, 1], 6) == [3]   # int(p/100 * n), zeros dropped


def test_baseline_is_gpu_bruteforce():
    """li.Baseline: exact cosine k-NN on the GPU (ids 1-based), against float64 numpy."""
    from learnedmetricindex_amd.li.Baseline import Baseline

    rs = np.random.RandomState(3)
    X = rs.randn(3000, 48).astype(np.float32) * rs.uniform(0.1, 10, size=(3000, 1)).astype(np.float32)
    Q = rs.randn(64, 48).astype(np.float32)
    d, ids, secs = Baseline().search(Q, X, k=10)
    xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    qn = Q / np.linalg.norm(Q, axis=1, keepdims=True)
    ref = 1 - qn.astype(np.float64) @ xn.astype(np.float64).T
    order = np.argsort(ref, axis=1, kind="stable")[:, :10]
    assert (ids == order + 1).mean() > 0.99 and secs > 0
    np.testing.assert_allclose(d, np.take_along_axis(ref, order, 1), atol=2e-6)
    d3, ids3, _ = Baseline().search(Q[:4], X[:3], k=5)   # fewer objects than k
    assert np.all(ids3[:, 3:] == 0) and np.all(np.isinf(d3[:, 3:]))
