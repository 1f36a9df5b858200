#!/usr/bin/env python3
"""Experiment driver with the reference's command line (reference: search/search.py:1-349).

    python learnedmetricindex_amd/search.py --dataset pca96v2 --emb pca96 --size 100K \\
        --n-categories 10 10 --epochs 100 --model-type MLP --lr 0.01 -bp 10 --clustering-algorithm scikit_kmeans

Same flags, same flow (`run` -> `evaluate_learned_index` -> `li.search` per bucket count ->
`store_results`), same log lines and result schema (`knns` uint32, `dists` float64, attrs `algo, data,
buildtime, querytime, size, params`).  Differences, all forced by the MI355X image:

* the SISAP S3 download (`prepare`, search.py:38-48) needs the network; when the h5 files are not
  already under `data/<kind>/<size>/` the driver generates a synthetic stand-in of the same shape
  (unit-norm Gaussian mixture, seed 2023; lower-dimensional kinds are projections of the 768-d set, like
  the challenge's PCA variants) -- pass `--no-synthetic` to insist on real files;
* h5py is not installed: results are written as `.npz` with the same keys when it is missing;
* `--eval` adds what the un-vendored `eval/` submodule did: recall@k against exact search on the GPU.

Known quirks of the reference's CLI are kept: `-b/--n-buckets` is parsed and unused (search.py:316),
`--preprocess/--save` are `type=bool` (any non-empty string is true)."""
import argparse
import logging
import os
import sys
import time
from pathlib import Path
from typing import Any, Dict, List

import numpy as np
import numpy.typing as npt
import pandas as pd

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:  # reference style: `li` is importable from the directory of search.py
    sys.path.insert(0, _HERE)

from li.Baseline import Baseline  # noqa: E402
from li.BuildConfiguration import BuildConfiguration  # noqa: E402
from li.clustering import algorithms  # noqa: E402
from li.LearnedIndexBuilder import LearnedIndexBuilder  # noqa: E402
from li.utils import save_as_pickle, serialize  # noqa: E402

np.random.seed(2023)
logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(levelname)-5.5s][%(name)-.20s] %(message)s")
LOG = logging.getLogger(__name__)

MODELS_DIR_NAME = "models"
SIZES = {"100K": 100_000, "300K": 300_000, "10M": 10_000_000, "30M": 30_000_000, "100M": 100_000_000}
DIMS = {"clip768v2": 768, "pca96v2": 96, "pca32v2": 32}
N_QUERIES = 10_000  # public-queries-10k
SYNTHETIC = True


def _l2n(x):
    n = np.linalg.norm(x, axis=1, keepdims=True)
    n[n == 0] = 1
    return (x / n).astype(np.float32)


def _synthetic(kind: str, size: str):
    """(dataset f32[N,d], queries f32[10k,d]) for `kind`: a 768-d unit-norm mixture, or its random
    projection to the kind's dimensionality (same seed -> the kinds describe the same objects)."""
    n, d = SIZES[size], DIMS[kind]
    rs = np.random.RandomState(2023)
    centres = rs.randn(256, 768).astype(np.float32)
    proj = None if d == 768 else (np.random.RandomState(d).randn(768, d) / np.sqrt(768)).astype(np.float32)

    def draw(m, seed):
        r = np.random.RandomState(seed)
        out = np.empty((m, d), dtype=np.float32)
        for s in range(0, m, 1 << 17):
            e = min(m, s + (1 << 17))
            x = _l2n(centres[r.randint(256, size=e - s)] + r.randn(e - s, 768).astype(np.float32))
            out[s:e] = x if proj is None else x @ proj
        return out

    return draw(n, 11), draw(N_QUERIES, 12)


def download(src, dst):
    if not os.path.exists(dst):
        from urllib.request import urlretrieve

        os.makedirs(Path(dst).parent, exist_ok=True)
        LOG.info("downloading %s -> %s..." % (src, dst))
        urlretrieve(src, dst)


def prepare(kind, size):
    """Makes data/<kind>/<size>/{dataset,query}.* available (search.py:38-48)."""
    base = os.path.join("data", kind, size)
    if all(os.path.exists(os.path.join(base, f"{v}.h5")) or os.path.exists(os.path.join(base, f"{v}.npy"))
           for v in ("query", "dataset")):
        return
    if SYNTHETIC:
        LOG.info(f"no local copy of {kind}/{size}: generating the synthetic stand-in")
        os.makedirs(base, exist_ok=True)
        data, queries = _synthetic(kind, size)
        np.save(os.path.join(base, "dataset.npy"), data)
        np.save(os.path.join(base, "query.npy"), queries)
        return
    url = "https://sisap-23-challenge.s3.amazonaws.com/SISAP23-Challenge"
    for version, src in {"query": f"{url}/public-queries-10k-{kind}.h5",
                         "dataset": f"{url}/laion2B-en-{kind}-n={size}.h5"}.items():
        target = os.path.join(base, f"{version}.h5")
        download(src, target)
        assert os.path.exists(target), f"Failed to download {src}"


def _load(kind, size, version, key):
    base = os.path.join("data", kind, size)
    if os.path.exists(os.path.join(base, f"{version}.h5")):
        import h5py

        return np.array(h5py.File(os.path.join(base, f"{version}.h5"), "r")[key])
    return np.load(os.path.join(base, f"{version}.npy"))


def store_results(dst, algo, kind, dists, anns, buildtime, querytime, params, size):
    """Result file with the reference's schema (search.py:51-63); `.npz` when h5py is absent."""
    os.makedirs(Path(dst).parent, exist_ok=True)
    try:
        import h5py
    except ImportError:
        dst = os.path.splitext(dst)[0] + ".npz"
        LOG.info(f"Storing results in {dst}")
        np.savez(dst, knns=anns, dists=dists, algo=algo, data=kind, buildtime=buildtime, querytime=querytime,
                 size=size, params=params)
        return dst
    LOG.info(f"Storing results in {dst}")
    with h5py.File(dst, "w") as f:
        for name, value in (("algo", algo), ("data", kind), ("buildtime", buildtime), ("querytime", querytime),
                            ("size", size), ("params", params)):
            f.attrs[name] = value
        f.create_dataset("knns", anns.shape, dtype=anns.dtype)[:] = anns
        f.create_dataset("dists", dists.shape, dtype=dists.dtype)[:] = dists
    return dst


def format_identifier(bucket: int, kind: str, config: BuildConfiguration, clustering_algorithms: List[str],
                      short_identifier: str, size: str):
    return (f"{short_identifier}-{kind}-{size}-ep={serialize(config.epochs)}-lr={serialize(config.lrs)}"
            f"-cat={serialize(config.n_categories)}-model={serialize(config.model_types)}-buck={bucket}"
            f"-clustering_algorithm={serialize(clustering_algorithms)}-{os.environ['PBS_JOBID']}")


def format_models_filename(kind: str, config: BuildConfiguration, clustering_algorithms: List[str],
                           preprocess: bool, size: str):
    return (f"./{MODELS_DIR_NAME}/{kind}-{size}-ep={serialize(config.epochs)}-lr={serialize(config.lrs)}"
            f"-cat={serialize(config.n_categories)}-model={serialize(config.model_types)}-prep={preprocess}"
            f"-clustering_algorithm={serialize(clustering_algorithms)}-{os.environ['PBS_JOBID']}")


def run(kind: str, key: str, size: str, k: int, index_type: str, n_buckets_perc: List[int],
        n_categories: List[int], epochs: List[int], model_types: List[str], lr: List[float], preprocess: bool,
        save: bool, clustering_algorithms: List[str], evaluate: bool = False):
    assert index_type in {"baseline", "learned-index"}, f"Unknown index type: {index_type}"
    LOG.info(f"Running with: kind={kind}, key={key}, size={size}, n_buckets_perc={n_buckets_perc},"
             f" n_categories={n_categories}, clustering_algorithms={clustering_algorithms},"
             f" epochs={epochs}, lr={lr}, model_types={model_types}, preprocess={preprocess}, save={save}")
    prepare(kind, size)
    data: npt.NDArray[np.float32] = _load(kind, size, "dataset", key)
    queries: npt.NDArray[np.float32] = _load(kind, size, "query", key)
    if preprocess:
        from sklearn import preprocessing

        data = preprocessing.normalize(data)
        queries = preprocessing.normalize(queries)
    n, d = data.shape
    LOG.info(f"Loaded downloaded data, shape: n={n}, d={d}")
    LOG.info(f"Loaded downloaded queries, shape: queries={queries.shape}")
    if index_type == "baseline":
        baseline = Baseline()
        LOG.info(f"Build time: {baseline.build(data)}")
        return baseline.search(queries=queries, data=data, k=k)
    return evaluate_learned_index(data, clustering_algorithms, epochs, model_types, lr, k, kind, n_buckets_perc,
                                  n_categories, preprocess, queries, save, size, evaluate)


def evaluate_learned_index(data, clustering_algorithms: List[str], epochs: List[int], model_type: List[str],
                           lr: List[float], k: int, kind: str, n_buckets_perc: List[int],
                           n_categories: List[int], preprocess: bool, queries, save: bool, size: str,
                           evaluate: bool = False):
    s = time.time()
    data_pd = pd.DataFrame(data)
    data_pd.index += 1  # 1-based object ids (search.py:190-191)
    kind_search, key_search = "clip768v2", "emb"
    if kind != kind_search:  # navigate in `kind`, scan in clip768v2 (search.py:194-213)
        LOG.info("Loading data to be used in search")
        prepare(kind_search, size)
        data_search = pd.DataFrame(_load(kind_search, size, "dataset", key_search))
        data_search.index += 1
        queries_search = _load(kind_search, size, "query", key_search)
        LOG.info(f"Loaded downloaded data, shape: n={data_search.shape[0]}, d={data_search.shape[1]}")
        LOG.info(f"Loaded downloaded queries, shape: queries={queries_search.shape}")
    else:
        data_search, queries_search = data_pd, queries
    config = BuildConfiguration([algorithms[a] for a in clustering_algorithms], epochs, model_type, lr, n_categories)
    li, data_prediction, n_buckets_in_index, build_t, cluster_t = LearnedIndexBuilder(data_pd, config).build()
    LOG.info(f"Total number of buckets in the index: {n_buckets_in_index}")
    LOG.info(f"Cluster time: {cluster_t}")
    LOG.info(f"Pure build time: {build_t}")
    LOG.info(f"Overall build time: {time.time() - s}")
    if save:
        os.makedirs(MODELS_DIR_NAME, exist_ok=True)
        filename = format_models_filename(kind, config, clustering_algorithms, preprocess, size)
        LOG.info(f"Saving as {filename}")
        save_as_pickle(f"{filename}.pkl", li)
    n_buckets = sorted({b for b in (int((p / 100) * n_buckets_in_index) for p in n_buckets_perc) if b > 0})
    LOG.info(f"Number of buckets to search in: {n_buckets}")
    outputs = {}
    for bucket in n_buckets:
        LOG.info(f"Searching with {bucket} buckets")
        dists, nns, measured_time = li.search(
            data_navigation=data_pd, queries_navigation=queries, data_search=data_search,
            queries_search=queries_search, data_prediction=data_prediction, n_categories=n_categories,
            n_buckets=bucket, k=k)
        LOG.info(f"Inference time: {measured_time['inference']}")
        LOG.info(f"Search time: {measured_time['search']}")
        LOG.info(f"Search within buckets time: {measured_time['search_within_buckets']}")
        LOG.info(f"Sequential search time: {measured_time['seq_search']}")
        LOG.info(f"Sort time: {measured_time['sort']}")
        if evaluate:
            outputs[f"recall_{bucket}"] = recall = _recall(data_search, queries_search, nns, k)
            LOG.info(f"Recall@{k} with {bucket} buckets: {recall:.5f}")
        identifier = format_identifier(bucket, kind, config, clustering_algorithms, "learned-index", size)
        store_results(os.path.join("result/", kind, size, f"{identifier}.h5"), "Learned-index", kind, dists, nns,
                      build_t, measured_time["search"], identifier, size)
        outputs[bucket] = (dists, nns, measured_time)
    li.close()
    return outputs


def _recall(data_search: pd.DataFrame, queries_search, nns, k: int, sample: int = 1000) -> float:
    """recall@k of the first `sample` queries against exact inner-product search on the GPU
    (notebook cell 31: |I & gt| / (k * nq); ids 1-based in index order)."""
    try:
        from .. import _capi  # type: ignore
    except ImportError:
        import _capi  # type: ignore
    m = min(sample, nns.shape[0])
    cols = [c for c in data_search.columns if not (isinstance(c, str) and c.startswith("category_L"))]
    _, gt = _capi.knn_ip(np.asarray(queries_search[:m], dtype=np.float32),
                         data_search[cols].to_numpy(dtype=np.float32), min(k, 10))
    labels = data_search.index.to_numpy()
    hits = sum(len(set(labels[g[g >= 0]].tolist()) & set(a.tolist())) for g, a in zip(gt, nns[:m, :k]))
    return hits / float(min(k, 10) * m)


def expand(array: List[Any], size: int):
    assert len(array) == 1
    return [array[0]] * size


def validate_and_expand_per_level_arguments(args: Dict[str, Any]):
    """Per-level flags are given once or once per level (search.py:292-303)."""
    for arg in ("clustering_algorithm", "model_type", "epochs", "lr", "n_categories"):
        if len(args[arg]) == 1:
            args[arg] = expand(args[arg], len(args["n_categories"]))
        else:
            assert len(args[arg]) == len(args["n_categories"])


def main(argv=None):
    global SYNTHETIC
    parser = argparse.ArgumentParser()
    parser.add_argument("--dataset", default="pca96v2")
    parser.add_argument("--emb", default="pca96")
    parser.add_argument("--size", default="100K", choices=list(SIZES))
    parser.add_argument("--k", default=10, type=int)
    parser.add_argument("--n-categories", nargs="+", default=[10, 10], type=int)
    parser.add_argument("--epochs", nargs="+", default=[100], type=int)
    parser.add_argument("--model-type", nargs="+", default=["MLP"])
    parser.add_argument("--lr", nargs="+", default=[0.01], type=float)
    parser.add_argument("-b", "--n-buckets", nargs="+", default=[2, 3, 4], type=int)  # unused, as in the reference
    parser.add_argument("-bp", "--buckets-perc", nargs="+", default=[10], type=int)
    parser.add_argument("--preprocess", default=True, type=bool)
    parser.add_argument("--save", default=True, type=bool)
    parser.add_argument("--clustering-algorithm", nargs="+", default=["faiss_kmeans"], choices=algorithms.keys())
    parser.add_argument("--no-synthetic", action="store_true", help="fail instead of generating stand-in data")
    parser.add_argument("--eval", action="store_true", help="log recall@k against exact search")
    args = parser.parse_args(argv)
    SYNTHETIC = not args.no_synthetic
    validate_and_expand_per_level_arguments(vars(args))
    if "PBS_JOBID" not in os.environ:
        os.environ["PBS_JOBID"] = "unknown"
    return run(args.dataset, args.emb, args.size, args.k, "learned-index", args.buckets_perc, args.n_categories,
               args.epochs, args.model_type, args.lr, args.preprocess, args.save, args.clustering_algorithm,
               evaluate=args.eval)


if __name__ == "__main__":
    main()
