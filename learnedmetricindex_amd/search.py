#!/usr/bin/env python3
"""Experiment driver for the MI355X build: build an index, answer the query set at several bucket budgets,
write one result file per budget.

It accepts the reference driver's command line (SURVEY.md section 5 "Config / flags"; reference
search/search.py:306-327) so that existing job scripts keep working:

    python learnedmetricindex_amd/search.py --dataset pca96v2 --emb pca96 --size 100K \\
        --n-categories 10 10 --epochs 100 --model-type MLP --lr 0.01 -bp 10 --clustering-algorithm scikit_kmeans

and produces the reference's result schema (datasets `knns` uint32 / `dists` float64, attributes `algo, data,
buildtime, querytime, size, params`; search/search.py:51-63) so the SISAP `eval` tooling reads it.  Everything
else is this build's own structure:

    Experiment   the parsed command line (per-level lists expanded)
    VectorSource where the vectors come from: local `data/<kind>/<size>/{dataset,query}.(h5|npy)` or, only
                 when `--synthetic` is given, a generated stand-in whose results are stamped `synthetic-<kind>`
    ResultSink   `.h5` when h5py is importable, `.npz` with the same keys otherwise
    run()        build (LearnedIndexBuilder) -> one resident index in HBM -> `search_resident` per budget ->
                 optional recall@k against the GPU brute-force `Baseline`

Flags the reference parses but ignores are accepted and ignored the same way (`-b/--n-buckets`); its
`type=bool` flags (`--preprocess`, `--save`) keep "any non-empty string is true".  Downloading from the SISAP
S3 bucket is not supported (no network on the target machines): place the files locally or use `--synthetic`.
"""
from __future__ import annotations

import argparse
import dataclasses
import logging
import os
import sys
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:  # `li` importable next to this file, as in the reference layout
    sys.path.insert(0, _HERE)

from li.Baseline import Baseline  # noqa: E402
from li.BuildConfiguration import BuildConfiguration  # noqa: E402
from li.clustering import algorithms  # noqa: E402
from li.LearnedIndexBuilder import LearnedIndexBuilder  # noqa: E402
from li.utils import save_as_pickle, serialize  # noqa: E402

LOG = logging.getLogger("lmi.driver")

SCAN_KIND, SCAN_KEY = "clip768v2", "emb"   # the scan always runs on the 768-d embeddings
PER_LEVEL = ("clustering_algorithm", "model_type", "epochs", "lr")
CARDINALITY = {"100K": 100_000, "300K": 300_000, "10M": 10_000_000, "30M": 30_000_000, "100M": 100_000_000}
WIDTH = {"clip768v2": 768, "pca96v2": 96, "pca32v2": 32}
QUERY_COUNT = 10_000


# ------------------------------------------------------------------------------------------- command line
@dataclasses.dataclass
class Experiment:
    dataset: str
    emb: str
    size: str
    k: int
    n_categories: List[int]
    epochs: List[int]
    model_type: List[str]
    lr: List[float]
    buckets_perc: List[int]
    preprocess: bool
    save: bool
    clustering_algorithm: List[str]
    synthetic: bool = False
    evaluate: bool = False
    job: str = "unknown"

    @classmethod
    def from_argv(cls, argv: Optional[Sequence[str]] = None) -> "Experiment":
        p = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
        p.add_argument("--dataset", default="pca96v2")
        p.add_argument("--emb", default="pca96")
        p.add_argument("--size", default="100K", choices=sorted(CARDINALITY, key=CARDINALITY.get))
        p.add_argument("--k", default=10, type=int)
        p.add_argument("--n-categories", nargs="+", default=[10, 10], type=int)
        p.add_argument("--epochs", nargs="+", default=[100], type=int)
        p.add_argument("--model-type", nargs="+", default=["MLP"])
        p.add_argument("--lr", nargs="+", default=[0.01], type=float)
        p.add_argument("-b", "--n-buckets", nargs="+", default=[2, 3, 4], type=int, help="accepted, unused (as upstream)")
        p.add_argument("-bp", "--buckets-perc", nargs="+", default=[10], type=int)
        p.add_argument("--preprocess", default=True, type=bool)
        p.add_argument("--save", default=True, type=bool)
        p.add_argument("--clustering-algorithm", nargs="+", default=["faiss_kmeans"], choices=sorted(algorithms))
        p.add_argument("--synthetic", action="store_true",
                       help="generate stand-in vectors when the dataset files are missing (results are stamped synthetic)")
        p.add_argument("--eval", dest="evaluate", action="store_true", help="report recall@k against GPU brute force")
        a = vars(p.parse_args(argv))
        a.pop("n_buckets")
        levels = len(a["n_categories"])
        for name in PER_LEVEL:  # one value for all levels, or one per level
            if len(a[name]) == 1:
                a[name] = a[name] * levels
            elif len(a[name]) != levels:
                p.error(f"--{name.replace('_', '-')} needs 1 or {levels} values, got {len(a[name])}")
        return cls(job=os.environ.get("PBS_JOBID", "unknown"), **a)

    def build_configuration(self) -> BuildConfiguration:
        return BuildConfiguration([algorithms[c] for c in self.clustering_algorithm], self.epochs, self.model_type,
                                  self.lr, self.n_categories)

    def tag(self, prefix: str, **extra) -> str:
        """File-name stem in the upstream format: <prefix>-ep=..-lr=..-cat=..-model=..[-k=v..]-<job>."""
        parts = [prefix, f"ep={serialize(self.epochs)}", f"lr={serialize(self.lr)}", f"cat={serialize(self.n_categories)}",
                 f"model={serialize(self.model_type)}"]
        parts += [f"{k}={v}" for k, v in extra.items()]
        parts += [f"clustering_algorithm={serialize(self.clustering_algorithm)}", self.job]
        return "-".join(parts)


# ------------------------------------------------------------------------------------------- vectors
class VectorSource:
    """`data/<kind>/<size>/{dataset,query}.h5|.npy`; synthetic stand-ins only on request."""

    def __init__(self, size: str, synthetic: bool, root: str = "data"):
        self.size, self.synthetic, self.root = size, synthetic, root
        self.generated = set()

    def _path(self, kind: str, what: str, ext: str) -> str:
        return os.path.join(self.root, kind, self.size, f"{what}.{ext}")

    def load(self, kind: str, key: str) -> Tuple[np.ndarray, np.ndarray]:
        """(objects f32[N,d], queries f32[nq,d]) of embedding `kind`."""
        out = []
        for what in ("dataset", "query"):
            if os.path.exists(self._path(kind, what, "h5")):
                import h5py

                with h5py.File(self._path(kind, what, "h5"), "r") as fh:
                    out.append(np.asarray(fh[key], dtype=np.float32))
            elif os.path.exists(self._path(kind, what, "npy")):
                out.append(np.load(self._path(kind, what, "npy")).astype(np.float32, copy=False))
            elif self.synthetic:
                self.generated.add(kind)
                out.append(self._generate(kind, what))
            else:
                raise FileNotFoundError(
                    f"{self._path(kind, what, 'h5')} (or .npy) not found; there is no network access to fetch the SISAP "
                    f"files -- copy them there or pass --synthetic for a generated stand-in")
        return out[0], out[1]

    def _generate(self, kind: str, what: str) -> np.ndarray:
        """Unit-norm mixture of 256 Gaussians in 768-d; the narrower kinds are one fixed random projection of the
        same objects (so that `pca32v2` navigation and `clip768v2` scan describe the same set)."""
        n = CARDINALITY[self.size] if what == "dataset" else QUERY_COUNT
        d = WIDTH[kind]
        centres = np.random.RandomState(2023).randn(256, 768).astype(np.float32)
        proj = None if d == 768 else (np.random.RandomState(d).randn(768, d) / np.sqrt(768.0)).astype(np.float32)
        rs = np.random.RandomState(11 if what == "dataset" else 12)
        out = np.empty((n, d), dtype=np.float32)
        step = 1 << 17
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            x = centres[rs.randint(256, size=hi - lo)] + rs.randn(hi - lo, 768).astype(np.float32)
            x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-30)
            out[lo:hi] = x if proj is None else x @ proj
        return out

    def stamp(self, kind: str) -> str:
        return f"synthetic-{kind}" if kind in self.generated else kind


def unit_rows(x: np.ndarray) -> np.ndarray:
    """sklearn.preprocessing.normalize(x) (l2, zero rows untouched) without the dependency."""
    n = np.linalg.norm(x, axis=1, keepdims=True)
    return (x / np.where(n == 0, 1, n)).astype(np.float32)


# ------------------------------------------------------------------------------------------- results
class ResultSink:
    """One file per bucket budget under result/<kind>/<size>/ with the upstream schema."""

    def __init__(self, root: str = "result"):
        self.root = root
        try:
            import h5py  # noqa: F401

            self.h5 = True
        except ImportError:
            self.h5 = False

    def write(self, folder_kind: str, folder_size: str, stem: str, dists: np.ndarray, knns: np.ndarray, **attrs) -> str:
        folder = os.path.join(self.root, folder_kind, folder_size)
        os.makedirs(folder, exist_ok=True)
        path = os.path.join(folder, stem + (".h5" if self.h5 else ".npz"))
        if self.h5:
            import h5py

            with h5py.File(path, "w") as fh:
                fh.attrs.update(attrs)
                fh.create_dataset("knns", data=knns)
                fh.create_dataset("dists", data=dists)
        else:
            np.savez(path, knns=knns, dists=dists, **attrs)
        LOG.info("results -> %s", path)
        return path


# ------------------------------------------------------------------------------------------- the run
def bucket_budgets(percentages: Sequence[int], n_buckets_in_index: int) -> List[int]:
    """`-bp`: per cent of the index's buckets -> bucket counts, int(p/100 * n) like upstream; zeros dropped."""
    return sorted({b for b in (int(p / 100 * n_buckets_in_index) for p in percentages) if b > 0})


def recall_against_bruteforce(queries: np.ndarray, objects: pd.DataFrame, knns: np.ndarray, k: int, sample: int = 1000):
    """recall@k of the first `sample` queries (notebook cell 31: |found & true| / (k * nq)); ground truth from the
    GPU brute-force Baseline over the scan vectors."""
    m, kk = min(sample, knns.shape[0]), min(k, 10)
    _, truth, _ = Baseline().search(queries[:m], objects.to_numpy(dtype=np.float32), k=kk)
    labels = objects.index.to_numpy()
    truth = labels[truth - 1]  # Baseline ids are 1-based positions
    return float(np.mean([len(set(t) & set(f)) / kk for t, f in zip(truth.tolist(), knns[:m, :k].tolist())]))


def run(exp: Experiment) -> Dict:
    LOG.info("experiment: %s", dataclasses.asdict(exp))
    src = VectorSource(exp.size, exp.synthetic)
    nav_x, nav_q = src.load(exp.dataset, exp.emb)
    if exp.preprocess:
        nav_x, nav_q = unit_rows(nav_x), unit_rows(nav_q)
    LOG.info("navigation vectors %s, queries %s", nav_x.shape, nav_q.shape)
    nav = pd.DataFrame(nav_x)
    nav.index += 1  # object ids are 1-based
    if exp.dataset == SCAN_KIND:
        scan, scan_q = nav, nav_q
    else:  # navigate in the narrow embedding, scan in clip768v2 (not normalised by the driver, as upstream)
        sx, scan_q = src.load(SCAN_KIND, SCAN_KEY)
        scan = pd.DataFrame(sx)
        scan.index += 1
        LOG.info("scan vectors %s, queries %s", scan.shape, scan_q.shape)

    t0 = time.time()
    index, placement, n_buckets_in_index, build_s, cluster_s = LearnedIndexBuilder(nav, exp.build_configuration()).build()
    LOG.info("index with %d buckets: clustering %.2fs, build %.2fs, total %.2fs", n_buckets_in_index, cluster_s, build_s,
             time.time() - t0)
    if exp.save:
        os.makedirs("models", exist_ok=True)
        save_as_pickle(os.path.join("models", exp.tag(f"{exp.dataset}-{exp.size}", prep=exp.preprocess) + ".pkl"), index)

    index.prepare(nav, scan, placement, exp.n_categories)  # one upload; every budget below reuses the resident slab
    sink, out = ResultSink(), {}
    for budget in bucket_budgets(exp.buckets_perc, n_buckets_in_index):
        dists, knns, clock = index.search_resident(nav_q, scan_q, exp.n_categories, n_buckets=budget, k=exp.k)
        LOG.info("%d buckets: search %.4fs (inference %.4fs, within buckets %.4fs, scan %.4fs, merge %.4fs)", budget,
                 clock["search"], clock["inference"], clock["search_within_buckets"], clock["seq_search"], clock["sort"])
        if exp.evaluate:
            out[f"recall_{budget}"] = recall_against_bruteforce(scan_q, scan, knns, exp.k)
            LOG.info("%d buckets: recall@%d = %.5f", budget, exp.k, out[f"recall_{budget}"])
        stem = exp.tag(f"learned-index-{exp.dataset}-{exp.size}", buck=budget)
        sink.write(exp.dataset, exp.size, stem, dists, knns, algo="Learned-index", data=src.stamp(exp.dataset),
                   buildtime=build_s, querytime=clock["search"], size=exp.size, params=stem)
        out[budget] = (dists, knns, clock)
    index.close()
    return out


def main(argv: Optional[Sequence[str]] = None) -> Dict:
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(levelname)-5.5s][%(name)-.20s] %(message)s")
    np.random.seed(2023)
    return run(Experiment.from_argv(argv))


if __name__ == "__main__":
    main()
