"""On-disk index: everything `LearnedIndex.search` needs, without the
DataFrames -- replaces the reference's whole-object pickle (search.py:234-241, utils.py:14-29),
which stores the models only and drops `data_prediction` and the vectors.

    <dir>/meta.json      format/version, N, d, metric ("ip" | "l2"), n_categories, bucket paths, model descriptions
    <dir>/weights.npz    Linear weights/biases of the root and internal models (float32)
    <dir>/sizes.npy      int64 [B]    objects per bucket, in bucket-id order
    <dir>/ids.npy        uint32 [N]   object labels, bucket-contiguous
    <dir>/vectors.f32    float32 [N, d] scan vectors, bucket-contiguous row-major (memory-mappable)

`save_index` reads the vectors back from HBM bucket by bucket (lmi_bucket_read); `load_index` streams
them into a fresh device slab in pieces, so neither needs N x d floats of host memory at once."""
import json
import os
from typing import List, Tuple

import numpy as np

from . import _capi
from .li.LearnedIndex import LearnedIndex
from .li.model import linear_layers, network_from_layers

FORMAT, VERSION = "lmi-mi355x-index", 2   # 2: + "metric" (a version-1 directory has none: inner product)


def _put(store, prefix, net):
    layers = linear_layers(net.model)
    for i, (W, b) in enumerate(layers):
        store[f"{prefix}W{i}"], store[f"{prefix}b{i}"] = W, b
    return len(layers)


def _get(store, prefix, n):
    return [(store[f"{prefix}W{i}"], store[f"{prefix}b{i}"]) for i in range(n)]


def save_index(path: str, li: LearnedIndex, n_categories: List[int]) -> None:
    """Writes the resident index of `li` (after prepare()/search()) to directory `path`."""
    eng = li._engine
    assert eng is not None, "nothing resident: call li.prepare(...) or li.search(...) first"
    os.makedirs(path, exist_ok=True)
    sizes = eng.bucket_sizes()
    N, d = int(sizes.sum()), eng.d
    weights = {}
    meta = {"format": FORMAT, "version": VERSION, "N": N, "d": d, "metric": eng.metric, "n_categories": [int(v) for v in n_categories],
            "bucket_paths": [[int(v) for v in p] for p in li.bucket_paths],
            "root_layers": _put(weights, "root_", li.root_model), "internal": []}
    for i, (p, net) in enumerate(li.internal_models.items()):
        meta["internal"].append({"path": [int(v) for v in p], "layers": _put(weights, f"int{i}_", net)})
    if li._path_ids is not None:  # multi-level: bucket id -> path
        meta["paths"] = [list(p) for p, _ in sorted(li._path_ids.items(), key=lambda kv: kv[1])]
    np.savez(os.path.join(path, "weights.npz"), **weights)
    np.save(os.path.join(path, "sizes.npy"), sizes)
    ids = np.empty(N, dtype=np.uint32)
    vec = np.lib.format.open_memmap(os.path.join(path, "vectors.f32.npy"), mode="w+", dtype=np.float32, shape=(N, d))
    o = 0
    for b, n in enumerate(sizes):
        if n:
            rows, bid = eng.read_bucket(b)
            vec[o:o + n], ids[o:o + n] = rows, bid
            o += int(n)
    vec.flush()
    del vec
    np.save(os.path.join(path, "ids.npy"), ids)
    with open(os.path.join(path, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


def load_index(path: str, device: int = 0) -> Tuple[LearnedIndex, List[int]]:
    """(LearnedIndex with the index resident on `device`, n_categories); use `li.search_resident`."""
    meta = json.load(open(os.path.join(path, "meta.json")))
    assert meta["format"] == FORMAT and meta["version"] in (1, VERSION), f"unknown index format {meta.get('format')} v{meta.get('version')}"
    metric = meta.get("metric", "ip")
    assert metric in _capi.Index.METRICS, f"unknown metric {metric!r} in {path}/meta.json"
    store = np.load(os.path.join(path, "weights.npz"))
    root = network_from_layers(_get(store, "root_", meta["root_layers"]))
    internal = {tuple(e["path"]): network_from_layers(_get(store, f"int{i}_", e["layers"]))
                for i, e in enumerate(meta["internal"])}
    li = LearnedIndex(root, internal, [tuple(p) for p in meta["bucket_paths"]])
    sizes = np.load(os.path.join(path, "sizes.npy"))
    ids = np.load(os.path.join(path, "ids.npy"))
    vec = np.load(os.path.join(path, "vectors.f32.npy"), mmap_mode="r")
    eng = _capi.Index(device, metric=metric)
    if len(meta["n_categories"]) == 1:
        eng.set_mlp(linear_layers(root.model))
    else:
        li._path_ids = {tuple(p): i for i, p in enumerate(meta["paths"])}
        li._upload_tree(eng, len(meta["n_categories"]))  # the nodes' models + the tree for the device-side walk
    labels = np.repeat(np.arange(sizes.shape[0], dtype=np.int64), sizes)
    eng.buckets_begin(labels, int(meta["d"]), int(sizes.shape[0]), ids=ids)
    piece = max(1, (256 << 20) // (4 * int(meta["d"])))
    for r0 in range(0, vec.shape[0], piece):
        eng.add_rows(np.ascontiguousarray(vec[r0:r0 + piece]), r0)
    eng.buckets_end()
    li._engine, li._engine_key = eng, ("disk", os.path.abspath(path))
    return li, [int(v) for v in meta["n_categories"]]
