"""oracle/cpu_baseline.py -- TEST/BENCH INFRASTRUCTURE, NOT PRODUCT CODE.

CPU baselines timed by bench.py's `cpu_baseline` leg beside the MI355X numbers.
Only bench.py and tests/ import this; nothing under learnedmetricindex_amd/ does.

Two restatements of the reference's query path (reference files under /root/reference/search/li/):

* `reference_structured` -- the reference's own loop structure, operation for operation: the per-rank
  `data_navigation.groupby(category_L1)` that materialises every group (LearnedIndex.py:350), `filter_path_idxs`
  (utils.py:61-65, call :353), the label-based `data_search.loc[g.index].to_numpy()` gather copy of every
  visited bucket on every rank (:357), an inner-product k-NN over the bucket (`faiss.knn`, :360-365; here
  BLAS sgemm + partial sort, faiss 1.7.4's published algorithm for METRIC_INNER_PRODUCT brute force),
  `1 - similarity` (:368), label mapping (:370) and the hstack + stable argsort merge (:125-146).  This is what
  `search.py` costs on a CPU -- the survey measured 75 % of it in pandas data movement.
* `best_effort` -- what a CPU implementation with this build's data layout can do: bucket-contiguous slab,
  torch-CPU `matmul` + `topk` per bucket for all ranks at once, no pandas, no per-call copies.

Both return (dists f64[nq,k], ids u32[nq,k]) like LearnedIndex.search so that the caller can compare them
with the GPU's answer (ids equal except inside float32 near-ties: the summation order of BLAS differs from
the canonical chain -- the bit-exact checker is oracle/lmi_oracle.c, not these).
"""
from __future__ import annotations

import time
from typing import Dict, Sequence, Tuple

import numpy as np

K_PER_BUCKET = 10  # LearnedIndex.py:334


def mlp_order_numpy(layers, queries: np.ndarray, nb: int) -> np.ndarray:
    """model.py:97-99, 232-239 + LearnedIndex.py:197-214 in numpy/BLAS: top-nb classes per query."""
    h = queries
    for li, (W, b) in enumerate(layers):
        h = h @ W.T + b
        if li + 1 < len(layers):
            h = np.maximum(h, 0)
    return np.argsort(-h, axis=1, kind="stable")[:, :nb].astype(np.int32)


def _knn_blas(xq: np.ndarray, xb: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """faiss.knn(xq, xb, k, METRIC_INNER_PRODUCT) stand-in: sgemm + partial selection, padded like faiss."""
    sim = xq @ xb.T
    n = xb.shape[0]
    if n > k:
        part = np.argpartition(-sim, k - 1, axis=1)[:, :k]
        ps = np.take_along_axis(sim, part, 1)
        o = np.argsort(-ps, axis=1, kind="stable")
        return np.take_along_axis(ps, o, 1), np.take_along_axis(part, o, 1).astype(np.int64)
    o = np.argsort(-sim, axis=1, kind="stable")
    D = np.full((xq.shape[0], k), -np.finfo(np.float32).max, dtype=np.float32)
    I = np.full((xq.shape[0], k), -1, dtype=np.int64)
    D[:, :n], I[:, :n] = np.take_along_axis(sim, o, 1), o
    return D, I


def reference_structured(data_navigation, data_search, queries_search: np.ndarray, bucket_order: np.ndarray,
                         data_prediction: np.ndarray, k: int = 10, deadline_s: float = None) -> Tuple[np.ndarray, np.ndarray, Dict[str, float]]:
    """The reference's search loop on pandas frames (see the module docstring for the line map).
    `data_navigation`/`data_search`: DataFrames with the objects' labels as index (distinct objects);
    `bucket_order` int32 [nq, nb] from the MLP; `data_prediction` int64 [N] bucket of every object.
    `deadline_s` (bench only): stop after the first rank that ends past it; `t["ranks_done"]` then says how many of the nb
    ranks ran (every rank costs the same data movement: the caller scales) and the merged result covers those ranks only."""
    t = {"groupby": 0.0, "gather": 0.0, "knn": 0.0, "merge": 0.0, "ranks_done": 0.0}
    t_start = time.perf_counter()
    nq, nb = bucket_order.shape
    data_navigation["category_L1"] = data_prediction                       # :101-104
    dists_final = anns_final = None
    for r in range(nb):                                                     # :107
        if deadline_s is not None and r > 0 and time.perf_counter() - t_start > deadline_s:
            break
        t["ranks_done"] += 1
        nns = np.zeros((nq, K_PER_BUCKET), dtype=np.uint32)                # :340-341
        dists = np.full((nq, K_PER_BUCKET), np.inf, dtype=float)
        t0 = time.perf_counter()
        for path, g in data_navigation.groupby(["category_L1"]):          # :350 (materialises g)
            t["groupby"] += time.perf_counter() - t0
            obj = g.index
            rel = np.where(bucket_order[:, r] == path[0])[0]               # utils.py:61-65
            if obj.shape[0] != 0 and rel.shape[0] != 0:
                t1 = time.perf_counter()
                data = data_search.loc[obj].to_numpy()                     # :357
                t["gather"] += time.perf_counter() - t1
                t1 = time.perf_counter()
                sim, idx = _knn_blas(queries_search[rel], data, K_PER_BUCKET)   # :360-365
                t["knn"] += time.perf_counter() - t1
                nns[rel] = obj.to_numpy()[idx]                             # :370 (-1 -> last label, Q4)
                dists[rel] = 1 - sim                                       # :368
            t0 = time.perf_counter()
        t1 = time.perf_counter()
        if anns_final is None:
            anns_final, dists_final = nns, dists
        else:                                                              # :125-146
            anns_final = np.hstack((anns_final, nns))
            dists_final = np.hstack((dists_final, dists))
            o = dists_final.argsort(kind="stable", axis=1)[:, :k]
            dists_final = np.take_along_axis(dists_final, o, 1)
            anns_final = np.take_along_axis(anns_final, o, 1)
        t["merge"] += time.perf_counter() - t1
    data_navigation.drop("category_L1", axis=1, inplace=True)              # :153-157
    return dists_final, anns_final, t


def best_effort(slab, offsets: np.ndarray, ids: np.ndarray, layers: Sequence, queries, nb: int, k: int, threads: int):
    """Bucket-contiguous slab + torch-CPU matmul/topk (all ranks of a bucket in one product), `threads` threads.
    slab: torch.FloatTensor [N,d] (host), offsets [L+1], ids u32 [N] in slab order, queries torch [nq,d].
    Returns (dists f64[nq,k], ids u32[nq,k], bucket_order i32[nq,nb], seconds)."""
    import torch

    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    with torch.no_grad():
        h = queries
        for li, (W, b) in enumerate(layers):
            h = torch.addmm(torch.from_numpy(b), h, torch.from_numpy(W).T)
            if li + 1 < len(layers):
                h = torch.relu(h)
        order = h.topk(nb, dim=1).indices                                   # [nq, nb]
        nq = queries.shape[0]
        rank_s = torch.full((nq, nb, K_PER_BUCKET), -torch.finfo(torch.float32).max)
        rank_i = torch.zeros((nq, nb, K_PER_BUCKET), dtype=torch.int64)
        visited = torch.zeros((nq, nb), dtype=torch.bool)
        flat = order.reshape(-1)
        perm = torch.argsort(flat, stable=True)
        counts = torch.bincount(flat, minlength=offsets.shape[0] - 1)
        starts = torch.cumsum(counts, 0) - counts
        ids_t = torch.from_numpy(ids.astype(np.int64))
        for b in torch.nonzero(counts).flatten().tolist():
            lo, hi = int(offsets[b]), int(offsets[b + 1])
            if hi == lo:
                continue
            slots = perm[starts[b]: starts[b] + counts[b]]                   # q*nb + r
            qi, ri = slots // nb, slots % nb
            sim = queries[qi] @ slab[lo:hi].T                               # [m_b, n_b]
            kk = min(K_PER_BUCKET, hi - lo)
            v, i = sim.topk(kk, dim=1)
            rank_s[qi, ri, :kk] = v
            rank_i[qi, ri, :kk] = ids_t[lo + i]
            if kk < K_PER_BUCKET:                                           # faiss padding -> last label (Q4)
                rank_i[qi, ri, kk:] = ids_t[hi - 1]
            visited[qi, ri] = True
        d = 1 - rank_s
        d[~visited] = float("inf")                                          # :340-341
        rank_i[~visited] = 0
        d = d.reshape(nq, nb * K_PER_BUCKET)
        o = torch.sort(d, dim=1, stable=True).indices[:, : (K_PER_BUCKET if nb == 1 else k)]
        out_d = torch.gather(d, 1, o).double().numpy()
        out_i = torch.gather(rank_i.reshape(nq, -1), 1, o).numpy().astype(np.uint32)
    return out_d, out_i, order.numpy().astype(np.int32), time.perf_counter() - t0


def best_effort_bucket_parallel(slab, offsets: np.ndarray, ids: np.ndarray, layers: Sequence, queries, nb: int, k: int, workers: int):
    """`best_effort` with the parallelism turned round: `workers` Python threads each take whole buckets (one single-threaded
    matmul + topk per bucket; torch releases the GIL inside its kernels) instead of every product being split over the BLAS
    threads.  Same arithmetic and results as `best_effort`; returns the same tuple."""
    import torch
    from concurrent.futures import ThreadPoolExecutor

    old = torch.get_num_threads()
    t0 = time.perf_counter()
    try:
        with torch.no_grad():
            torch.set_num_threads(workers)
            h = queries
            for li, (W, b) in enumerate(layers):
                h = torch.addmm(torch.from_numpy(b), h, torch.from_numpy(W).T)
                if li + 1 < len(layers):
                    h = torch.relu(h)
            order = h.topk(nb, dim=1).indices
            torch.set_num_threads(1)
            nq = queries.shape[0]
            rank_s = torch.full((nq, nb, K_PER_BUCKET), -torch.finfo(torch.float32).max)
            rank_i = torch.zeros((nq, nb, K_PER_BUCKET), dtype=torch.int64)
            visited = torch.zeros((nq, nb), dtype=torch.bool)
            flat = order.reshape(-1)
            perm = torch.argsort(flat, stable=True)
            counts = torch.bincount(flat, minlength=offsets.shape[0] - 1)
            starts = torch.cumsum(counts, 0) - counts
            ids_t = torch.from_numpy(ids.astype(np.int64))

            def one(b):
                lo, hi = int(offsets[b]), int(offsets[b + 1])
                if hi == lo:
                    return
                slots = perm[starts[b]: starts[b] + counts[b]]
                qi, ri = slots // nb, slots % nb
                sim = queries[qi] @ slab[lo:hi].T
                kk = min(K_PER_BUCKET, hi - lo)
                v, i = sim.topk(kk, dim=1)
                rank_s[qi, ri, :kk] = v          # (distinct (query, rank) slots per bucket: the threads write disjoint rows)
                rank_i[qi, ri, :kk] = ids_t[lo + i]
                if kk < K_PER_BUCKET:
                    rank_i[qi, ri, kk:] = ids_t[hi - 1]
                visited[qi, ri] = True

            todo = torch.nonzero(counts).flatten().tolist()
            todo.sort(key=lambda b: -(int(counts[b]) * int(offsets[b + 1] - offsets[b])))   # heaviest first
            with ThreadPoolExecutor(max_workers=workers) as ex:
                list(ex.map(one, todo))
            torch.set_num_threads(workers)
            d = 1 - rank_s
            d[~visited] = float("inf")
            rank_i[~visited] = 0
            d = d.reshape(nq, nb * K_PER_BUCKET)
            o = torch.sort(d, dim=1, stable=True).indices[:, : (K_PER_BUCKET if nb == 1 else k)]
            out_d = torch.gather(d, 1, o).double().numpy()
            out_i = torch.gather(rank_i.reshape(nq, -1), 1, o).numpy().astype(np.uint32)
    finally:
        torch.set_num_threads(old)
    return out_d, out_i, order.numpy().astype(np.int32), time.perf_counter() - t0


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup's CPU quota (a GPU box of the pool shows all
    256 host cores in os.cpu_count() while a one-GPU job is given 16)."""
    import math
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, math.ceil(q / per)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def id_agreement(a: np.ndarray, b: np.ndarray) -> float:
    """fraction of result positions whose ids agree as SETS per query (near-tie swaps do not count)."""
    return float(np.mean([len(set(x) & set(y)) / float(len(x)) for x, y in zip(a, b)]))
