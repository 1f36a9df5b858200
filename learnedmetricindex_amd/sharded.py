"""Bucket-sharded multi-GPU search (no counterpart in the reference).

One process per GPU.  Bucket b lives on rank owner[b]; the MLP weights and the query batch are
replicated; every rank scans only the (query, rank) pairs whose bucket it owns, then an all-gather
(RCCL over xGMI; `gloo` in the CPU tests) moves every rank's `[dists | ids | keys]` block of nq*k*12
bytes and a merge kernel orders the union by (distance, bucket rank, position) -- the same total
order the single-GPU merge uses, so results are byte-identical for every world size.
Two layouts of the routing step (`ShardedSearcher(shard_inference=...)`):
  * False: every rank runs the MLP on the whole batch -- ONE collective per search (the all-gather above);
  * True (default for world > 1): every rank routes its 1/world slice of the batch and the bucket order
    (nq*nb*4 bytes) is all-gathered first -- TWO small collectives, 1/world of the MLP work per rank.
`ReplicaSearcher` is the other mode SURVEY section 8e names: QUERY-sharded replicas -- every rank holds the whole
index and answers its 1/world slice of the batch; one all-gather of [dists | ids | bucket order] rows.  A query's
answer does not depend on the rest of its batch, so the result is again byte-identical for every world size.  It
pays off when a rank's slice still fills the query tiles (>= ~350 queries per bucket and rank: batches of >= 80 k
at 8 ranks on the 10M benchmark); at 10 k queries bucket-sharding is 2-3 x faster.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


def assign_buckets(sizes, world: int, weights=None) -> np.ndarray:
    """owner[L]: greedy longest-processing-time assignment of buckets to ranks.

    `weights` (default: sizes) is the expected scan work of a bucket; ties and zero-weight buckets
    are spread round-robin so that every rank's result is deterministic and identical on all ranks."""
    sizes = np.asarray(sizes, dtype=np.int64)
    w = sizes.astype(np.float64) if weights is None else np.asarray(weights, dtype=np.float64)
    owner = np.zeros(sizes.shape[0], dtype=np.int32)
    load = np.zeros(world, dtype=np.float64)
    count = np.zeros(world, dtype=np.int64)
    for b in np.argsort(-w, kind="stable"):
        r = int(np.lexsort((np.arange(world), count, load))[0])  # least load, then fewest buckets, then rank
        owner[b] = r
        load[r] += w[b]
        count[r] += 1
    return owner


def estimate_bucket_work(index, sample_nav_t, nb: int, sizes) -> np.ndarray:
    """Expected scan work of every bucket = its rows x the queries it will receive, the second factor
    estimated at build time by routing a sample of the DATA through the MLP like a query batch (top-nb).
    `index` needs its MLP set (`set_mlp`), not its buckets.  Deterministic: every rank computes the same
    weights for `assign_buckets` without any knowledge of the queries.  (Weights sizes**2 left one of eight
    ranks with 1.7x the mean work on the 10M x 768 benchmark; these are within 2 %.)"""
    import torch

    sizes = np.asarray(sizes, dtype=np.float64)
    probe = torch.empty((sample_nav_t.shape[0], nb), dtype=torch.int32, device=sample_nav_t.device)
    index.mlp_topk_device(sample_nav_t.contiguous(), nb, probe)
    torch.cuda.synchronize(sample_nav_t.device)
    routed = np.bincount(probe.cpu().numpy().ravel(), minlength=sizes.shape[0]).astype(np.float64)
    return sizes * (routed[: sizes.shape[0]] + 1.0)


def pack_block(xp, dists, ids, keys):
    """[3, nq, kout] int32 block: float32 distance bits | uint32 ids | uint32 keys (xp: numpy or torch)."""
    if xp is np:
        return np.stack([dists.view(np.int32), ids.view(np.int32), keys.view(np.int32)])
    import torch

    return torch.stack([dists.view(torch.int32), ids.view(torch.int32), keys.view(torch.int32)])


def _all_gather_cat(t, world: int, group=None):
    """all_gather_into_tensor along dim 0.  RCCL ("nccl") takes device tensors as they are; under `gloo`
    (CPU rehearsals of the multi-rank path, also with tensors living on a GPU) the payload is staged on the host."""
    import torch
    import torch.distributed as dist

    t = t.contiguous()
    if t.is_cuda and dist.get_backend(group) == "gloo":
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
        dist.all_gather_into_tensor(out, t.cpu(), group=group)
        return out.to(t.device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def all_gather_blocks(block, world: int, group=None):
    """The one collective of the path: [world, 3, nq, kout] <- all_gather(block)."""
    import torch
    import torch.distributed as dist

    shape = tuple(block.shape)
    out = _all_gather_cat(block, world, group)  # rank-major concatenation
    return out.view((world,) + shape)


def row_slice(nq: int, rank: int, world: int):
    """(per, lo, hi): rank's contiguous share [lo, hi) of nq rows in equal slices of `per` rows (the last ones short or empty)."""
    per = -(-nq // world)
    lo = min(rank * per, nq)
    return per, lo, min(lo + per, nq)


def all_gather_rows(local, nq: int, world: int, group=None):
    """[nq, ...] <- the ranks' [per, ...] slices concatenated in rank order (rows past a rank's share are padding)."""
    import torch
    import torch.distributed as dist

    per = local.shape[0]
    assert per * world >= nq
    return _all_gather_cat(local, world, group)[:nq]


def merge_blocks_numpy(gathered: np.ndarray, kout: int):
    """Host restatement of merge_gathered_kernel, used by the CPU (gloo) tests only."""
    world, three, nq, kk = gathered.shape
    assert three == 3 and kk == kout
    d = gathered[:, 0].view(np.float32).transpose(1, 0, 2).reshape(nq, world * kout).astype(np.float64)
    i = gathered[:, 1].view(np.uint32).transpose(1, 0, 2).reshape(nq, world * kout)
    k = gathered[:, 2].view(np.uint32).transpose(1, 0, 2).reshape(nq, world * kout).astype(np.int64)
    src = np.broadcast_to(np.repeat(np.arange(world), kout)[None, :], k.shape)
    order = np.lexsort((src, k, d), axis=1)[:, :kout]
    return np.take_along_axis(d, order, 1).astype(np.float32), np.take_along_axis(i, order, 1)


class ShardedSearcher:
    """Device-side driver of the sharded search for one rank (torch tensors in, torch tensors out).

    `shard_inference` (default on for world > 1): every rank runs the MLP on its 1/world slice of the query
    batch and the bucket order is all-gathered (nq*nb*4 bytes), instead of every rank routing the whole batch
    (at 8 ranks the replicated MLP was a sixth of a rank's step).  `calls_per_search` tells a caller that
    averages `Index.timings_mean()` how many C-ABI calls one search makes."""

    def __init__(self, index, rank: int, world: int, group=None, shard_inference: bool = True, lib_comm=None):
        """`lib_comm`: an ncclComm_t from `Index.comm_init` -- the result exchange then runs inside the library
        (`lmi_allgather_merge`: ncclAllGather + merge kernel on the handle's stream) instead of torch.distributed."""
        self.index, self.rank, self.world, self.group = index, rank, world, group
        self.lib_comm = lib_comm
        self.shard_inference = bool(shard_inference) and world > 1
        self.calls_per_search = 2 if self.shard_inference else 1
        self._buf = None
        self._bo_loc = None
        # time_collectives: every collective of a search is bracketed by two events on the current stream (device tensors) -- what a
        # rank really WAITS for each exchange, the other ranks' lateness included; read with collective_ms() after a synchronize
        self.time_collectives = False
        self._coll = {"bucket_order_allgather": [], "result_allgather_merge": []}

    def _timed(self, name, fn, on_cuda: bool):
        if not self.time_collectives or self.world == 1:
            return fn()
        import time

        import torch

        if on_cuda:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn()
            b.record()
            self._coll[name].append((a, b))
        else:
            t0 = time.perf_counter()
            out = fn()
            self._coll[name].append((time.perf_counter() - t0) * 1e3)
        return out

    def collective_ms(self, reset: bool = True):
        """{collective: mean exposed ms per search} since the last reset (call after torch.cuda.synchronize())."""
        out = {}
        for name, items in self._coll.items():
            vals = [it if isinstance(it, float) else it[0].elapsed_time(it[1]) for it in items]
            out[name] = round(float(np.mean(vals)), 4) if vals else None
            if reset:
                items.clear()
        return out

    def _buffers(self, nq: int, nb: int, kout: int, dev):
        import torch

        if self._buf is None or self._buf[0].shape != (3, nq, kout):
            self._buf = (torch.empty((3, nq, kout), dtype=torch.int32, device=dev),
                         torch.empty((nq, kout), dtype=torch.float32, device=dev),
                         torch.empty((nq, kout), dtype=torch.int32, device=dev),
                         torch.empty((nq, nb), dtype=torch.int32, device=dev))
        return self._buf

    def new_route_buffer(self, nq: int, nb: int, dev):
        """[ceil(nq / world), nb] int32 buffer for `route_local` (a pipeline keeps one per batch in flight)."""
        import torch

        per, _, _ = row_slice(nq, self.rank, self.world)
        return torch.full((per, nb), -1, dtype=torch.int32, device=dev)

    def route_local(self, qn_t, nb: int, bo_loc) -> None:
        """This rank's share of the routing: the MLP on its 1/world slice of the batch -> bo_loc[: hi - lo].  No
        collective, so a pipeline may run it for batch i+1 on a stream of its own beside the scan of batch i."""
        assert self.shard_inference
        _, lo, hi = row_slice(qn_t.shape[0], self.rank, self.world)
        if hi > lo:
            self.index.mlp_topk_device(qn_t[lo:hi], nb, bo_loc[: hi - lo])

    def search_routed(self, qs_t, bo_loc, nb: int, k: int):
        """The rest of a sharded search: all-gather of the bucket order, scan of the owned buckets, all-gather + merge."""
        nq = qs_t.shape[0]
        kout = self.index.kout(nb, k)
        blk, out_d, out_i, _ = self._buffers(nq, nb, kout, qs_t.device)
        bo = self._timed("bucket_order_allgather", lambda: all_gather_rows(bo_loc, nq, self.world, self.group), bo_loc.is_cuda)
        self.index.scan_topk_device(qs_t, bo, nb, k, blk[0], blk[1], blk[2])
        return self._exchange(blk, out_d, out_i, bo, nq, kout)

    def _exchange(self, blk, out_d, out_i, bo, nq: int, kout: int):
        import torch

        if self.world == 1 and self.lib_comm is None:
            return blk[0].view(torch.float32), blk[1], bo
        if self.lib_comm is not None:
            self._timed("result_allgather_merge",
                        lambda: self.index.allgather_merge(self.lib_comm, self.rank, self.world, blk[0], blk[1], blk[2], out_d, out_i), blk.is_cuda)
            return out_d, out_i, bo

        def exchange():
            g = all_gather_blocks(blk, self.world, self.group)  # [world, 3, nq, kout]
            plane = nq * kout
            self.index.merge_gathered(g[0, 0], g[0, 1], g[0, 2], self.world, nq, kout, out_d, out_i, world_stride=3 * plane)

        self._timed("result_allgather_merge", exchange, blk.is_cuda)
        return out_d, out_i, bo

    def search(self, qn_t, qs_t, nb: int, k: int):
        nq = qn_t.shape[0]
        kout = self.index.kout(nb, k)
        blk, out_d, out_i, bo = self._buffers(nq, nb, kout, qn_t.device)
        # the three planes of `blk` are written in place by lmi_search / lmi_scan_topk
        if self.shard_inference:
            if self._bo_loc is None or self._bo_loc.shape[1] != nb or self._bo_loc.shape[0] != row_slice(nq, self.rank, self.world)[0]:
                self._bo_loc = self.new_route_buffer(nq, nb, qn_t.device)
            self.route_local(qn_t, nb, self._bo_loc)
            return self.search_routed(qs_t, self._bo_loc, nb, k)
        self.index.search_device(qn_t, qs_t, nb, k, blk[0], blk[1], blk[2], bo)
        return self._exchange(blk, out_d, out_i, bo, nq, kout)


class ReplicaSearcher:
    """Query-sharded replicas: the index is whole on every rank (`set_buckets` / `add_rows` without an `owned`
    mask), rank r answers rows [lo, hi) of the batch (`row_slice`) and ONE all-gather assembles the batch's
    answer: per query [kout distances | kout ids | nb buckets] as int32 words.  No merge step: the slices are
    disjoint.  Same call signature and return value as `ShardedSearcher.search`."""

    calls_per_search = 1
    shard_inference = False   # (HostPipeline: nothing to route ahead of the search call)
    lib_comm = None

    def __init__(self, index, rank: int, world: int, group=None):
        self.index, self.rank, self.world, self.group = index, rank, world, group
        self._buf = None

    def _buffers(self, per: int, nb: int, kout: int, dev):
        import torch

        if self._buf is None or self._buf[0].shape != (per, 2 * kout + nb):
            self._buf = (torch.empty((per, 2 * kout + nb), dtype=torch.int32, device=dev),
                         torch.empty((per, kout), dtype=torch.float32, device=dev),
                         torch.empty((per, kout), dtype=torch.int32, device=dev),
                         torch.empty((per, kout), dtype=torch.int32, device=dev),
                         torch.empty((per, nb), dtype=torch.int32, device=dev))
        return self._buf

    def search(self, qn_t, qs_t, nb: int, k: int):
        import torch

        nq = qn_t.shape[0]
        kout = self.index.kout(nb, k)
        per, lo, hi = row_slice(nq, self.rank, self.world)
        row, d, i, keys, bo = self._buffers(per, nb, kout, qn_t.device)
        if hi > lo:
            n = hi - lo
            self.index.search_device(qn_t[lo:hi], qs_t[lo:hi], nb, k, d[:n], i[:n], keys[:n], bo[:n])
        if self.world == 1:
            return d[:nq], i[:nq], bo[:nq]
        row[:, :kout] = d.view(torch.int32)
        row[:, kout:2 * kout] = i
        row[:, 2 * kout:] = bo
        g = all_gather_rows(row, nq, self.world, self.group)   # the one collective
        return g[:, :kout].contiguous().view(torch.float32), g[:, kout:2 * kout].contiguous(), g[:, 2 * kout:].contiguous()
