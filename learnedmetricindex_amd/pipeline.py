"""Host-in -> host-out query pipeline (SURVEY.md section 8d defines QPS from host `queries_*` arrays in to host
`(dists, nns)` out, LearnedIndex.py:87, 159).

One batch needs 31 MB of queries over PCIe (10 000 x 768 f32: ~0.55 ms at Gen5 x16) and 0.8 MB of results
back; the search itself is ~6 ms.  `HostPipeline` keeps `depth` batches in flight on two HIP streams so the
upload of batch i+1 runs under the scan of batch i:

    copy-in stream    :  H2D(q[i+1]) ............
    navigation stream :  wait(H2D i+1) -> lmi_mlp_topk(batch i+1) -> event               (`overlap_inference`)
    compute stream    :  wait(nav i) -> lmi_scan_topk(batch i, device pointers) -> D2H(dists, ids [, bucket order]) -> event

The MLP of the next batch depends on nothing the scan of the current one produces.  On a stream of its own its blocks
start as soon as CUs come free: in the tail of the scan's persistent kernels and of the re-rank, instead of after them
(0.18 ms of a 5.7 ms step otherwise spent with the chip to itself).

The 1 MB result download is a KERNEL on the compute stream that stores into the pinned result buffers
(`lmi_copy_out`, ~30 us): every `hipMemcpyAsync` D2H form tried (own stream behind an event wait; in the compute
stream) blocked the submitting host thread for 15-20 ms every few batches behind the queued search kernels
(ROCm 7.2, seen by timing the calls) and cost up to 8 % of the throughput.

`use_graph`: the whole batch -- upload, MLP, scan, download, three streams -- is captured ONCE per (slot, pinned source batch) as a
hipGraph and replayed with one launch per batch: a 1 000-query search is ~0.18 ms of GPU work behind ~0.26 ms of Python and HIP API
calls (a dozen launches and event operations), so small batches are host-bound without it.  Nothing in the captured work depends on a
per-call kernel argument (the routing kernels' call tag is a device word, lmi_front.h); `capture()` pre-builds the graphs outside a timed
region.  Pageable inputs go through the slot's pinned staging buffer (one graph per slot).

Host buffers are pinned (hipHostMalloc through torch): `submit` copies a pageable numpy batch into the
slot's pinned staging buffer (a CPU memcpy that overlaps the GPU's work), or takes a pinned torch tensor as
it is.  PyTorch is used for pinned memory, streams and events only; the search is the C ABI's `lmi_search`.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

_STREAMS: dict = {}   # device index -> the four streams every pipeline of this process uses there


class HostPipeline:
    def __init__(self, index, nq: int, d_nav: int, d_search: int, nb: int, k: int = 10, depth: int = 2,
                 device: Optional[int] = None, same_queries: bool = False, want_bucket_order: bool = False,
                 search_fn=None, overlap_inference: bool = True, two_handles: bool = False, sharded=None, use_graph: bool = False,
                 direct_out: bool = False, native_submit: bool = True, share_streams: bool = True):
        """`search_fn(qn_dev, qs_dev) -> (dists_t, ids_t, bucket_order_t)`: optional replacement of the single-GPU
        `lmi_search` call, run on the compute stream (the bucket-sharded searcher of sharded.py, whose collectives
        then run on that stream too); its output tensors may be reused by its next call."""
        import torch

        self.index, self.nq, self.nb, self.k, self.depth = index, int(nq), int(nb), int(k), int(depth)
        self.kout = index.kout(nb, k)
        self.same = bool(same_queries) and d_nav == d_search   # navigation and scan vectors are one array
        self.want_bo = bool(want_bucket_order)
        self.search_fn = search_fn
        # two C-ABI calls per batch (lmi_mlp_topk on the navigation stream, lmi_scan_topk on the compute stream) instead
        # of one lmi_search: `calls_per_batch` tells a reader of lmi_timings_mean how many calls make up one batch
        # two_handles: batches alternate between the index handle and a clone of it (`lmi_clone_view`: same index memory, own
        # workspaces), each with its compute stream -- every phase of batch i+1 may start in the tails of batch i's kernels
        # sharded: a sharded.ShardedSearcher (instead of search_fn).  With its inference sharded, the rank's MLP slice of
        # batch i+1 (no collective) runs on the navigation stream; the collectives and the scan stay on the compute stream
        self.sharded = sharded
        assert sharded is None or search_fn is None
        self.sh_overlap = sharded is not None and bool(overlap_inference) and sharded.shard_inference
        if sharded is not None and not self.sh_overlap:
            search_fn = lambda qn, qs: sharded.search(qn, qs, self.nb, self.k)   # noqa: E731
            self.search_fn = search_fn
        self.two = bool(two_handles) and search_fn is None and sharded is None
        self.overlap = bool(overlap_inference) and search_fn is None and not self.two and sharded is None
        self.calls_per_batch = 2 if self.overlap else 1
        dev = torch.device("cuda", index.device if device is None else device)
        self.dev = dev
        # the pipelines of a process share one set of streams per device: every new HIP stream is bound to one of the runtime's few
        # hardware queues (GPU_MAX_HW_QUEUES, 4 by default) by creation order, and the pipeline built sixth in a process measured
        # 15 % slower at 10M x 45 than the same pipeline built first (profiles/r05_stream_queues.txt) -- its three streams no longer
        # overlapped the way the first three do
        key = (dev.index if dev.index is not None else torch.cuda.current_device())
        if share_streams:
            pool = _STREAMS.setdefault(key, [])
            while len(pool) < 4:
                pool.append(torch.cuda.Stream(dev))
            self.s_in, self.s_run, self.s_nav, s2 = pool
        else:
            self.s_in, self.s_run, self.s_nav, s2 = (torch.cuda.Stream(dev) for _ in range(4))
        index.set_stream(self.s_run.cuda_stream)
        self.handles = [(index, self.s_run)]
        if self.two:
            twin = index.clone_view()
            twin.set_stream(s2.cuda_stream)
            self.handles.append((twin, s2))
        f32, i32 = torch.float32, torch.int32
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)   # noqa: E731
        pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)   # noqa: E731
        # direct_out: the search's last kernels store (dists, ids) straight into the pinned host buffers (device-accessible memory) -- no
        # download kernel behind the search (1 MB over PCIe: 18 us of a 0.56-ms search at 10M x 45); single-GPU forms only
        self.direct_out = bool(direct_out) and search_fn is None and sharded is None
        # native_submit: a batch's stream / event / launch sequence as ONE C call (lmi_pipeline_submit) instead of a dozen Python-level
        # operations (0.2-0.3 ms of host time per batch: more than the GPU needs for a 1 000-query search); single-handle, single-GPU forms
        self.native = bool(native_submit) and search_fn is None and sharded is None and not self.two
        # hipGraph replay: single-GPU forms only (a collective inside a capture is RCCL's business, not rehearsed here)
        self.use_graph = bool(use_graph) and search_fn is None and sharded is None and not self.two
        self.slots = []
        for _ in range(self.depth):
            s = dict(qn_h=pin((nq, d_nav), f32), qn_d=mk((nq, d_nav), f32),
                     d_d=mk((nq, self.kout), f32), i_d=mk((nq, self.kout), i32), bo_d=mk((nq, nb), i32),
                     d_h=pin((nq, self.kout), f32), i_h=pin((nq, self.kout), i32), bo_h=pin((nq, nb), i32),
                     ev_in=torch.cuda.Event(), ev_nav=torch.cuda.Event(), ev_out=torch.cuda.Event(), busy=False, graphs={}, warm=False)
            if not self.same:
                s["qs_h"], s["qs_d"] = pin((nq, d_search), f32), mk((nq, d_search), f32)
            if self.sh_overlap:
                s["bo_loc"] = sharded.new_route_buffer(nq, nb, dev)
            self.slots.append(s)
        self.t = 0

    def _enqueue(self, s, qn_src, qs_src, s_run, index, origin=None) -> None:
        """The batch's GPU work on the three streams.  `origin`: the capturing stream (graph capture) -- the upload stream forks from it
        and the compute stream joins back into it, so that everything is part of the capture."""
        import torch

        with torch.cuda.stream(self.s_in):
            if origin is not None:
                self.s_in.wait_stream(origin)
            s["qn_d"].copy_(qn_src, non_blocking=True)
            if not self.same:
                s["qs_d"].copy_(qs_src, non_blocking=True)
            s["ev_in"].record(self.s_in)
        if self.overlap:
            with torch.cuda.stream(self.s_nav):
                self.s_nav.wait_event(s["ev_in"])
                self.index.set_stream(self.s_nav.cuda_stream)
                self.index.mlp_topk_device(s["qn_d"], self.nb, s["bo_d"])
                s["ev_nav"].record(self.s_nav)
            self.index.set_stream(self.s_run.cuda_stream)
        if self.sh_overlap:
            with torch.cuda.stream(self.s_nav):
                self.s_nav.wait_event(s["ev_in"])
                self.index.set_stream(self.s_nav.cuda_stream)
                self.sharded.route_local(s["qn_d"], self.nb, s["bo_loc"])
                s["ev_nav"].record(self.s_nav)
            self.index.set_stream(self.s_run.cuda_stream)
        od, oi = ("d_h", "i_h") if self.direct_out else ("d_d", "i_d")   # the search's output rows: pinned host buffers or device buffers
        with torch.cuda.stream(s_run):
            s_run.wait_event(s["ev_nav"] if (self.overlap or self.sh_overlap) else s["ev_in"])
            if self.sh_overlap:
                d_t, i_t, bo_t = self.sharded.search_routed(s["qn_d"] if self.same else s["qs_d"], s["bo_loc"], self.nb, self.k)
            elif self.two:
                index.search_device(s["qn_d"], s["qn_d"] if self.same else s["qs_d"], self.nb, self.k,
                                    s[od], s[oi], None, s["bo_d"])
                d_t, i_t, bo_t = s[od], s[oi], s["bo_d"]
            elif self.overlap:
                self.index.scan_topk_device(s["qn_d"] if self.same else s["qs_d"], s["bo_d"], self.nb, self.k, s[od], s[oi])
                d_t, i_t, bo_t = s[od], s[oi], s["bo_d"]
            elif self.search_fn is None:
                self.index.search_device(s["qn_d"], s["qn_d"] if self.same else s["qs_d"], self.nb, self.k,
                                         s[od], s[oi], None, s["bo_d"])
                d_t, i_t, bo_t = s[od], s[oi], s["bo_d"]
            else:
                d_t, i_t, bo_t = self.search_fn(s["qn_d"], s["qn_d"] if self.same else s["qs_d"])
            # ONE kernel storing to the pinned buffers (see the module docstring; three launches were 25 us of a 0.6-ms search)
            pairs = ([] if self.direct_out else [(s["d_h"], d_t), (s["i_h"], i_t)]) + ([(s["bo_h"], bo_t)] if self.want_bo else [])
            if pairs:
                index.copy_out_many(pairs)
            if origin is not None:
                origin.wait_stream(s_run)

    def _stage(self, src, pinned):
        import torch

        if isinstance(src, torch.Tensor) and src.is_pinned():
            return src                                   # already DMA-able: no staging copy
        pinned.copy_(torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32)) if isinstance(src, np.ndarray) else src)
        return pinned

    def _graph_for(self, s, qn_src, qs_src, s_run, index):
        """The slot's captured batch for these pinned sources (built on first use: an eager batch first, so that the library has made
        its allocations -- nothing may allocate or synchronise inside a capture).  With the MLP on the navigation stream the batch is
        TWO graphs: (upload + MLP) replayed on the navigation stream and (scan + download) on the compute stream behind it -- graph
        launches on one stream run strictly one after the other, so a single graph per batch would serialise the upload and the MLP
        of batch i + 1 behind the scan of batch i (measured: 0.32 ms per 1 000-query batch instead of 0.20)."""
        import torch

        key = (qn_src.data_ptr(), 0 if qs_src is None else qs_src.data_ptr())
        g = s["graphs"].get(key)
        if g is None:
            if not s["warm"]:
                self._enqueue(s, qn_src, qs_src, s_run, index)
                s["warm"] = True
            torch.cuda.synchronize(self.dev)
            if self.overlap:
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, stream=self.s_nav):
                    s["qn_d"].copy_(qn_src, non_blocking=True)
                    if not self.same:
                        s["qs_d"].copy_(qs_src, non_blocking=True)
                    self.index.set_stream(self.s_nav.cuda_stream)
                    self.index.mlp_topk_device(s["qn_d"], self.nb, s["bo_d"])
                self.index.set_stream(self.s_run.cuda_stream)
                torch.cuda.synchronize(self.dev)
                with torch.cuda.graph(gb, stream=self.s_run):
                    self.index.scan_topk_device(s["qn_d"] if self.same else s["qs_d"], s["bo_d"], self.nb, self.k, s["d_d"], s["i_d"])
                    self.index.copy_out_many([(s["d_h"], s["d_d"]), (s["i_h"], s["i_d"])] + ([(s["bo_h"], s["bo_d"])] if self.want_bo else []))
                g = (ga, gb)
            else:
                g1 = torch.cuda.CUDAGraph()
                cap = torch.cuda.Stream(self.dev)
                with torch.cuda.graph(g1, stream=cap):
                    self._enqueue(s, qn_src, qs_src, s_run, index, origin=cap)
                g = (g1,)
            s["graphs"][key] = g
        return g

    def capture(self, queries_nav, queries_search=None) -> None:
        """Pre-builds the graphs of one pinned source batch for every slot (outside a timed region; `use_graph` only)."""
        if not self.use_graph:
            return
        self.drain()
        for n, s in enumerate(self.slots):
            index, s_run = self.handles[n % len(self.handles)]
            qn_src = self._stage(queries_nav, s["qn_h"])
            qs_src = None if self.same else self._stage(queries_nav if queries_search is None else queries_search, s["qs_h"])
            self._graph_for(s, qn_src, qs_src, s_run, index)

    def submit(self, queries_nav, queries_search=None) -> int:
        """Enqueues one batch (numpy arrays or pinned CPU torch tensors, [nq, d]); returns its ticket.
        Blocks only if the slot's previous batch (`depth` submits ago) has not finished."""
        import torch

        s = self.slots[self.t % self.depth]
        if s["busy"]:
            s["ev_out"].synchronize()
        s["busy"] = True
        qn_src = self._stage(queries_nav, s["qn_h"])
        qs_src = None if self.same else self._stage(queries_nav if queries_search is None else queries_search, s["qs_h"])
        index, s_run = self.handles[self.t % len(self.handles)]
        if self.native and not self.use_graph:
            if not s["warm"]:   # torch creates an event's handle with its first record
                for e_ in ("ev_in", "ev_nav", "ev_out"):
                    s[e_].record(self.s_in)
                s["warm"] = True
            od, oi = ("d_h", "i_h") if self.direct_out else ("d_d", "i_d")
            self.index.pipeline_submit(self.s_in.cuda_stream, self.s_nav.cuda_stream, s_run.cuda_stream, s["ev_in"].cuda_event, s["ev_nav"].cuda_event,
                                       s["ev_out"].cuda_event, qn_src, qs_src, s["qn_d"], None if self.same else s["qs_d"], self.nb, self.k,
                                       s[od], s[oi], s["bo_d"], s["bo_h"] if self.want_bo else None, self.overlap)
            if not self.direct_out:
                with torch.cuda.stream(s_run):
                    index.copy_out_many([(s["d_h"], s["d_d"]), (s["i_h"], s["i_d"])])
                    s["ev_out"].record(s_run)
        elif self.use_graph:
            g = self._graph_for(s, qn_src, qs_src, s_run, index)
            if len(g) == 2:   # (upload + MLP) on the navigation stream, (scan + download) on the compute stream behind it
                with torch.cuda.stream(self.s_nav):
                    g[0].replay()
                    s["ev_nav"].record(self.s_nav)
            with torch.cuda.stream(s_run):
                if len(g) == 2:
                    s_run.wait_event(s["ev_nav"])
                g[-1].replay()
                s["ev_out"].record(s_run)
        else:
            self._enqueue(s, qn_src, qs_src, s_run, index)
            with torch.cuda.stream(s_run):
                s["ev_out"].record(s_run)
        self.t += 1
        return self.t - 1

    def result(self, ticket: int) -> Tuple[np.ndarray, np.ndarray]:
        """(dists f32[nq,kout], ids u32[nq,kout]) of a batch, as views of its pinned output slot (valid until
        the slot is reused `depth` submits later).  Waits for that batch only."""
        assert self.t - self.depth <= ticket < self.t, "ticket no longer (or not yet) in the ring"
        s = self.slots[ticket % self.depth]
        s["ev_out"].synchronize()
        return s["d_h"].numpy(), s["i_h"].numpy().view(np.uint32)

    def bucket_order(self, ticket: int) -> np.ndarray:
        """The batch's bucket order [nq, nb].  `want_bucket_order=True` downloads it with every batch (a kernel behind the search: 18 us of a
        0.5-ms step at 10M x 45); without it -- the reference's `search` returns no bucket order -- the single-GPU forms keep it in the
        slot's device buffer and this call fetches it on demand (valid until the slot is reused)."""
        assert self.t - self.depth <= ticket < self.t, "ticket no longer (or not yet) in the ring"
        s = self.slots[ticket % self.depth]
        s["ev_out"].synchronize()
        if self.want_bo:
            return s["bo_h"].numpy()
        assert self.search_fn is None and self.sharded is None, "bucket order on demand: single-GPU forms only (pass want_bucket_order=True)"
        return s["bo_d"].cpu().numpy()

    def drain(self) -> None:
        for s in self.slots:
            if s["busy"]:
                s["ev_out"].synchronize()
                s["busy"] = False
