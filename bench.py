#!/usr/bin/env python3
"""bench.py -- queries/sec of the LearnedMetricIndex query hot path on MI355X.

A "step" = one LearnedIndex.search of a whole query batch: MLP forward -> top-n_buckets -> routing -> bucket
scan -> merge.  The index is resident in HBM before the timed region (it is built once, like the reference's DataFrames).
EXACTLY K steps are timed twice, each between barrier + synchronize: (i) with the query batches already in HBM and the results left
there -- `value`, `ms_per_step`, `roofline`, `phases_ms` (`value_loop: "resident"`); (ii) HOST-IN -> HOST-OUT, the reference's own
boundary: every step takes its batch from pinned host memory and leaves (dists, ids) in pinned host memory, the
upload of batch i+1 and the MLP of batch i+1 overlap the search of batch i (learnedmetricindex_amd/pipeline.py) -- the PCIe-inclusive
rate, reported beside it as `host_to_host` (`--value-loop host` makes it `value`, as rounds 1-4 did).
Default workload = BASELINE.json configs[1]: 10M x 768 synthetic unit-norm vectors (LAION-10M shape),
120 leaves, MLP-4 (768->512->120), top-4 buckets, 10k queries, 1 x MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: the index is bucket-sharded over the ranks (each rank ingests only the rows it owns), every rank
answers the same batch on its own buckets and an RCCL all-gather + merge kernel produces the result (total
work fixed: "strong").  Rank 0 prints ONE JSON line (see the keys at the bottom).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

CONFIGS = {
    # name: (N, d, leaves, hidden model, n_buckets, nq)
    "c1": dict(n=100_000, d=768, leaves=120, model="MLP-4", nb=4, nq=1_000),
    "c2": dict(n=10_000_000, d=768, leaves=120, model="MLP-4", nb=4, nq=10_000),
    "c4": dict(n=100_000_000, d=768, leaves=1024, model="MLP-4", nb=8, nq=10_000),
    "c5": dict(n=10_000_000, d=45, leaves=256, model="MLP-4", nb=4, nq=10_000),
}
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: BF16/F16 MFMA, dense (never the 2:1-sparse figure)
PEAK_HBM_GBS = 8000.0
CHUNK = 1 << 19  # rows generated / ingested per piece


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def lib_provenance() -> dict:
    """Which library this process LOADED: path, sha256/16 of the .so file, the source hash baked into it at build time
    (lmi_build_info) and the hash of the sources in the working tree.  PMC summaries under profiles/ carry the same fields
    (profiles/summarize.py copies them from the bench line of the profiled run) and are replayed only when they match."""
    from learnedmetricindex_amd import _capi, _srchash

    info = _capi.lib().lmi_build_info().decode()
    built = dict(kv.split("=", 1) for kv in info.split() if "=" in kv)
    return {"path": os.path.relpath(_capi.LIB_PATH, ROOT), "so_sha16": _srchash.file_sha16(_capi.LIB_PATH),
            "built_from_source_sha16": built.get("src"), "tree_source_sha16": _srchash.source_sha16(ROOT), "build_info": info}


class Workload:
    """Synthetic data of SURVEY 8d (L Gaussian clusters, unit-norm rows, per-piece counter seeds on the device)
    + the MLP trained on it + the HBM-resident index.  `sigma` is the cluster noise (1.0 = the survey's
    generator), `zipf` > 0 draws the cluster of an object with probability ~ 1/(1 + c/zipf) (heavy-tailed
    bucket sizes), `centre_scale` < 1 pulls the centres together (overlapping clusters)."""

    def __init__(self, args, cfg, dev, rank, world, local_rank, sigma=1.0, zipf=0.0, centre_scale=1.0, tag="main", exact=None, layers=None, label=None):
        import torch
        import torch.distributed as dist

        from learnedmetricindex_amd import _capi
        from learnedmetricindex_amd.li.model import NeuralNetwork, linear_layers
        from learnedmetricindex_amd.sharded import assign_buckets, estimate_bucket_work

        self.args, self.cfg, self.dev, self.rank, self.world, self.tag = args, cfg, dev, rank, world, tag
        self.exact = args.exact if exact is None else exact   # all-f32 scan_kernel instead of fp16 prefilter + exact re-rank
        N, d, L, nb, nq = cfg["n"], cfg["d"], cfg["leaves"], cfg["nb"], cfg["nq"]
        t_setup = time.time()
        gcpu = torch.Generator().manual_seed(args.seed + (0 if tag == "main" else 7919))
        centres = (torch.randn(L, d, generator=gcpu) * centre_scale).to(dev)
        cdf = None
        if zipf > 0:
            w = 1.0 / (1.0 + torch.arange(L, dtype=torch.float64) / zipf)
            cdf = (w / w.sum()).to(dev, torch.float32)

        def gen_rows(tagno: int, piece: int, n: int):
            g = torch.Generator(device=dev).manual_seed(args.seed * 1_000_003 + tagno * 100_003 + piece)
            if cdf is None:
                a = torch.randint(0, L, (n,), generator=g, device=dev)
            else:
                a = torch.multinomial(cdf, n, replacement=True, generator=g)
            x = centres[a] + sigma * torch.randn(n, d, generator=g, device=dev)
            return torch.nn.functional.normalize(x, dim=1).contiguous()

        self.gen_rows = gen_rows
        pieces = [(p, min(CHUNK, N - p * CHUNK)) for p in range((N + CHUNK - 1) // CHUNK)]
        self.queries = gen_rows(7, 0, nq)  # fresh draws, not members of the set

        # ---- MLP: k-means labels -> Adam/CE (the reference's hyper-parameters: 200 epochs, lr 0.01)
        net = NeuralNetwork(input_dim=d, output_dim=L, lr=0.01, model_type=cfg["model"])
        if layers is not None:   # a second index over the same data with the weights of the first (the exact leg)
            pass
        elif rank == 0:
            torch.manual_seed(args.seed)
            ntr = min(args.train_rows, N)
            xtr = torch.cat([gen_rows(1, p, n) for p, n in pieces[: (ntr + CHUNK - 1) // CHUNK]])[:ntr]
            cent = xtr[torch.randperm(ntr, device=dev)[:L]].clone()
            for _ in range(10):  # Lloyd iterations (faiss/sklearn k-means stand-in; offline build step)
                lab = (xtr @ cent.T).argmax(1)
                cent = torch.zeros_like(cent).index_add_(0, lab, xtr)
                cent = torch.nn.functional.normalize(cent, dim=1)
            lab = (xtr @ cent.T).argmax(1)
            net.train(xtr, lab, epochs=args.epochs)
            del xtr, lab, cent
        if world > 1 and layers is None:
            for p_ in net.model.parameters():
                dist.broadcast(p_.data, src=0)
        self.layers = layers if layers is not None else linear_layers(net.model)

        # ---- placement: argmax MLP(x) over all N (LearnedIndexBuilder.py:76) with the HIP MLP kernels
        eng = _capi.Index(local_rank, chunk_rows=args.chunk_rows, prefilter=not self.exact)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.set_mlp(self.layers)
        eng.set_timing(args.timing_level)
        labels = torch.empty(N, dtype=torch.int32, device=dev)
        for p, n in pieces:
            eng.mlp_topk_device(gen_rows(1, p, n), 1, labels[p * CHUNK: p * CHUNK + n])
        torch.cuda.synchronize()
        labels_h = labels.cpu().numpy().astype(np.int64)
        self.sizes = sizes = np.bincount(labels_h, minlength=L)
        # Expected scan work of a bucket = its rows x the queries it will receive, the latter estimated at build
        # time by routing a sample of the DATA through the MLP (every rank computes the same estimate).
        work_w = estimate_bucket_work(eng, gen_rows(1, 0, pieces[0][1])[: min(20_000, N)], nb, sizes)
        self.work_w = work_w
        owner = assign_buckets(sizes, world, weights=work_w)
        owned = (owner == rank).astype(np.uint8) if world > 1 else None
        self.shard_world, self.owner_all = world, owner
        self.replica = world > 1 and getattr(args, "shard_mode", "bucket") == "replica"
        if self.replica:   # query-sharded replicas: every rank ingests everything
            owner, owned, self.shard_world = np.full(L, rank, dtype=np.int32), None, 1
        if args.emulate_shard and tag == "main":
            er, ew = (int(v) for v in args.emulate_shard.split("/"))
            self.shard_world = ew
            self.owner_all = assign_buckets(sizes, ew, weights=work_w)
            owned = (self.owner_all == er).astype(np.uint8)
            owner = np.where(self.owner_all == er, 0, -1)
        self.owner = owner
        eng.buckets_begin(labels_h, d, L, owned=owned)
        if owned is None:
            for p, n in pieces:
                eng.add_rows(gen_rows(1, p, n), p * CHUNK)
                torch.cuda.synchronize()
        else:
            # owned-only ingest: a rank passes in (a real deployment: reads) only the objects of its own buckets
            own_mask = torch.from_numpy(owned.astype(bool)).to(dev)
            for p, n in pieces:
                keep = torch.nonzero(own_mask[labels[p * CHUNK: p * CHUNK + n].long()]).flatten()
                if keep.numel():
                    eng.add_owned_rows(gen_rows(1, p, n)[keep].contiguous(), (keep + p * CHUNK).contiguous())
                torch.cuda.synchronize()
        eng.buckets_end()
        del labels
        torch.cuda.empty_cache()
        self.eng = eng
        if rank == 0:
            log(f"[bench:{label or tag}] index built in {time.time() - t_setup:.1f}s: N={N} d={d} L={L} bucket sizes "
                f"min/median/max = {sizes.min()}/{int(np.median(sizes))}/{sizes.max()}, empty={int((sizes == 0).sum())}")

    # ------------------------------------------------------------------------------------------------
    def run(self, steps, warmup, shard_inference=True, measure_resident=True):
        """Times `steps` host-in -> host-out searches (pipelined) and, optionally (main() picks `value` from these), as many device-resident ones.
        Returns a dict of raw measurements."""
        import torch
        import torch.distributed as dist

        from learnedmetricindex_amd.pipeline import HostPipeline
        from learnedmetricindex_amd.sharded import ShardedSearcher

        eng, world, rank, dev = self.eng, self.world, self.rank, self.dev
        nb, nq, d, k = self.cfg["nb"], self.cfg["nq"], self.cfg["d"], self.args.k
        lib_comm = None
        if world > 1 and os.environ.get("LMI_BENCH_LIBCOMM"):  # result exchange by lmi_allgather_merge (RCCL inside the library)
            if getattr(self, "_lib_comm", None) is None:
                wd = Watchdog(rank)
                wd.arm("lmi_comm_init (RCCL inside the library)")
                ids = [self.eng.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                self._lib_comm = eng.comm_init(rank, world, ids[0])
                wd.done()
            lib_comm = self._lib_comm
        if self.replica:
            from learnedmetricindex_amd.sharded import ReplicaSearcher
            searcher = ReplicaSearcher(eng, rank, world)
        else:
            searcher = ShardedSearcher(eng, rank, world, shard_inference=shard_inference, lib_comm=lib_comm)
            searcher.time_collectives = world > 1   # two events per collective: the time a rank waits in each exchange (per_rank)

        def sync_all():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()

        # the batches as they arrive: host memory (pinned, DMA-able).  The timed loop ROTATES several distinct batches (fresh draws
        # of the same distribution) so that no step finds its queries warm in the Infinity Cache from the step before; batch 0 is
        # the one the checks below (oracle, recall) look at and the last one submitted.
        nrot = max(1, self.args.rotate_batches)
        rot = [self.queries] + [self.gen_rows(7, 100 + i, nq) for i in range(nrot - 1)]
        q_hosts = [t.cpu().pin_memory() for t in rot]
        q_host = q_hosts[0]
        ov = os.environ.get("LMI_PIPE_OVERLAP", "1")
        pipe = HostPipeline(eng, nq, d, d, nb, k, depth=int(os.environ.get("LMI_PIPE_DEPTH", "2")), same_queries=True,
                            # (the bucket order is the bench's own need -- roofline, oracle check -- not part of the reference's search() result: on
                            # one GPU it is fetched from the slot's device buffer AFTER the timed region, not downloaded with every batch)
                            want_bucket_order=world > 1,
                            overlap_inference=(ov == "1") if world == 1 else os.environ.get("LMI_PIPE_OVERLAP_SHARDED", "0") == "1",
                            two_handles=ov == "2", sharded=searcher if world > 1 else None,
                            use_graph=world == 1 and os.environ.get("LMI_PIPE_GRAPH", "0") == "1",
                            direct_out=world == 1 and os.environ.get("LMI_PIPE_DIRECT", "1") == "1")   # (dists, ids) stored straight into the pinned host buffers
        for qh in q_hosts:   # LMI_PIPE_GRAPH=1: hipGraph replay (opt-in: measured SLOWER than eager submission on ROCm 7.2 -- C1 0.27 against 0.22 ms
            pipe.capture(qh)  # per batch: two graph launches cost more than the dozen eager ones); the graphs are built here, untimed
        # (world > 1: the rank's MLP slice of batch i+1 beside the scan of batch i is opt-in -- it could only be rehearsed with
        # gloo on one card, where it was slower; on a single GPU the same overlap is measured: -1.9 %)
        for i in range(warmup):
            pipe.submit(q_hosts[(i + 1) % nrot])
        pipe.drain()
        sync_all()
        eng.timings_reset()
        stamps = []
        # no garbage collection inside the timed region: a collection that finalises an earlier leg's device tensors ends in hipFree, which
        # synchronises the device (seen as one 30-ms step in a tenth of the multi-leg runs' hard legs: 6.1 -> 8.7-10.3 ms per step over 10 steps)
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        t0 = time.perf_counter()
        for i in range(steps):
            ticket = pipe.submit(q_hosts[(steps - 1 - i) % nrot])   # ... the last step submits batch 0
            stamps.append(time.perf_counter() - t0)
        pipe.drain()
        sync_all()
        elapsed = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        if os.environ.get("LMI_BENCH_DEBUG") and rank == 0:
            log(f"[bench:{self.tag}] submit returned at (ms): {[round(v * 1e3, 2) for v in stamps]}; end {elapsed * 1e3:.2f}")
        # device clock stamps (or, at --timing-level 3, hipEvents) on the kernels' own stream around every phase of every step; read
        # once, after the timed region (the handle keeps the newest 128 sets), so the loop itself has no host synchronisation
        phases, n_timed = eng.timings_mean()
        calls = searcher.calls_per_search * pipe.calls_per_batch
        phases = phases * calls
        out_d, out_i = (a.copy() for a in pipe.result(ticket))
        bo = pipe.bucket_order(ticket).copy()
        tm = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        coll = searcher.collective_ms() if getattr(searcher, "time_collectives", False) else None
        res = dict(elapsed=float(tm.item()), elapsed_local=elapsed, phases=phases, n_timed=int(n_timed) // calls, collectives_ms=coll,
                   out_d=out_d, out_i=out_i, bo=bo, calls=calls, overlapped=bool(pipe.overlap or pipe.sh_overlap))
        if measure_resident:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            q = self.queries
            for _ in range(max(1, warmup)):
                searcher.search(q, q, nb, k)
            sync_all()
            eng.timings_reset()
            if getattr(searcher, "time_collectives", False):
                searcher.collective_ms()   # (drops the warm-up's events)
            t0 = time.perf_counter()
            for i in range(steps):
                qi = rot[(steps - 1 - i) % nrot]
                rd, ri, _ = searcher.search(qi, qi, nb, k)
            sync_all()
            # phases of the SEQUENTIAL loop: in the pipelined one the next batch's MLP runs on a stream of its own, its
            # events span the time it waits for CUs
            pr_, nr_ = eng.timings_mean()
            res["phases_resident"] = pr_ * searcher.calls_per_search
            res["n_timed_resident"] = int(nr_) // searcher.calls_per_search
            res["resident_elapsed_local"] = time.perf_counter() - t0
            tr = torch.tensor([res["resident_elapsed_local"]], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            res["resident_elapsed"] = float(tr.item())
            res["collectives_ms_resident"] = searcher.collective_ms() if getattr(searcher, "time_collectives", False) else None
            assert np.array_equal(ri.cpu().numpy().view(np.uint32), out_i), "resident and host-boundary results differ"
            if self.args.timing_level == 2:
                # cross-check of the device stamps: the same phases from hipEvents recorded between the kernels (timing level 3: each
                # event is a ~5 us bubble, so this loop is slower than the timed one and not part of it)
                eng.set_timing(3)
                for _ in range(2):
                    searcher.search(q, q, nb, k)
                sync_all()
                eng.timings_reset()
                for i in range(min(5, steps)):
                    searcher.search(rot[i % nrot], rot[i % nrot], nb, k)
                sync_all()
                res["phases_events"] = eng.timings_mean()[0] * searcher.calls_per_search
                eng.set_timing(2)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        res["scan_stats"] = eng.scan_stats()
        res["pf_stats"] = eng.prefilter_stats()
        try:   # candidates pass 2 emitted for the last batch (all columns)
            res["pf_candidates"] = int(eng.debug_peek("cand_total", 8).view(np.uint64)[0]) if not self.exact else 0
        except Exception:  # noqa: BLE001
            res["pf_candidates"] = None
        try:   # columns whose candidate buffer overflowed in the last batch (they get a second run of pass 2)
            res["pf_redo_columns"] = int(eng.debug_peek("pf_redo", 4).view(np.uint32)[0]) if not self.exact else 0
        except Exception:  # noqa: BLE001
            res["pf_redo_columns"] = None
        if os.environ.get("LMI_P2_ENDS") and not self.exact:   # -DLMI_P2_ENDS builds: the ragged end of pass 2 (100 MHz ticks)
            raw = eng.debug_peek("pf_stamps", 8 * (192 + 256)).view(np.uint64).astype(np.int64)
            t0, ends = int(raw[191]), np.sort(raw[192:][raw[192:] > 0])
            if len(ends):
                rel = (ends - t0) / 100.0
                sys.stderr.write("[p2 ends %s] blocks %d: first %.1f us, median %.1f, last %.1f; mean idle before the last %.1f us (%.2f %% of the launch)\n"
                                 % (self.tag, len(ends), rel[0], float(np.median(rel)), rel[-1], float((rel[-1] - rel).mean()),
                                    100.0 * float((rel[-1] - rel).mean()) / rel[-1]))
        return res

    def recall(self, out_i, nr):
        import torch

        from learnedmetricindex_amd.sharded import ShardedSearcher

        L, k = self.cfg["leaves"], self.args.k
        q = self.queries[:nr].contiguous()
        gt_searcher = ShardedSearcher(self.eng, 0, 1) if self.replica else ShardedSearcher(self.eng, self.rank, self.world)
        _, gt_i, _ = gt_searcher.search(q, q, L, k)
        gt = gt_i.cpu().numpy().astype(np.int64)
        got = out_i[:nr].astype(np.int64)
        return float(np.mean([len(set(a) & set(b)) / float(k) for a, b in zip(got, gt)]))


def oracle_check_sample(wl, res, args, ns, nthr):
    """The bit-exact checker (oracle/lmi_oracle.c) on the first `ns` queries of a workload whose index is NOT copied to the
    host as a whole: the visited buckets are read back one at a time.  Asserts identical ids and distances; returns the
    checker's own rate.  Used for the hard leg (the main leg checks inside cpu_baselines, on its host copy of the index)."""
    from oracle import lmi_oracle

    eng, layers = wl.eng, wl.layers
    nb, k = wl.cfg["nb"], args.k
    out_d, out_i, bo = res["out_d"], res["out_i"], res["bo"]
    qh = wl.queries[:ns].cpu().numpy()
    t_cpu = time.perf_counter()
    order_o = lmi_oracle.precompute_bucket_order(layers, qh, nb, nthreads=nthr)
    t_cpu = time.perf_counter() - t_cpu
    assert np.array_equal(order_o[:, :, 0], bo[:ns]), "oracle bucket order differs from the GPU's"
    sizes = eng.bucket_sizes()
    rank_d = np.full((nb, ns, 10), np.inf)
    rank_i = np.zeros((nb, ns, 10), dtype=np.uint32)
    for b in np.unique(order_o[:, :, 0]):
        if sizes[b] == 0:
            continue
        rows, ids = eng.read_bucket(int(b))
        t1 = time.perf_counter()
        for r in range(nb):
            rel = np.flatnonzero(order_o[:, r, 0] == b)
            if rel.size:
                sim, idx = lmi_oracle.knn_ip(qh[rel], rows, 10, nthreads=nthr)
                rank_d[r, rel] = np.float32(1) - sim
                rank_i[r, rel] = ids[idx]
        t_cpu += time.perf_counter() - t1
        del rows, ids
    fd = fi = None
    for r in range(nb):
        fd, fi = lmi_oracle.merge_rank(fd, fi, rank_d[r], rank_i[r], k)
    assert np.array_equal(fi, out_i[:ns]) and np.array_equal(fd, out_d[:ns].astype(np.float64)), \
        "CPU oracle and GPU results differ on the sampled queries"
    return round(ns / t_cpu, 3)


def cpu_baselines(wl, res, args, reference_full=True, reference_sample=True, oracle_queries=None, ref_deadline_s=75.0):
    """rank 0, N = 1: the CPU restatements timed on this box's host cores, on the same index and queries as the GPU leg
    `wl` / `res`.  reference_full: the reference-structured pandas loop run IN FULL over the whole index;
    reference_sample: the same loop on a few whole buckets, scaled by rows (1 core; the cross-check of round 1-3)."""
    import torch

    from oracle import cpu_baseline as cb
    from oracle import lmi_oracle

    eng, layers = wl.eng, wl.layers
    N, d, L, nb, nq, k = wl.cfg["n"], wl.cfg["d"], wl.cfg["leaves"], wl.cfg["nb"], wl.cfg["nq"], args.k
    host_cores = os.cpu_count() or 1
    usable = cb.usable_cpus()
    nthr = args.cpu_threads or min(16, usable)  # the GPU box's CPU share for one GPU is 16 cores (os.cpu_count() shows the host's 256)
    out_d, out_i, bo = res["out_d"], res["out_i"], res["bo"]
    variants = {}

    # ---- (0) the bit-exact checker: oracle/lmi_oracle.c on a query sample (canonical fmaf chains; slow by design)
    ns = min(args.cpu_queries if oracle_queries is None else oracle_queries, nq)
    qh_all = wl.queries.cpu().numpy()
    qh = qh_all[:ns]
    t_cpu = time.perf_counter()
    order_o = lmi_oracle.precompute_bucket_order(layers, qh, nb, nthreads=nthr)
    t_cpu = time.perf_counter() - t_cpu
    assert np.array_equal(order_o[:, :, 0], bo[:ns]), "oracle bucket order differs from the GPU's"
    # host copy of the index in bucket order (not timed: the reference's frames are resident before search too)
    sizes = eng.bucket_sizes()
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    slab = torch.empty((int(offsets[-1]), d), dtype=torch.float32)
    slab_np = slab.numpy()
    ids_all = np.empty(int(offsets[-1]), dtype=np.uint32)
    t_copy = time.perf_counter()
    for b in range(L):
        if sizes[b]:
            eng.read_bucket(b, rows_out=slab_np[offsets[b]: offsets[b + 1]], ids_out=ids_all[offsets[b]: offsets[b + 1]])
    t_copy = time.perf_counter() - t_copy
    rank_d = np.full((nb, ns, 10), np.inf)
    rank_i = np.zeros((nb, ns, 10), dtype=np.uint32)
    for b in np.unique(order_o[:, :, 0]):
        if sizes[b] == 0:
            continue
        rows, ids = slab_np[offsets[b]: offsets[b + 1]], ids_all[offsets[b]: offsets[b + 1]]
        t1 = time.perf_counter()
        for r in range(nb):
            rel = np.flatnonzero(order_o[:, r, 0] == b)
            if rel.size:
                sim, idx = lmi_oracle.knn_ip(qh[rel], rows, 10, nthreads=nthr)
                rank_d[r, rel] = np.float32(1) - sim
                rank_i[r, rel] = ids[idx]
        t_cpu += time.perf_counter() - t1
    t1 = time.perf_counter()
    fd = fi = None
    for r in range(nb):
        fd, fi = lmi_oracle.merge_rank(fd, fi, rank_d[r], rank_i[r], k)
    t_cpu += time.perf_counter() - t1
    assert np.array_equal(fi, out_i[:ns]) and np.array_equal(fd, out_d[:ns].astype(np.float64)), \
        "CPU oracle and GPU results differ on the sampled queries"
    variants["oracle_checker"] = {"value": round(ns / t_cpu, 3), "unit": "queries/s", "cores": nthr, "queries": ns,
                                  "identical_ids_and_distances": True,
                                  "what": "oracle/lmi_oracle.c: canonical k-ordered fmaf chains (bit-exact with the GPU, "
                                          "asserted); a checker, not a tuned CPU implementation"}

    # ---- (1) best-effort CPU: bucket-contiguous slab + torch-CPU matmul/topk, all ranks of a bucket in one product
    nq_b = min(args.cpu_best_queries, nq)
    qt = torch.from_numpy(qh_all[:nq_b])
    bd, bi, border, secs = cb.best_effort(slab, offsets, ids_all, layers, qt, nb, k, threads=nthr)
    variants["best_effort_torch"] = {
        "value": round(nq_b / secs, 2), "unit": "queries/s", "cores": nthr, "queries": nq_b, "seconds": round(secs, 3),
        "id_set_agreement_with_gpu": round(cb.id_agreement(bi, out_i[:nq_b]), 6),
        "what": "bucket-contiguous slab in host memory, torch-CPU addmm/relu/topk routing, per visited bucket ONE "
                "matmul for all its (query, rank) slots + topk(10), stable sort merge; full index, no pandas"}
    # ---- (1b) the same arithmetic parallel over BUCKETS: one single-threaded product per worker thread
    bd2, bi2, _, secs2 = cb.best_effort_bucket_parallel(slab, offsets, ids_all, layers, qt, nb, k, workers=nthr)
    assert cb.id_agreement(bi2, bi) > 0.9999, "the two best-effort CPU variants disagree"
    variants["best_effort_bucket_parallel"] = {
        "value": round(nq_b / secs2, 2), "unit": "queries/s", "cores": nthr, "queries": nq_b, "seconds": round(secs2, 3),
        "what": f"the best-effort variant with {nthr} worker threads each taking whole buckets (single-threaded matmul + topk per "
                f"bucket) instead of {nthr} BLAS threads per product; same results (ids equal outside float32 near-ties)"}
    del bd, bd2, bi2

    import pandas as pd
    from threadpoolctl import threadpool_limits

    # ---- (2) reference-structured, measured IN FULL: pandas groupby + .loc gather + BLAS + partial sort, per rank x bucket,
    # over the whole index (LearnedIndex.py:101-157, 328-373).  Its cost is data movement (75 % pandas) and
    # hardly depends on the number of queries, so a query SUBSAMPLE keeps it inside the bench's budget; a deadline stops it after
    # the first rank that ends past it (every rank moves the same data: scaled by nb / ranks done and flagged).
    nq_r = min(args.cpu_ref_queries, nq)
    if reference_full:
        dp_all = np.repeat(np.arange(L, dtype=np.int64), sizes)
        labels_all = ids_all.astype(np.int64)
        nav = pd.DataFrame(slab_np, index=labels_all, copy=False)
        srch = pd.DataFrame(slab_np, index=labels_all, copy=False)   # a distinct frame object
        torch.set_num_threads(nthr)
        with threadpool_limits(limits=nthr):
            t1 = time.perf_counter()
            order_r = cb.mlp_order_numpy(layers, qh_all[:nq_r], nb)
            t_mlp = time.perf_counter() - t1
            rd, rn, parts = cb.reference_structured(nav, srch, qh_all[:nq_r], order_r, dp_all, k, deadline_s=ref_deadline_s)
            t_meas = time.perf_counter() - t1
        done = int(parts.pop("ranks_done"))
        complete = done == nb
        t_full = t_meas if complete else t_mlp + (t_meas - t_mlp) * nb / done
        variants["reference_structured_measured"] = {
            "value": round(nq_r / t_full, 3), "unit": "queries/s", "cores": nthr, "queries": nq_r, "estimated": not complete,
            "measured_seconds": round(t_meas, 3), "ranks_run": done, "ranks": nb, "rows": int(offsets[-1]),
            "seconds_by_part": {kk: round(v, 3) for kk, v in parts.items()},
            # its data movement does not depend on the batch size, its k-NN part scales with it: the whole batch's rate from these parts
            "whole_batch_estimate_queries_per_s": round(nq / (t_full + parts.get("knn", 0.0) * (nq / nq_r - 1.0) * (1.0 if complete else nb / done)), 3),
            "id_set_agreement_with_gpu": round(cb.id_agreement(rn, out_i[:nq_r]), 6) if complete else None,
            "what": "the reference's loop (LearnedIndex.py:101-157, 328-373) over the WHOLE index: per rank a groupby "
                    "materialising every group, a label-based .loc gather copy of each visited bucket, BLAS sgemm + partial sort "
                    "(faiss.knn stand-in), 1 - sim, stable merge" + ("" if complete else f"; stopped by the {ref_deadline_s:.0f}-s "
                    f"deadline after {done} of {nb} ranks, scaled")}
        del nav, srch, dp_all, labels_all

    # ---- (2b) the cross-check of rounds 1-3: the same loop on a few whole buckets, one core, scaled by rows
    if reference_sample:
        order_s = bo[:nq]
        visited = np.bincount(order_s.ravel(), minlength=L)
        cand = np.flatnonzero((sizes > 0) & (visited > 0))
        pick = np.sort(np.random.RandomState(args.seed).choice(cand, size=min(args.cpu_ref_buckets, cand.size), replace=False))
        rows_s = np.concatenate([slab_np[offsets[b]: offsets[b + 1]] for b in pick])
        labels_s = np.concatenate([ids_all[offsets[b]: offsets[b + 1]] for b in pick]).astype(np.int64)
        dp_s = np.concatenate([np.full(int(sizes[b]), b, dtype=np.int64) for b in pick])
        scale = float(offsets[-1]) / float(rows_s.shape[0])
        nav = pd.DataFrame(rows_s, index=labels_s, copy=False)
        srch = pd.DataFrame(rows_s, index=labels_s, copy=False)
        torch.set_num_threads(1)
        with threadpool_limits(limits=1):
            t1 = time.perf_counter()
            cb.mlp_order_numpy(layers, qh_all[:nq], nb)
            t_mlp = time.perf_counter() - t1
            _, _, parts = cb.reference_structured(nav, srch, qh_all[:nq], order_s, dp_s, k)
            t_meas = time.perf_counter() - t1
        torch.set_num_threads(nthr)
        parts.pop("ranks_done", None)
        variants["reference_structured_1core_sampled"] = {
            "value": round(nq / (t_mlp + (t_meas - t_mlp) * scale), 3), "unit": "queries/s", "cores": 1, "queries": nq, "estimated": True,
            "measured_seconds": round(t_meas, 3), "sampled_buckets": int(pick.size), "sampled_rows": int(rows_s.shape[0]),
            "scale_rows_total_over_sampled": round(scale, 3), "seconds_by_part": {kk: round(v, 3) for kk, v in parts.items()},
            "what": "the same loop on whole sampled buckets with all the batch's queries, ONE core, scaled by rows to the full index "
                    "(the reference publishes ~45 q/s on one core of a Xeon 6130, README.md:54-68)"}
        del nav, srch, rows_s
    best = max((variants["best_effort_torch"], variants["best_effort_bucket_parallel"]), key=lambda v: v["value"])
    best_name = "best_effort_torch" if best is variants["best_effort_torch"] else "best_effort_bucket_parallel"
    return {"value": best["value"], "unit": "queries/s", "cores": best["cores"], "kind": "port",
            "sample": f"value = `{best_name}` on {best['cores']} threads (this job's CPU share: {usable} usable of the host's {host_cores} cores): "
                      f"all {N} x {d} rows resident in host memory, first {nq_b} of {nq} queries, all {nb} ranks, torch-CPU matmul + topk over "
                      f"the bucket-contiguous slab; `variants` also holds the reference-structured pandas/BLAS loop measured over the whole "
                      f"index on a {nq_r}-query subsample, its one-core figure from a bucket sample, and the bit-exact oracle used as the checker",
            "host_cpu_count": host_cores, "usable_cpus": usable, "index_copy_to_host_s": round(t_copy, 2), "variants": variants}


def pmc_replay(roof, path, how):
    """`traffic` (HBM bytes per launch) and the MFMA pipe's busy fraction cannot be measured inside the run (counters need rocprofv3
    passes of their own): they are REPLAYED from a committed summary of separate `rocprofv3 --pmc` passes, and only when that summary
    was collected from the very library this process loaded -- the source hash the profiled run reported must be known (not None,
    not 'unknown': a library built outside build.sh) and equal to the loaded library's, and the profiled kernel must be this leg's."""
    roof.setdefault("traffic", None)
    roof.setdefault("mfma_pipe_busy_frac", None)
    if not os.path.exists(path):
        return
    pj = json.load(open(path))
    have = (pj.get("lib") or {}).get("built_from_source_sha16")
    mine = lib_provenance()["built_from_source_sha16"]
    kern = str(pj.get("kernel", "")).rstrip("(").split("<")[0]
    same = bool(have) and have != "unknown" and have == mine and bool(kern) and kern in roof["kernel"]
    if same:
        roof["traffic"], roof["mfma_pipe_busy_frac"] = pj.get("hbm_bytes_per_launch"), pj.get("mfma_pipe_busy_frac")
    roof["traffic_source"] = {"file": os.path.relpath(path, ROOT), "collected_utc": pj.get("collected_utc"), "commit": pj.get("commit"),
                              "matches_loaded_library": same, "how": "replayed from separate rocprofv3 --pmc passes of " + how}


def dominant_roofline(args, cfg, res, sizes, owner, rank, capi, exact=None):
    """Roofline of the dominant kernel of one timed leg.  Algorithmic work per launch: flops = 2*d*sum over (query, rank)
    slots of the bucket size (sharded runs: this rank's slots); bytes = every visited bucket read once in the kernel's
    operand type + the packed queries once.  The larger of the two floors names the bound."""
    d, L, nb, nq = cfg["d"], cfg["leaves"], cfg["nb"], cfg["nq"]
    exact = args.exact if exact is None else exact
    phases, bo = res["phases"], res["bo"]
    flops = res["scan_stats"][0]
    dom_slot = capi.T_SCAN if exact else capi.T_PF_EMIT
    dom_s = max(float(phases[dom_slot]) * 1e-3, 1e-12)  # 0 when --timing-level < 2: the roofline fields are meaningless then
    visited = np.unique(bo)
    visited = visited[(visited >= 0) & (owner[np.clip(visited, 0, L - 1)] == rank)]
    rows_visited = float(sizes[visited].sum())
    if exact:
        kernel, op_bytes, peak_tf = "lmi::scan_kernel", 4, PEAK_F32_MFMA_TFLOPS
    else:
        kernel, op_bytes, peak_tf = ("lmi::pass2_kernel<false>" if d > 128 else "lmi::pass2_small_kernel<KG, false>"), 2, PEAK_F16_MFMA_TFLOPS
    # the fp16 slab pads K to whole k16-groups (d <= 128, lmi_pass2_small.h) or to pairs of them (a stage of pass2_kernel holds two)
    dpad = d if exact else (-(-d // 16) * 16 if d <= 128 else -(-d // 32) * 32)
    alg_bytes = op_bytes * d * (rows_visited + nq * nb)
    t_mfma, t_hbm = flops / (peak_tf * 1e12), alg_bytes / (PEAK_HBM_GBS * 1e9)
    if t_mfma >= t_hbm:
        roof = {"bound": "mfma", "achieved": round(flops / dom_s / 1e12, 3), "peak": peak_tf, "unit": "TFLOP/s",
                "frac": round(flops / dom_s / 1e12 / peak_tf, 4)}
    else:
        roof = {"bound": "hbm", "achieved": round(alg_bytes / dom_s / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(alg_bytes / dom_s / 1e9 / PEAK_HBM_GBS, 4)}
    clk = float(phases[capi.T_CLOCK_MHZ]) if len(phases) > capi.T_CLOCK_MHZ else 0.0
    roof["shader_clock_mhz_under_kernel"] = round(clk, 1) if clk > 0 else None   # s_memtime / s_memrealtime over the life of the kernel's block 0
    roof["timed_with"] = ("device clock stamps written by the kernel's own workgroups in the timed region (lmi_set_timing 2: no hipEvent bubbles); "
                          "avg_launch_ms_hipevents: the same launch between two hipEvents on its stream in a separate 5-step loop"
                          if args.timing_level == 2 else "hipEvents on the kernel's stream in the timed region")
    ev = res.get("phases_events")
    roof["avg_launch_ms_hipevents"] = None if ev is None else round(float(ev[dom_slot]), 4)
    roof.update({"kernel": kernel, "flops_per_launch": flops, "bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom_s * 1e3, 4),
                 "launches_timed": res["n_timed"], "floors_ms": {"mfma": round(t_mfma * 1e3, 3), "hbm": round(t_hbm * 1e3, 3)},
                 "k_padded_to": dpad})
    return roof, flops, dom_s


def notebook_leg(args, dev):
    """The reference's OTHER published configuration (/root/reference/01-Introduction.ipynb cells 15-24, the shape
    of search/search.py's default CLI, :308-327): 100 000 x 768 scan vectors, 32-d navigation vectors, a 2-level index [10, 10] of
    `MLP` (128 hidden) models, 10 buckets per query, 10 000 queries, k = 10 -> 7.75 s = ~1 290 queries/s published.  Here: synthetic
    unit-norm mixture vectors, navigation vectors = a fixed random projection to 32-d (the notebook's pca32 stand-in), the index built by
    li.LearnedIndexBuilder, searched through li.LearnedIndex (lmi_nav_order: the batched priority-queue walk, LearnedIndex.py:216-325,
    then lmi_scan_topk), host arrays in -> host arrays out per call.  Checked on a query sample by the oracle's restatement of that walk
    (precompute_bucket_order_multilevel) + per-bucket knn + stable merge; the reference-structured CPU loop runs IN FULL beside it."""
    import pandas as pd
    import torch
    from threadpoolctl import threadpool_limits

    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.li.BuildConfiguration import BuildConfiguration
    from learnedmetricindex_amd.li.clustering import algorithms
    from learnedmetricindex_amd.li.LearnedIndexBuilder import LearnedIndexBuilder
    from learnedmetricindex_amd.li.model import linear_layers
    from oracle import cpu_baseline as cb
    from oracle import lmi_oracle

    N, d, d_nav, ncat, nb, nq, k = 100_000, 768, 32, [10, 10], 10, 10_000, 10
    t_setup = time.time()
    g = torch.Generator(device=dev).manual_seed(args.seed * 31 + 5)
    centres = torch.randn(100, d, generator=g, device=dev)
    draw = lambda n: torch.nn.functional.normalize(centres[torch.randint(0, 100, (n,), generator=g, device=dev)]   # noqa: E731
                                                   + torch.randn(n, d, generator=g, device=dev), dim=1)
    X, Q = draw(N).cpu().numpy(), draw(nq).cpu().numpy()
    proj = (np.random.RandomState(args.seed).randn(d, d_nav) / np.sqrt(d_nav)).astype(np.float32)
    Xn, Qn = (X @ proj).astype(np.float32), (Q @ proj).astype(np.float32)
    nav = pd.DataFrame(Xn)
    nav.index += 1                                     # 1-based labels (search.py:190-191)
    srch = pd.DataFrame(X)
    srch.index += 1
    torch.manual_seed(args.seed)
    cfg = BuildConfiguration([algorithms["scikit_kmeans"]], [min(args.epochs, 40)], ["MLP"], [0.01], ncat)
    li, dp, n_buckets, build_s, _ = LearnedIndexBuilder(nav, cfg).build()
    d0, n0, mt0 = li.search(nav, Qn, srch, Q, dp, ncat, nb, k)          # uploads the resident index
    steps = max(5, args.steps // 2)
    for _ in range(2):
        li.search_resident(Qn, Q, ncat, nb, k)
    torch.cuda.synchronize()
    acc = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        dg, ng, mt = li.search_resident(Qn, Q, ncat, nb, k)
        for kk, v in mt.items():
            acc[kk] = acc.get(kk, 0.0) + float(v)
    elapsed = time.perf_counter() - t0
    assert np.array_equal(ng, n0) and np.array_equal(dg, d0)
    log(f"[bench:nb100k] built in {time.time() - t_setup:.1f}s ({n_buckets} buckets); {nq * steps / elapsed:.0f} q/s")
    # ---- the oracle on a sample: the walk's bucket order, then ids and distance bits
    ns = min(args.cpu_queries, 256)
    root = linear_layers(li.root_model.model)
    internal = [(pth, linear_layers(m.model)) for pth, m in li.internal_models.items()]
    t1 = time.perf_counter()
    bo = lmi_oracle.precompute_bucket_order_multilevel(root, internal, li.bucket_paths, Qn[:ns], nb, ncat, nthreads=args.cpu_threads or 16)
    do, no, _ = lmi_oracle.search(root, Qn[:ns], X, Q[:ns], dp, nb, k, bucket_order=bo, nthreads=args.cpu_threads or 16)
    t_or = time.perf_counter() - t1
    assert np.array_equal(no, ng[:ns]) and np.array_equal(do, dg[:ns]), "nb100k_2level: CPU oracle and GPU results differ"
    # ---- recall@10 against exact brute force (notebook cell 31)
    gt = np.argsort(-(Q[:1000].astype(np.float64) @ X.astype(np.float64).T), axis=1)[:, :k] + 1
    recall = float(np.mean([len(set(a) & set(b)) / float(k) for a, b in zip(ng[:1000].astype(np.int64), gt)]))
    # ---- the reference-structured pandas loop IN FULL (all 10 000 queries, all 10 ranks, the whole index): a bucket = a distinct path,
    # flattened to one id (the reference groups by [category_L1, category_L2]: the same groupby + .loc gather per bucket and rank);
    # the bucket order is the device walk's (checked above on the sample): navigation is 0.6 % of the reference's own time (cell 24)
    cpu = None
    if not args.no_cpu_baseline:
        nthr = args.cpu_threads or min(16, cb.usable_cpus())
        order_paths, _ = li._precompute_bucket_order(Qn, nb, ncat)
        flat = lambda a: np.where(a[..., 0] < 0, -1, a[..., 0] * 100 + np.maximum(a[..., 1], 0))   # noqa: E731
        torch.set_num_threads(nthr)
        with threadpool_limits(limits=nthr):
            t1 = time.perf_counter()
            rd, rn, parts = cb.reference_structured(nav.copy(), srch, Q, flat(order_paths).astype(np.int32), flat(dp), k)
            t_ref = time.perf_counter() - t1
        parts.pop("ranks_done", None)
        cpu = {"value": round(nq / t_ref, 2), "unit": "queries/s", "cores": nthr, "kind": "port", "queries": nq, "estimated": False,
               "measured_seconds": round(t_ref, 3), "seconds_by_part": {kk: round(v, 3) for kk, v in parts.items()},
               "id_set_agreement_with_gpu": round(cb.id_agreement(rn, ng), 6),
               "sample": f"the reference's loop (LearnedIndex.py:101-157, 328-373) in full: all {nq} queries x {nb} ranks over the whole "
                         f"{N} x {d} frame on {nthr} threads; bucket order from the device walk (navigation not timed)"}
    li.close()
    return {"workload": f"{N}x{d} scan vectors, {d_nav}-d navigation vectors (random projection), 2-level [10,10] MLP(128) index, "
                        f"{nb} buckets, {nq}-query batch, k={k}; li.LearnedIndex.search_resident (lmi_search_tree: the walk + the scan of its buckets in one call), host in -> host out",
            "value": round(nq * steps / elapsed, 2), "unit": "queries/s", "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 4),
            "published_reference_qps": 1290, "published_reference_note": "01-Introduction.ipynb cell 24: 10 000 queries in 7.75 s on LAION-100K (other hardware, real data)",
            "recall_at_10": round(recall, 5), "buckets": int(n_buckets), "build_s": round(build_s, 2),
            "phases_ms": {"inference": round(acc.get("inference", 0.0) / steps * 1e3, 4), "seq_search": round(acc.get("seq_search", 0.0) / steps * 1e3, 4),
                          "search_within_buckets": round(acc.get("search_within_buckets", 0.0) / steps * 1e3, 4), "sort": round(acc.get("sort", 0.0) / steps * 1e3, 4),
                          "search_wall": round(acc.get("search", 0.0) / steps * 1e3, 4)},
            "oracle_check": {"queries": ns, "identical_ids_and_distances": True, "checker_queries_per_s": round(ns / t_or, 3)},
            "cpu_baseline": cpu}


class Watchdog:
    """Multi-rank bring-up must fail FAST and LOUDLY: the first time more than one RCCL rank exists is the driver's SCALE run, and
    a rank stuck in communicator init or in the first collective would otherwise hang until the lease's own limit.  A daemon
    thread ends THIS rank with a non-zero code when `phase` is not done within `seconds` (LMI_BENCH_WATCHDOG_S, default 60);
    torchrun then ends the other ranks and `python bench.py --gpus N` returns that code.  (Exiting is fine for a process that has
    touched the GPU; it must never be replaced by exec.)"""

    def __init__(self, rank: int):
        self.rank = rank
        self.seconds = float(os.environ.get("LMI_BENCH_WATCHDOG_S", "60"))
        self._timer = None

    def arm(self, phase: str):
        import threading

        self.done()

        def fire():
            log(f"[bench] WATCHDOG rank {self.rank}: '{phase}' did not finish within {self.seconds:.0f} s -- exiting 86 "
                f"(check RCCL / xGMI bring-up: NCCL_DEBUG=INFO, HSA_ENABLE_IPC_MODE_LEGACY=0, one process per GPU)")
            sys.stderr.flush()
            os._exit(86)

        self._timer = threading.Timer(self.seconds, fire)
        self._timer.daemon = True
        self._timer.start()

    def done(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None


def bring_up(rank: int, world: int, backend: str, dev, wd: "Watchdog", lib_engine=None):
    """init_process_group + the FIRST collective under the watchdog, then who is there: every rank's device / bus id and the
    RCCL version, logged by rank 0 before any index is built.  Returns {"rccl_ranks_seen", "rccl_version", "devices",
    "lib_comm"} (lib_comm: the library's own ncclComm_t when LMI_BENCH_LIBCOMM=1 and an engine is given)."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    wd.arm(f"init_process_group({backend})")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    wd.arm("first all-reduce")
    t = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t)
    if backend == "nccl":
        torch.cuda.synchronize()
    seen = int(t.item())
    wd.arm("all_gather_object of the ranks' devices")
    mine = {"rank": rank, "pid": os.getpid(), "device": None, "pci_bus_id": None}
    if dev is not None and getattr(dev, "type", "cpu") == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        mine.update(device=f"{pr.name} #{dev.index}", pci_bus_id=getattr(pr, "pci_bus_id", None), hbm_gib=round(pr.total_memory / 2 ** 30, 1))
    devices = [None] * world
    dist.all_gather_object(devices, mine)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception:  # noqa: BLE001
        ver = None
    lib_comm = None
    if lib_engine is not None and os.environ.get("LMI_BENCH_LIBCOMM"):   # result exchange by lmi_allgather_merge (RCCL inside the library)
        wd.arm("lmi_comm_init (RCCL inside the library)")
        ids = [lib_engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        lib_comm = lib_engine.comm_init(rank, world, ids[0])
    wd.done()
    if rank == 0:
        log(f"[bench] {world} ranks up over {backend}" + (f" (RCCL {ver})" if ver else "") + f": first all-reduce saw {seen} ranks")
        for d_ in devices:
            log(f"[bench]   rank {d_['rank']}: pid {d_['pid']}, {d_['device']}, bus {d_['pci_bus_id']}")
    assert seen == world, f"the first all-reduce saw {seen} of {world} ranks"
    return {"rccl_ranks_seen": seen, "rccl_version": ver, "devices": devices, "lib_comm": lib_comm}


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N ranks as a CHILD torchrun (one process per GPU) before this
    process has touched the GPU (no torch import, no HIP call so far -- a process that has initialised the GPU must
    never be replaced or forked from), forward the children's stdout (rank 0's JSON line) and return their exit code."""
    import socket
    import subprocess

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           "--", os.path.abspath(__file__)] + list(argv)   # "--": the launcher's argparse would otherwise claim a bench flag
                                                            # that is a prefix of one of its own (--n -> --nnodes, ...)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    log(f"[bench] --gpus {n} without WORLD_SIZE: launching {' '.join(cmd[1:9])} ...")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    try:
        for line in child.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
        return child.wait()
    except BaseException:
        child.kill()
        child.wait()
        raise


def launch_check() -> None:
    """LMI_BENCH_LAUNCH_CHECK=1: the bring-up (rendezvous, first all-reduce, device roll call, watchdog) over gloo and a JSON line
    from rank 0, nothing else -- lets the CPU test suite (no GPU) drive `python bench.py --gpus N` through launch_ranks end to end.
    LMI_BENCH_LAUNCH_CHECK_FAIL_RANK / _HANG_RANK make one rank exit / never reach the first collective; LMI_BENCH_LIBCOMM=1 runs
    the library-communicator branch of bring_up on a stand-in engine (the real one needs the GPU)."""
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    fail = os.environ.get("LMI_BENCH_LAUNCH_CHECK_FAIL_RANK")
    hang = os.environ.get("LMI_BENCH_LAUNCH_CHECK_HANG_RANK")
    wd = Watchdog(rank)
    if hang is not None and int(hang) == rank:
        wd.arm("a rank that never reaches the first collective (test)")
        time.sleep(3600)
    if fail is not None and int(fail) == rank:
        sys.exit(7)

    class _StubEngine:   # the two calls bring_up makes on the library engine
        inits = 0

        @staticmethod
        def comm_unique_id():
            return bytes(range(128))

        def comm_init(self, r, w, uid):
            assert uid == bytes(range(128)) and 0 <= r < w
            _StubEngine.inits += 1
            return ("stub-comm", r, w)

    eng = _StubEngine() if os.environ.get("LMI_BENCH_LIBCOMM") else None
    up = bring_up(rank, world, "gloo", None, wd, lib_engine=eng)
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "sum": world * (world + 1) // 2, "rccl_ranks_seen": up["rccl_ranks_seen"],
                          "lib_comm": None if up["lib_comm"] is None else list(up["lib_comm"])}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--n", type=int)
    ap.add_argument("--nq", type=int)
    ap.add_argument("--nb", type=int)
    ap.add_argument("--d", type=int, help="diagnostic: another dimensionality on the chosen config's shape (not a bench line)")
    ap.add_argument("--leaves", type=int)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--seed", type=int, default=2023)
    ap.add_argument("--train-rows", type=int, default=200_000)
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--recall-queries", type=int, default=1000)
    ap.add_argument("--cpu-queries", type=int, default=256, help="queries re-computed by the bit-exact oracle (the checker)")
    ap.add_argument("--cpu-best-queries", type=int, default=10_000, help="queries of the best-effort torch-CPU baseline")
    ap.add_argument("--cpu-ref-queries", type=int, default=1_000, help="query subsample of the reference-structured pandas baseline run over the whole index")
    ap.add_argument("--cpu-ref-buckets", type=int, default=6, help="whole buckets the reference-structured baseline is measured on")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU baseline threads (0: min(16, host cores))")
    ap.add_argument("--rotate-batches", type=int, default=4, help="distinct query batches rotated through the timed loop")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C1 / C5 legs (BASELINE.json configs[0] / configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-hard-leg", action="store_true", help="skip the second, harder workload (overlapping clusters)")
    ap.add_argument("--hard-only", action="store_true", help="profiling aid: the MAIN leg runs on the hard generator (overlapping clusters, zipf sizes); nothing else")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the all-f32 leg (the same workload with lmi_set_prefilter(0))")
    ap.add_argument("--hard-sigma", type=float, default=1.0)
    ap.add_argument("--hard-centre-scale", type=float, default=0.26)
    ap.add_argument("--hard-zipf", type=float, default=20.0)
    ap.add_argument("--chunk-rows", type=int, default=None)
    ap.add_argument("--value-loop", choices=("resident", "host"), default="resident",
                    help="which timed loop `value` / `ms_per_step` / `roofline` / `phases_ms` are taken from: `resident` (default) -- the query "
                         "batches are in HBM when the timed region starts and the results stay there; `host` -- the host-in -> host-out pipeline "
                         "(PCIe inside the timed region).  Both loops run either way; the other one is reported beside it")
    ap.add_argument("--timing-level", type=int, default=2, choices=(0, 1, 2, 3),
                    help="lmi_set_timing: 2 (default) times every phase with device-side clock stamps (no bubbles); 3: with hipEvents "
                         "between the kernels (each a ~5 us bubble); 0/1: nothing / the whole call only (roofline fields then null/0)")
    ap.add_argument("--exact", action="store_true",
                    help="all-f32 scan (lmi_set_prefilter(0)) instead of fp16 prefilter + exact re-rank; same results")
    ap.add_argument("--shard-mode", choices=("bucket", "replica"), default="bucket",
                    help="N > 1: `bucket` (default, the headline): bucket b lives on one rank; `replica`: SURVEY 8e's other mode -- "
                         "the whole index on every rank, a rank answers 1/N of the batch, one all-gather of the rows")
    ap.add_argument("--emulate-shard", default=None, metavar="R/W",
                    help="diagnostic, single GPU: own only the buckets rank R of a W-way sharded run would own "
                         "(no collective); shows the per-rank step time of the N>1 bench on one card")
    ap.add_argument("--traffic-json", default=None,
                    help="PMC-derived HBM bytes per scan launch (default: profiles/scan_pmc_<config>.json, "
                         "written by profiles/summarize.py from separate rocprofv3 --pmc passes of this bench)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for key in ("n", "nq", "nb", "leaves", "d"):
        if getattr(args, key) is not None:
            cfg[key] = getattr(args, key)
    N, d, L, nb, nq, k = cfg["n"], cfg["d"], cfg["leaves"], cfg["nb"], cfg["nq"], args.k

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("LMI_BENCH_LAUNCH_CHECK"):
        assert int(os.environ.get("WORLD_SIZE", "1")) == args.gpus
        return launch_check()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # LMI_BENCH_BACKEND=gloo rehearses the multi-rank path on fewer cards than ranks (ranks share devices,
    # collectives staged on the host); the measured configuration is always one rank per GPU over RCCL
    backend = os.environ.get("LMI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    up = None
    if world > 1:   # (the library communicator, when asked for, is set up by Workload.run: it needs the engine)
        up = bring_up(rank, world, backend, dev, Watchdog(rank))
    if args.emulate_shard:
        assert world == 1, "--emulate-shard is a single-GPU diagnostic"
        args.no_cpu_baseline = args.no_recall = args.no_hard_leg = True

    from learnedmetricindex_amd import _capi

    if args.hard_only:   # (tools/profile_round.sh: the PMC passes of the hard leg's dominant kernel)
        args.no_cpu_baseline = args.no_recall = args.no_hard_leg = args.no_other_configs = args.no_exact_leg = True
        wl = Workload(args, cfg, dev, rank, world, local_rank, sigma=args.hard_sigma, zipf=args.hard_zipf, centre_scale=args.hard_centre_scale, tag="hard")
    else:
        wl = Workload(args, cfg, dev, rank, world, local_rank, tag=os.environ.get("LMI_BENCH_TAG", "main"))   # (developer aid: another data seed)
    res = wl.run(args.steps, args.warmup, shard_inference=True)
    # `value` is the rate with the inputs already in HBM when the timed region starts (the PCIe-inclusive host-in -> host-out rate is
    # reported beside it, never as `value`): the step's time, phases and the dominant kernel's duration all come from that SAME loop
    host_loop = {"elapsed": res["elapsed"], "elapsed_local": res["elapsed_local"], "phases": res["phases"], "n_timed": res["n_timed"],
                 "overlapped": res.get("overlapped")}
    value_loop = "resident" if (args.value_loop == "resident" and "resident_elapsed" in res) else "host"
    if value_loop == "resident":
        res = dict(res, elapsed=res["resident_elapsed"], elapsed_local=res["resident_elapsed_local"], phases=res["phases_resident"],
                   n_timed=res.get("n_timed_resident", args.steps), collectives_ms=res.get("collectives_ms_resident"))
    elapsed, phases, out_d, out_i, bo = res["elapsed"], res["phases"], res["out_d"], res["out_i"], res["bo"]
    flops, pairs, items = res["scan_stats"]
    pf_active, pf_survivors, pf_fallbacks = res["pf_stats"]
    sizes, owner = wl.sizes, wl.owner
    if rank == 0 and wl.shard_world > 1:  # how even the bucket assignment turned out for this batch
        bo_h = bo.ravel()
        bo_h = bo_h[(bo_h >= 0) & (bo_h < L)]
        per_rank = np.bincount(wl.owner_all[bo_h], weights=sizes[bo_h].astype(np.float64), minlength=wl.shard_world)
        log(f"[bench] scan work per rank (pairs, share of the mean): {np.round(per_rank / per_rank.mean(), 3).tolist()}")
        if os.environ.get("LMI_BENCH_DUMP_BUCKETS"):   # developer aid: the assignment's inputs and this batch's real routing
            np.savez(os.environ["LMI_BENCH_DUMP_BUCKETS"], sizes=sizes, owner_all=wl.owner_all, work_w=getattr(wl, "work_w", None),
                     m=np.bincount(bo_h, minlength=L))
    # N > 1: the other collective layout (every rank routes the whole batch: ONE all-gather, as north_star words it)
    alt = None
    if world > 1 and not wl.replica:
        r2 = wl.run(args.steps, args.warmup, shard_inference=False, measure_resident=False)
        assert np.array_equal(r2["out_i"], out_i), "the two sharded modes disagree"
        alt = {"mode": "replicated MLP on every rank, ONE all-gather (per-rank top-k)",
               "value": round(nq * args.steps / r2["elapsed"], 2), "ms_per_step": round(r2["elapsed"] / args.steps * 1e3, 4)}
    recall = None if args.no_recall else wl.recall(out_i, min(args.recall_queries, nq))
    # N > 1: every rank's own phase means and scan share, so that a SCALE line explains itself (rank 0 prints them)
    per_rank = None
    if world > 1:
        names = ("inference", "route_pack", "scan", "merge", "total", "pf_sample", "pf_emit", "rescore", "fallback")
        mine = {"rank": rank, "scan_pairs": int(pairs), "scan_items": int(items),
                "rows_owned": int(sizes[owner == rank].sum()), "buckets_owned": int((owner == rank).sum()),
                "step_ms_this_rank": round(res["elapsed_local"] / args.steps * 1e3, 4),
                "phases_ms": {n_: round(float(phases[i]), 4) for i, n_ in enumerate(names)},
                # what the rank WAITS in each exchange (events around the collective on its stream: the other ranks' lateness included)
                "collectives_exposed_ms": res.get("collectives_ms")}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baselines(wl, res, args)

    def leg_loop(r_, nsteps):
        """A secondary leg's (elapsed, phases, n_timed) from the same loop as the main leg's `value`, plus the other loop's figures."""
        if value_loop == "resident" and "resident_elapsed" in r_:
            return (dict(r_, elapsed=r_["resident_elapsed"], phases=r_["phases_resident"], n_timed=r_.get("n_timed_resident", nsteps)),
                    {"host_to_host_ms_per_step": round(r_["elapsed"] / nsteps * 1e3, 4),
                     "host_to_host_dominant_kernel_ms": round(float(r_["phases"][_capi.T_SCAN if r_.get("exact_leg") else _capi.T_PF_EMIT]), 4)})
        return r_, ({"resident_ms_per_step": round(r_["resident_elapsed"] / nsteps * 1e3, 4)} if "resident_elapsed" in r_ else {})

    # ---- second leg: a HARDER workload (overlapping clusters, heavy-tailed bucket sizes): what the prefilter's
    # candidate logic and the throughput look like when recall@10 at top-4 is ~0.9 as on LAION (README.md:54-57)
    hard = None
    if world == 1 and not args.no_hard_leg and args.config == "c2" and not args.exact:
        wl.eng.close()
        wl.eng = None
        torch.cuda.empty_cache()
        wh = Workload(args, cfg, dev, rank, world, local_rank, sigma=args.hard_sigma, zipf=args.hard_zipf,
                      centre_scale=args.hard_centre_scale, tag="hard")
        rh, rh_other = leg_loop(wh.run(max(5, args.steps // 2), 2, measure_resident=True), max(5, args.steps // 2))
        hs = wh.sizes
        hroof, _, _ = dominant_roofline(args, cfg, rh, wh.sizes, wh.owner, rank, _capi)
        pmc_replay(hroof, os.path.join(ROOT, "profiles", "scan_pmc_hard.json"), "bench.py --hard-only")
        hard_checked = None
        if rank == 0 and not args.no_cpu_baseline:   # the checker on the HARD workload too: small score gaps, the bound's window matters
            nthr_h = args.cpu_threads or min(16, os.cpu_count() or 1)
            hq = min(args.cpu_queries, nq, 256)
            hard_checked = {"queries": hq, "identical_ids_and_distances": True,
                            "checker_queries_per_s": oracle_check_sample(wh, rh, args, hq, nthr_h)}
        hard = {"generator": f"centres x{args.hard_centre_scale}, sigma {args.hard_sigma}, cluster weights ~ 1/(1 + c/{args.hard_zipf})",
                "recall_at_10": round(wh.recall(rh["out_i"], min(args.recall_queries, nq)), 5),
                "value": round(nq * max(5, args.steps // 2) / rh["elapsed"], 2), "unit": "queries/s",
                "ms_per_step": round(rh["elapsed"] / max(5, args.steps // 2) * 1e3, 4), "value_loop": value_loop, **rh_other,
                "bucket_sizes_min_median_max": [int(hs.min()), int(np.median(hs)), int(hs.max())],
                "survivors_per_slot": round(rh["pf_stats"][1] / max(1, nq * nb), 2), "fallback_slots": int(rh["pf_stats"][2]),
                "overflowed_columns": rh.get("pf_redo_columns"),
                "scan_pairs": int(rh["scan_stats"][1]), "oracle_check": hard_checked, "roofline": hroof,
                "phases_ms": {"pf_sample": round(float(rh["phases"][5]), 4), "pf_emit": round(float(rh["phases"][6]), 4),
                              "rescore": round(float(rh["phases"][7]), 4), "fallback": round(float(rh["phases"][8]), 4)}}

    # ---- the SAME workload at the reference's own precision: all-f32 scan (lmi_set_prefilter(0): scan_kernel, f32 MFMA with
    # the canonical chain, no prefilter, no re-rank), same MLP weights, same index data, same batch -- its answers must be
    # byte-identical to the default leg's (LearnedIndex.py:360-368 computes in f32 end to end)
    exact_leg = None
    if world == 1 and args.config == "c2" and not args.exact and not args.no_exact_leg and not args.emulate_shard:
        for w_ in (wl, locals().get("wh")):
            if w_ is not None and getattr(w_, "eng", None) is not None:
                w_.eng.close()
                w_.eng = None
        torch.cuda.empty_cache()
        we = Workload(args, cfg, dev, rank, world, local_rank, tag="main", exact=True, layers=wl.layers, label="exact")
        esteps = max(3, args.steps // 4)
        re_ = we.run(esteps, 1, measure_resident=True)
        re_["exact_leg"] = True
        re_, re_other = leg_loop(re_, esteps)
        eroof, eflops, _ = dominant_roofline(args, cfg, re_, we.sizes, we.owner, rank, _capi, exact=True)
        same = bool(np.array_equal(re_["out_i"], out_i) and np.array_equal(re_["out_d"], out_d))
        assert same, "the all-f32 leg and the default leg returned different results"
        eroof["traffic"] = eroof["mfma_pipe_busy_frac"] = None
        if not any(getattr(args, key) is not None for key in ("n", "nq", "nb", "leaves", "d")):
            pmc_replay(eroof, os.path.join(ROOT, "profiles", "scan_pmc_c2_exact.json"), "bench.py --exact")
        echeck = None
        if rank == 0 and not args.no_cpu_baseline:
            eq_ = min(args.cpu_queries, nq, 64)
            echeck = {"queries": eq_, "identical_ids_and_distances": True,
                      "checker_queries_per_s": oracle_check_sample(we, re_, args, eq_, args.cpu_threads or 16)}
        exact_leg = {"what": "lmi_set_prefilter(0): all-f32 scan_kernel (f32 MFMA, canonical k-ordered chain), same weights / index / batch",
                     "dtype": "f32", "value": round(nq * esteps / re_["elapsed"], 2), "unit": "queries/s", "steps": esteps,
                     "ms_per_step": round(re_["elapsed"] / esteps * 1e3, 4), "value_loop": value_loop, **re_other,
                     "identical_to_default_leg": same, "roofline": eroof, "oracle_check": echeck,
                     "phases_ms": {"inference": round(float(re_["phases_resident"][0]), 4), "route_pack": round(float(re_["phases"][1]), 4),
                                   "scan": round(float(re_["phases"][2]), 4), "merge": round(float(re_["phases"][3]), 4)}}
        we.eng.close()
        we.eng = None

    # ---- the other single-GPU configurations of BASELINE.json, each as a short leg of its own: C1 (configs[0]: 100k x 768,
    # 1 000 queries -- the reference's CPU-runnable case) and C5 (configs[4]: 10M x 45, 256 leaves, cosine)
    others = None
    if world == 1 and args.config == "c2" and not args.no_other_configs and not args.exact and not args.emulate_shard:
        others = {}
        for cname in ("c1", "c5"):
            for w_ in (wl, locals().get("wh")):   # the main (and hard) indexes are no longer needed: free their HBM
                if w_ is not None and getattr(w_, "eng", None) is not None:
                    w_.eng.close()
                    w_.eng = None
            torch.cuda.empty_cache()
            ocfg = dict(CONFIGS[cname])
            wo = Workload(args, ocfg, dev, rank, world, local_rank, tag=cname)
            osteps = max(60, 3 * args.steps)   # (a step is 0.2-0.6 ms: 20 steps are 5-10 ms of wall clock, too few to average the host's jitter)
            # (these steps are shorter than the MLP + launch sequence of a batch: the pipelined host -> host loop -- the next batch's MLP on a stream of
            # its own -- is the faster one although PCIe is inside it; it stays these legs' `value`, the sequential resident loop beside it)
            ro = wo.run(osteps, 8, measure_resident=True)
            oroof, _, _ = dominant_roofline(args, ocfg, ro, wo.sizes, wo.owner, rank, _capi)
            pmc_replay(oroof, os.path.join(ROOT, "profiles", f"scan_pmc_{cname}.json"), "bench.py --config " + cname)
            others[cname] = {"workload": f"{ocfg['n']}x{ocfg['d']}, {ocfg['leaves']} leaves, top-{ocfg['nb']}, {ocfg['nq']}-query batch",
                             "value": round(ocfg["nq"] * osteps / ro["elapsed"], 2), "unit": "queries/s",
                             "ms_per_step": round(ro["elapsed"] / osteps * 1e3, 4), "value_loop": "host",
                             "resident_ms_per_step": round(ro["resident_elapsed"] / osteps * 1e3, 4),
                             "recall_at_10": None if args.no_recall else round(wo.recall(ro["out_i"], min(args.recall_queries, ocfg["nq"])), 5),
                             "roofline": oroof,
                             "phases_ms": {"inference": round(float(ro["phases"][0]), 4), "route_pack": round(float(ro["phases"][1]), 4),
                                           "pf_sample": round(float(ro["phases"][5]), 4), "pf_emit": round(float(ro["phases"][6]), 4),
                                           "rescore": round(float(ro["phases"][7]), 4), "merge": round(float(ro["phases"][3]), 4)}}
            if rank == 0 and not args.no_cpu_baseline:
                # C1 is the reference's own CPU-runnable case: everything in full (all 1 000 queries through the reference-structured
                # loop on 16 and on 1 core); C5: the best-effort variants and the checker (its pandas loop moves 10M rows per rank)
                ocpu = cpu_baselines(wo, ro, args, reference_full=(cname == "c1"), reference_sample=False, oracle_queries=128, ref_deadline_s=20.0)
                if cname == "c1":
                    import pandas as pd
                    from threadpoolctl import threadpool_limits
                    from oracle import cpu_baseline as cb_
                    sizes_o = wo.eng.bucket_sizes()
                    off_o = np.concatenate([[0], np.cumsum(sizes_o)]).astype(np.int64)
                    rows_o = np.empty((int(off_o[-1]), ocfg["d"]), dtype=np.float32)
                    ids_o = np.empty(int(off_o[-1]), dtype=np.uint32)
                    for b_ in range(ocfg["leaves"]):
                        if sizes_o[b_]:
                            wo.eng.read_bucket(b_, rows_out=rows_o[off_o[b_]: off_o[b_ + 1]], ids_out=ids_o[off_o[b_]: off_o[b_ + 1]])
                    qo = wo.queries.cpu().numpy()
                    torch.set_num_threads(1)
                    with threadpool_limits(limits=1):
                        t1 = time.perf_counter()
                        order_1 = cb_.mlp_order_numpy(wo.layers, qo, ocfg["nb"])
                        _, rn1, parts1 = cb_.reference_structured(pd.DataFrame(rows_o, index=ids_o.astype(np.int64), copy=False),
                                                                  pd.DataFrame(rows_o, index=ids_o.astype(np.int64), copy=False),
                                                                  qo, order_1, np.repeat(np.arange(ocfg["leaves"], dtype=np.int64), sizes_o), args.k)
                        t_1 = time.perf_counter() - t1
                    torch.set_num_threads(args.cpu_threads or 16)
                    parts1.pop("ranks_done", None)
                    ocpu["variants"]["reference_structured_measured_1core"] = {
                        "value": round(ocfg["nq"] / t_1, 3), "unit": "queries/s", "cores": 1, "queries": ocfg["nq"], "estimated": False,
                        "measured_seconds": round(t_1, 3), "seconds_by_part": {kk: round(v, 3) for kk, v in parts1.items()},
                        "id_set_agreement_with_gpu": round(cb_.id_agreement(rn1, ro["out_i"]), 6),
                        "what": "the reference's loop over the whole C1 index, all 1 000 queries, ONE core -- the configuration the reference "
                                "publishes ~45 q/s for (README.md:54-68: LAION 10M; here 100k rows)"}
                    del rows_o, ids_o
                others[cname]["cpu_baseline"] = ocpu
                others[cname]["oracle_check"] = {"queries": ocpu["variants"]["oracle_checker"]["queries"], "identical_ids_and_distances": True,
                                                 "checker_queries_per_s": ocpu["variants"]["oracle_checker"]["value"]}
            wo.eng.close()
            wo.eng = None

    nb100k = None
    if world == 1 and args.config == "c2" and not args.no_other_configs and not args.exact and not args.emulate_shard and rank == 0:
        for w_ in (wl, locals().get("wh")):
            if w_ is not None and getattr(w_, "eng", None) is not None:
                w_.eng.close()
                w_.eng = None
        torch.cuda.empty_cache()
        nb100k = notebook_leg(args, dev)
        others["nb100k_2level"] = nb100k

    if rank == 0:
        scan_s = max(float(phases[_capi.T_SCAN]) * 1e-3, 1e-12)
        roof, flops, dom_s = dominant_roofline(args, cfg, res, sizes, owner, rank, _capi)
        # `traffic` (HBM bytes per launch from the PMC counters) and the MFMA pipe's busy fraction cannot be measured inside this
        # run (counters need rocprofv3 passes of their own): they are REPLAYED from the committed summary of separate
        # `rocprofv3 --pmc` passes of this same command (tools/profile_round.sh + profiles/summarize.py), and only when that
        # summary was collected from the library sources this run was built from -- otherwise they are null.
        traffic = mfma_busy = None
        prov = lib_provenance()
        tj = args.traffic_json or os.path.join(ROOT, "profiles", f"scan_pmc_{'hard' if args.hard_only else args.config}{'_exact' if args.exact else ''}.json")
        overridden = any(getattr(args, key) is not None for key in ("n", "nq", "nb", "leaves", "d", "emulate_shard"))
        traffic_source = None
        if world == 1 and not overridden and os.path.exists(tj):
            pj = json.load(open(tj))
            # the summary carries the provenance the PROFILED bench run reported about the library it had loaded (not a hash taken
            # later from some working tree): replay only for the same build
            pl = pj.get("lib") or {}
            same = (pl.get("built_from_source_sha16") is not None and pl.get("built_from_source_sha16") == prov["built_from_source_sha16"]
                    and str(pj.get("kernel", "")) in roof["kernel"])
            traffic_source = {"file": os.path.relpath(tj, ROOT), "collected_utc": pj.get("collected_utc"), "commit": pj.get("commit"),
                              "lib": pl, "matches_loaded_library": bool(same),
                              "how": "replayed from separate rocprofv3 --pmc passes of this command, not measured in this run"}
            if same:
                traffic, mfma_busy = pj.get("hbm_bytes_per_launch"), pj.get("mfma_pipe_busy_frac")
        roof.update({"traffic": traffic, "mfma_pipe_busy_frac": mfma_busy, "traffic_source": traffic_source,
                     # the whole scan phase (all its kernels) priced as SURVEY 8d does: algorithmic f32 flops
                     # against the f32 MFMA peak, whatever precision the prefilter used
                     "scan_phase_f32_equiv": {"achieved": round(flops / scan_s / 1e12, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                                              "unit": "TFLOP/s", "frac": round(flops / scan_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                              "avg_ms": round(scan_s * 1e3, 4)}})
        resident = None
        if "resident_elapsed" in res:
            resident = {"value": round(nq * args.steps / res["resident_elapsed"], 2), "unit": "queries/s",
                        "ms_per_step": round(res["resident_elapsed"] / args.steps * 1e3, 4),
                        "what": "the same step with the query batch and the results already/left in HBM (no PCIe), one lmi_search call per "
                                "batch on one stream: no overlap between neighbouring batches"}
        result = {
            "metric": "queries/sec @ recall@10, 768-d 10M index, 10k query batch",
            "value": round(nq * args.steps / elapsed, 2),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if args.exact else "f16 prefilter (f32 accumulate) + f32 exact re-rank; outputs identical to the all-f32 path",
            "data": "synthetic",
            "value_loop": value_loop,
            "boundary": ("`value` / `ms_per_step` / `roofline` / `phases_ms`: the query batches are in HBM when the timed region starts and the "
                         "results stay in HBM (one lmi_search per batch on one stream, MLP included, nothing overlapped between batches); "
                         "`host_to_host` is the PCIe-inclusive rate of the same steps (pinned host buffers in and out, pipelined)"
                         if value_loop == "resident" else
                         "host-in -> host-out: every step uploads its query batch from pinned host memory and downloads "
                         "(dists, ids) to pinned host memory; transfers of neighbouring batches overlap the search"
                         + ("; the MLP of batch i+1 runs on a stream of its own beside the scan of batch i (fills the tails of "
                            "its kernels): phases_ms.inference is the figure of the sequential (resident) loop" if res.get("overlapped") else "")),
            "recall_at_10": None if recall is None else round(recall, 5),
            "config": {"workload": f"{N}x{d} unit-norm gaussian-mixture vectors, 1-level LMI ({L} leaves, "
                                   f"{cfg['model']} {d}->512->{L} trained {args.epochs} epochs), top-{nb} buckets, "
                                   f"{nq}-query batch, k={k}",
                       "baseline_config": args.config, "parallelism": "single GPU" if world == 1 else
                                      f"query-sharded replicas x{world}: the whole index on every rank, a rank answers 1/{world} of the "
                                      f"batch, one all-gather of [dists | ids | bucket order] rows" if wl.replica else
                                      f"bucket-sharded x{world} (owned-only ingest): MLP on 1/{world} of the batch + all-gather of "
                                      f"the bucket order, scan of the owned buckets + all-gather of the per-rank top-k",
                       "scan_pairs": int(pairs), "scan_items": int(items)},
            "roofline": roof,
            "prefilter": None if args.exact else {"survivors_per_slot": round(pf_survivors / max(1, nq * nb), 2),
                                                   "fallback_slots": int(pf_fallbacks), "overflowed_columns": res.get("pf_redo_columns"),
                                                   "candidates": res.get("pf_candidates")},
            "cpu_baseline": cpu,
            "resident": resident,
            # the PCIe-inclusive rate: every step uploads its batch from pinned host memory and stores (dists, ids) into pinned host memory;
            # uploads / downloads of neighbouring batches and the next batch's MLP (a stream of its own) overlap the search
            "host_to_host": {"value": round(nq * args.steps / host_loop["elapsed"], 2), "unit": "queries/s",
                             "ms_per_step": round(host_loop["elapsed"] / args.steps * 1e3, 4),
                             "dominant_kernel_ms": round(float(host_loop["phases"][_capi.T_SCAN if args.exact else _capi.T_PF_EMIT]), 4),
                             "mlp_of_next_batch_overlapped": bool(host_loop["overlapped"])},
            "sharded_alt_mode": alt,
            "rccl_ranks_seen": None if up is None else up["rccl_ranks_seen"],
            "rccl_version": None if up is None else up["rccl_version"],
            "rank_devices": None if up is None else up["devices"],
            "per_rank": per_rank,
            # N > 1: how uneven the ranks' dominant kernel was (max / mean of pf_emit over the ranks: 1.0 = perfectly balanced shards)
            "per_rank_imbalance": None if not per_rank else {
                "pf_emit_max_ms": round(max(p_["phases_ms"]["pf_emit"] for p_ in per_rank), 4),
                "pf_emit_mean_ms": round(float(np.mean([p_["phases_ms"]["pf_emit"] for p_ in per_rank])), 4),
                "pf_emit_max_over_mean": round(max(p_["phases_ms"]["pf_emit"] for p_ in per_rank) / max(1e-9, float(np.mean([p_["phases_ms"]["pf_emit"] for p_ in per_rank]))), 4),
                "step_ms_max": round(max(p_["step_ms_this_rank"] for p_ in per_rank), 4)},
            "hard_leg": hard,
            "exact_leg": exact_leg,
            "other_configs": others,
            "lib": prov,
            "collected_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
            "rotated_batches": args.rotate_batches,
            **({"diagnostic": f"emulated shard {args.emulate_shard}: NOT a bench line"} if args.emulate_shard else {}),
            "phases_ms": {"inference": round(float(res["phases_resident"][0] if res.get("overlapped") and "phases_resident" in res else phases[0]), 4), "route_pack": round(float(phases[1]), 4),
                          "scan": round(float(phases[2]), 4), "merge": round(float(phases[3]), 4),
                          "pf_sample": round(float(phases[5]), 4), "pf_emit": round(float(phases[6]), 4),
                          "rescore": round(float(phases[7]), 4), "fallback": round(float(phases[8]), 4)},
        }
        # every leg once more, compact and LAST on the line (the driver keeps the line's tail): queries/s, ms per step, the dominant
        # kernel's roofline fraction and duration, whether the oracle check passed (it asserts: a leg that ran it passed it)
        def leg(v, ms, roofv, checked, **extra):
            out = {"qps": None if v is None else round(v), "ms": None if ms is None else round(ms, 4),
                   "frac": None if not roofv else roofv.get("frac"), "kernel_ms": None if not roofv else roofv.get("avg_launch_ms"),
                   "oracle_ok": bool(checked)}
            if roofv and roofv.get("shader_clock_mhz_under_kernel"):   # the clock THIS box held under the kernel (boxes differ: 1.65-1.88 GHz under pass 2)
                out["mhz"] = round(roofv["shader_clock_mhz_under_kernel"])
            out.update(extra)
            return out
        legs = {"c2": leg(result["value"], result["ms_per_step"], roof, cpu is not None, recall=result["recall_at_10"])}
        if value_loop == "resident":   # `value`: inputs and results in HBM; beside it the PCIe-inclusive host -> host rate of the same steps
            legs["c2"]["host_to_host_qps"] = round(result["host_to_host"]["value"])
        elif isinstance(result.get("resident"), dict):
            legs["c2"]["hbm_resident_qps"] = round(result["resident"]["value"])
        if hard:
            legs["hard"] = leg(hard["value"], hard["ms_per_step"], hard.get("roofline"), hard.get("oracle_check"), recall=hard["recall_at_10"])
        if exact_leg:
            legs["exact"] = leg(exact_leg["value"], exact_leg["ms_per_step"], exact_leg["roofline"], exact_leg.get("oracle_check"), same_as_c2=exact_leg["identical_to_default_leg"])
        for cname in ("c1", "c5"):
            if others and cname in others:
                o_ = others[cname]
                legs[cname] = leg(o_["value"], o_["ms_per_step"], o_["roofline"], o_.get("oracle_check"), recall=o_["recall_at_10"])
        if nb100k:
            legs["nb100k_2level"] = leg(nb100k["value"], nb100k["ms_per_step"], None, nb100k["oracle_check"], recall=nb100k["recall_at_10"],
                                        nav_ms=nb100k["phases_ms"]["inference"], ref_published_qps=1290,
                                        cpu_ref_qps=None if not nb100k["cpu_baseline"] else nb100k["cpu_baseline"]["value"])
        result["legs"] = legs
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
