#!/usr/bin/env python3
"""bench.py -- queries/sec of the LearnedMetricIndex query hot path on MI355X.

A "step" = one LearnedIndex.search of the whole query batch: MLP forward -> top-n_buckets ->
routing -> bucket scan -> merge, inputs and outputs resident in HBM (the PCIe-inclusive figure is
in DESIGN.md).  Default workload = BASELINE.json configs[1]: 10M x 768 synthetic unit-norm vectors
(LAION-10M shape), 120 leaves, MLP-4 (768->512->120), top-4 buckets, 10k queries, 1 x MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: the index is bucket-sharded over the ranks, every rank answers the same batch on its own
buckets and ONE RCCL all-gather + merge kernel produces the result (total work fixed: "strong").
Rank 0 prints ONE JSON line (see the keys at the bottom).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--n", type=int)
    ap.add_argument("--nq", type=int)
    ap.add_argument("--nb", type=int)
    ap.add_argument("--leaves", type=int)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--seed", type=int, default=2023)
    ap.add_argument("--train-rows", type=int, default=200_000)
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--recall-queries", type=int, default=1000)
    ap.add_argument("--cpu-queries", type=int, default=256)
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads (0: min(16, host cores))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--chunk-rows", type=int, default=None)
    ap.add_argument("--timing-level", type=int, default=2, choices=(0, 1, 2),
                    help="lmi_set_timing: 2 (default) times every phase with hipEvents -- the roofline needs the dominant "
                         "kernel's duration; 0/1 show the step without the events' bubbles (roofline fields then null/0)")
    ap.add_argument("--exact", action="store_true",
                    help="all-f32 scan (lmi_set_prefilter(0)) instead of fp16 prefilter + exact re-rank; same results")
    ap.add_argument("--emulate-shard", default=None, metavar="R/W",
                    help="diagnostic, single GPU: own only the buckets rank R of a W-way sharded run would own "
                         "(no collective); shows the per-rank step time of the N>1 bench on one card")
    ap.add_argument("--traffic-json", default=None,
                    help="PMC-derived HBM bytes per scan launch (default: profiles/scan_pmc_<config>.json, "
                         "written by profiles/summarize.py from separate rocprofv3 --pmc passes of this bench)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for key in ("n", "nq", "nb", "leaves"):
        if getattr(args, key) is not None:
            cfg[key] = getattr(args, key)
    N, d, L, nb, nq, k = cfg["n"], cfg["d"], cfg["leaves"], cfg["nb"], cfg["nq"], args.k

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.li.model import NeuralNetwork, linear_layers
    from learnedmetricindex_amd.sharded import ShardedSearcher, assign_buckets, estimate_bucket_work

    t_setup = time.time()
    # ------------------------------------------------------------------ synthetic data
    gcpu = torch.Generator().manual_seed(args.seed)
    centres = torch.randn(L, d, generator=gcpu).to(dev)

    def gen_rows(tag: int, piece: int, n: int):
        g = torch.Generator(device=dev).manual_seed(args.seed * 1_000_003 + tag * 100_003 + piece)
        a = torch.randint(0, L, (n,), generator=g, device=dev)
        x = centres[a] + torch.randn(n, d, generator=g, device=dev)
        return torch.nn.functional.normalize(x, dim=1).contiguous()

    pieces = [(p, min(CHUNK, N - p * CHUNK)) for p in range((N + CHUNK - 1) // CHUNK)]
    queries = gen_rows(7, 0, nq)  # fresh draws, not members of the set

    # ------------------------------------------------------------------ MLP: k-means labels -> Adam/CE
    net = NeuralNetwork(input_dim=d, output_dim=L, lr=0.01, model_type=cfg["model"])
    if rank == 0:
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
    if world > 1:
        for p_ in net.model.parameters():
            dist.broadcast(p_.data, src=0)
    layers = linear_layers(net.model)

    # ------------------------------------------------------------------ placement: argmax MLP(x) over all N
    eng = _capi.Index(local_rank, chunk_rows=args.chunk_rows, prefilter=not args.exact)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_mlp(layers)
    eng.set_timing(args.timing_level)
    labels = torch.empty(N, dtype=torch.int32, device=dev)
    for p, n in pieces:
        x = gen_rows(1, p, n)
        eng.mlp_topk_device(x, 1, labels[p * CHUNK: p * CHUNK + n])
    torch.cuda.synchronize()
    labels_h = labels.cpu().numpy().astype(np.int64)
    sizes = np.bincount(labels_h, minlength=L)
    # Expected scan work of a bucket = its rows x the queries it will receive.  The second factor is estimated
    # at build time by routing a sample of the DATA through the MLP (top-nb, like a query): every rank computes
    # the same estimate, no knowledge of the query batch is used.
    work_w = estimate_bucket_work(eng, gen_rows(1, 0, pieces[0][1])[: min(20_000, N)], nb, sizes)
    owner = assign_buckets(sizes, world, weights=work_w)
    owned = (owner == rank).astype(np.uint8) if world > 1 else None
    shard_world = world
    if args.emulate_shard:
        assert world == 1, "--emulate-shard is a single-GPU diagnostic"
        er, ew = (int(v) for v in args.emulate_shard.split("/"))
        shard_world = ew
        owner = assign_buckets(sizes, ew, weights=work_w)
        owner_all = owner.copy()
        owned = (owner == er).astype(np.uint8)
        owner = np.where(owner == er, 0, -1)
        args.no_cpu_baseline = args.no_recall = True
    eng.buckets_begin(labels_h, d, L, owned=owned)
    for p, n in pieces:
        eng.add_rows(gen_rows(1, p, n), p * CHUNK)
        torch.cuda.synchronize()
    eng.buckets_end()
    del labels
    torch.cuda.empty_cache()
    if rank == 0:
        log(f"[bench] index built in {time.time() - t_setup:.1f}s: N={N} d={d} L={L} bucket sizes "
            f"min/median/max = {sizes.min()}/{int(np.median(sizes))}/{sizes.max()}, empty={int((sizes == 0).sum())}")

    # ------------------------------------------------------------------ the timed region
    searcher = ShardedSearcher(eng, rank, world)

    def step():
        return searcher.search(queries, queries, nb, k)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    dom_slot = _capi.T_SCAN if args.exact else _capi.T_PF_EMIT
    eng.timings_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_d, out_i, bo = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    # hipEvents recorded on the kernels' own stream around every phase of every step; read once, after the
    # timed region (the handle keeps the newest 128 sets), so the loop itself has no host synchronisation
    phases, n_timed = eng.timings_mean()
    phases = phases * searcher.calls_per_search  # a search with sharded inference is two C-ABI calls (MLP slice, scan)
    tm = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    elapsed = float(tm.item())
    flops, pairs, items = eng.scan_stats()
    if rank == 0 and shard_world > 1:  # how even the bucket assignment turned out for this batch
        own = owner_all if args.emulate_shard else owner
        bo_h = bo.cpu().numpy().ravel()
        bo_h = bo_h[(bo_h >= 0) & (bo_h < L)]
        per_rank = np.bincount(own[bo_h], weights=sizes[bo_h].astype(np.float64), minlength=shard_world)
        log(f"[bench] scan work per rank (pairs, share of the mean): {np.round(per_rank / per_rank.mean(), 3).tolist()}")
    pf_active, pf_survivors, pf_fallbacks = eng.prefilter_stats()
    # ------------------------------------------------------------------ recall@10 vs exact brute force
    recall = None
    if not args.no_recall:
        nr = min(args.recall_queries, nq)
        gt_d, gt_i, _ = ShardedSearcher(eng, rank, world).search(queries[:nr].contiguous(), queries[:nr].contiguous(), L, k)
        got = out_i[:nr].cpu().numpy().astype(np.int64)
        gt = gt_i.cpu().numpy().astype(np.int64)
        recall = float(np.mean([len(set(a) & set(b)) / float(k) for a, b in zip(got, gt)]))

    # ------------------------------------------------------------------ host boundary (PCIe inclusive), N=1
    # host numpy in -> host numpy out through the C ABI with host pointers (pageable memory): reported
    # beside the bench line, never as `value`
    host_boundary = None
    if rank == 0 and world == 1 and not args.emulate_shard:
        qh_all = queries.cpu().numpy()
        eng.search(qh_all, qh_all, nb, k)
        t_h = time.perf_counter()
        for _ in range(3):
            hd, hi, _hb = eng.search(qh_all, qh_all, nb, k)
        t_h = (time.perf_counter() - t_h) / 3
        assert np.array_equal(hi.view(np.int32), out_i.cpu().numpy().view(np.int32)), "host-pointer path differs"
        host_boundary = {"value": round(nq / t_h, 1), "unit": "queries/s", "ms_per_batch": round(t_h * 1e3, 3),
                         "what": "lmi_search with host pointers: pageable query upload + search + result download"}

    # ------------------------------------------------------------------ CPU baseline (oracle), rank 0, N=1
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import lmi_oracle

        ns = min(args.cpu_queries, nq)
        nthr = args.cpu_threads or min(16, os.cpu_count() or 1)
        qh = queries[:ns].cpu().numpy()
        t_cpu = time.perf_counter()
        order_o = lmi_oracle.precompute_bucket_order(layers, qh, nb, nthreads=nthr)
        t_cpu = time.perf_counter() - t_cpu
        assert np.array_equal(order_o[:, :, 0], bo[:ns].cpu().numpy()), "oracle bucket order differs from the GPU's"
        # reference structure (LearnedIndex.py:107-146, 350-371): for every visited bucket, per rank,
        # knn over the bucket + 1 - sim + id mapping, then the stable merge; only that work is timed,
        # not the device->host copy of the bucket.
        rank_d = np.full((nb, ns, 10), np.inf)
        rank_i = np.zeros((nb, ns, 10), dtype=np.uint32)
        for b in np.unique(order_o[:, :, 0]):
            rows, ids = eng.read_bucket(int(b))
            if rows.shape[0] == 0:
                continue
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
        same = bool(np.array_equal(fi, out_i[:ns].cpu().numpy().view(np.uint32)) and
                    np.array_equal(fd, out_d[:ns].cpu().numpy().astype(np.float64)))
        assert same, "CPU oracle and GPU results differ on the sampled queries"
        cpu = {"value": round(ns / t_cpu, 3), "unit": "queries/s", "cores": nthr, "kind": "port",
               "sample": f"first {ns} of {nq} queries, all {nb} ranks, full {N}x{d} index, bucket by bucket like "
                         f"LearnedIndex.py:107-146/350-371; oracle/lmi_oracle.c (canonical fmaf chain, OpenMP over "
                         f"the bucket's rows, {nthr} threads); device->host copies of the buckets not timed; "
                         f"ids and distances identical to the GPU's"}

    if rank == 0:
        scan_s = float(phases[_capi.T_SCAN]) * 1e-3
        dom_s = max(float(phases[dom_slot]) * 1e-3, 1e-12)  # 0 when --timing-level < 2: the roofline fields are meaningless then
        scan_s = max(scan_s, 1e-12)
        visited = np.unique(bo.cpu().numpy())
        visited = visited[(visited >= 0) & (owner[np.clip(visited, 0, L - 1)] == rank)]
        rows_visited = float(sizes[visited].sum())
        # Dominant kernel and its roofline.  Algorithmic work per launch: flops = 2*d*sum over (query, rank)
        # slots of the bucket size (sharded runs: this rank's slots); bytes = every visited bucket read once
        # in the kernel's operand type + the packed queries once.
        if args.exact:
            kernel, op_bytes, peak_tf = "lmi::scan_kernel", 4, PEAK_F32_MFMA_TFLOPS
        else:
            kernel, op_bytes, peak_tf = "lmi::prefilter_kernel<false, 2>", 2, PEAK_F16_MFMA_TFLOPS
        alg_bytes = op_bytes * d * (rows_visited + nq * nb)
        t_mfma, t_hbm = flops / (peak_tf * 1e12), alg_bytes / (PEAK_HBM_GBS * 1e9)
        if t_mfma >= t_hbm:
            roof = {"bound": "mfma", "achieved": round(flops / dom_s / 1e12, 3), "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": round(flops / dom_s / 1e12 / peak_tf, 4)}
        else:
            roof = {"bound": "hbm", "achieved": round(alg_bytes / dom_s / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(alg_bytes / dom_s / 1e9 / PEAK_HBM_GBS, 4)}
        traffic = None
        tj = args.traffic_json or os.path.join(ROOT, "profiles", f"scan_pmc_{args.config}{'_exact' if args.exact else ''}.json")
        overridden = any(getattr(args, key) is not None for key in ("n", "nq", "nb", "leaves", "emulate_shard"))
        if world == 1 and not overridden and os.path.exists(tj):
            traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
        roof.update({"traffic": traffic, "kernel": kernel, "flops_per_launch": flops, "bytes_per_launch": alg_bytes,
                     "avg_launch_ms": round(dom_s * 1e3, 4), "launches_timed": int(n_timed) // searcher.calls_per_search,
                     "floors_ms": {"mfma": round(t_mfma * 1e3, 3), "hbm": round(t_hbm * 1e3, 3)},
                     # the whole scan phase (all its kernels) priced as SURVEY 8d does: algorithmic f32 flops
                     # against the f32 MFMA peak, whatever precision the prefilter used
                     "scan_phase_f32_equiv": {"achieved": round(flops / scan_s / 1e12, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                                              "unit": "TFLOP/s", "frac": round(flops / scan_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                              "avg_ms": round(scan_s * 1e3, 4)}})
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
            "recall_at_10": None if recall is None else round(recall, 5),
            "config": {"workload": f"{N}x{d} unit-norm gaussian-mixture vectors, 1-level LMI ({L} leaves, "
                                   f"{cfg['model']} {d}->512->{L} trained {args.epochs} epochs), top-{nb} buckets, "
                                   f"{nq}-query batch, k={k}",
                       "baseline_config": args.config, "parallelism": "single GPU" if world == 1 else
                                      f"bucket-sharded x{world}: MLP on 1/{world} of the batch + all-gather of the bucket order, "
                                      f"scan of the owned buckets + all-gather of the per-rank top-k",
                       "scan_pairs": int(pairs), "scan_items": int(items)},
            "roofline": roof,
            "prefilter": None if args.exact else {"survivors_per_slot": round(pf_survivors / max(1, nq * nb), 2),
                                                   "fallback_slots": int(pf_fallbacks)},
            "cpu_baseline": cpu,
            "host_boundary": host_boundary,
            **({"diagnostic": f"emulated shard {args.emulate_shard}: NOT a bench line"} if args.emulate_shard else {}),
            "phases_ms": {"inference": round(float(phases[0]), 4), "route_pack": round(float(phases[1]), 4),
                          "scan": round(float(phases[2]), 4), "merge": round(float(phases[3]), 4),
                          "pf_sample": round(float(phases[5]), 4), "pf_emit": round(float(phases[6]), 4),
                          "rescore": round(float(phases[7]), 4), "fallback": round(float(phases[8]), 4)},
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
