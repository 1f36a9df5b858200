"""GPU (`-m gpu`): bucket-sharded mode on one card.  The one GPU of the test box plays every rank in
turn (one handle per rank with its `owned` mask); the all-gather is a `torch.stack` of the ranks'
blocks -- bit-identical to what RCCL delivers -- and lmi_merge_gathered must reproduce the
single-handle result exactly for world sizes 2, 3 and 8 (SURVEY section 8e identity requirement)."""
import numpy as np
import pytest
import torch

from helpers import inputs_for, layers_from, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,nb,k,world", [("G3", 4, 10, 2), ("G3", 4, 10, 8), ("G4", 4, 10, 3),
                                              ("G4", 3, 15, 2), ("G4", 1, 5, 2), ("G5", 4, 10, 8)])
def test_sharded_equals_single(oracle, name, nb, k, world):
    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.sharded import ShardedSearcher, assign_buckets, estimate_bucket_work, row_slice

    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    layers = layers_from(g)
    dp = g["data_prediction"]
    L = layers[-1][0].shape[0]
    sizes = np.bincount(dp[:, 0], minlength=L)
    dev = torch.device("cuda", 0)
    qn, qs = torch.from_numpy(Qn).to(dev), torch.from_numpy(Qs).to(dev)
    nq = Qn.shape[0]
    stream = torch.cuda.current_stream().cuda_stream

    single = _capi.Index(0, chunk_rows=256)
    single.set_stream(stream)
    single.set_mlp(layers)
    single.set_buckets(Xs, dp[:, 0], L)
    # bucket ownership from the build-time work estimate (a data sample routed like queries)
    work = estimate_bucket_work(single, torch.from_numpy(Xn[:2000]).to(dev), nb, sizes)
    assert work.shape == sizes.shape and (work[sizes > 0] > 0).all() and (work[sizes == 0] == 0).all()
    owner = assign_buckets(sizes, world, weights=work)
    sd, si, sbo = ShardedSearcher(single, 0, 1).search(qn, qs, nb, k)
    torch.cuda.synchronize()
    kout = si.shape[1]
    do, io, _ = oracle.search(layers, Qn, Xs, Qs, dp, nb, k, nthreads=4)
    np.testing.assert_array_equal(si.cpu().numpy().view(np.uint32), io)
    np.testing.assert_array_equal(sd.cpu().numpy().astype(np.float64), do)

    blocks = []
    handles = []
    bo_parts = []
    for r in range(world):
        h = _capi.Index(0, chunk_rows=256)
        h.set_stream(stream)
        h.set_mlp(layers)
        h.set_buckets(Xs, dp[:, 0], L, owned=(owner == r).astype(np.uint8))
        assert h.bucket_sizes().sum() == sizes[owner == r].sum()
        blk = torch.empty((3, nq, kout), dtype=torch.int32, device=dev)
        h.search_device(qn, qs, nb, k, blk[0], blk[1], blk[2], None)
        # query-sharded inference (ShardedSearcher's default for world > 1): the rank routes its slice only;
        # scanning with the gathered order must give the same block
        per, lo, hi = row_slice(nq, r, world)
        part = torch.full((per, nb), -1, dtype=torch.int32, device=dev)
        if hi > lo:
            h.mlp_topk_device(qn[lo:hi], nb, part[: hi - lo])
        bo_parts.append(part)
        blocks.append(blk)
        handles.append(h)
    bo_cat = torch.cat(bo_parts)[:nq].contiguous()  # == all_gather_rows' result
    torch.cuda.synchronize()
    assert torch.equal(bo_cat, sbo)
    for r, h in enumerate(handles):
        blk2 = torch.empty((3, nq, kout), dtype=torch.int32, device=dev)
        h.scan_topk_device(qs, bo_cat, nb, k, blk2[0], blk2[1], blk2[2])
        torch.cuda.synchronize()
        assert torch.equal(blk2, blocks[r])
    gathered = torch.stack(blocks).contiguous()  # == all_gather_into_tensor's result
    out_d = torch.empty((nq, kout), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, kout), dtype=torch.int32, device=dev)
    handles[0].merge_gathered(gathered[0, 0], gathered[0, 1], gathered[0, 2], world, nq, kout, out_d, out_i,
                              world_stride=3 * nq * kout)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out_i.cpu().numpy(), si.cpu().numpy())
    np.testing.assert_array_equal(out_d.cpu().numpy(), sd.cpu().numpy())
    # host-pointer form of the merge (dense layout)
    gd = np.ascontiguousarray(gathered[:, 0].cpu().numpy().view(np.float32))
    gi = np.ascontiguousarray(gathered[:, 1].cpu().numpy().view(np.uint32))
    gk = np.ascontiguousarray(gathered[:, 2].cpu().numpy().view(np.uint32))
    hd = np.empty((nq, kout), np.float32)
    hi = np.empty((nq, kout), np.uint32)
    handles[0].merge_gathered(gd, gi, gk, world, nq, kout, hd, hi)
    np.testing.assert_array_equal(hi, io)
    for h in handles + [single]:
        h.close()


@pytest.mark.parametrize("name,nb,k,world", [("G3", 4, 10, 2), ("G4", 3, 15, 3), ("G5", 4, 10, 8), ("G4", 1, 5, 7)])
def test_query_sharded_replica_slices_equal_whole_batch(oracle, name, nb, k, world):
    """ReplicaSearcher's premise on the GPU path: a query's answer (prefilter bounds included) does not depend on the rest
    of its batch -- the ranks' slices, concatenated as the all-gather does, are the whole batch's answer bit for bit."""
    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.sharded import ReplicaSearcher, row_slice

    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    layers = layers_from(g)
    dp = g["data_prediction"]
    dev = torch.device("cuda", 0)
    qn, qs = torch.from_numpy(Qn).to(dev), torch.from_numpy(Qs).to(dev)
    nq = Qn.shape[0]
    h = _capi.Index(0, chunk_rows=256)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    h.set_mlp(layers)
    h.set_buckets(Xs, dp[:, 0], layers[-1][0].shape[0])
    wd, wi, wbo = (t.clone() for t in ReplicaSearcher(h, 0, 1).search(qn, qs, nb, k))
    torch.cuda.synchronize()
    do, io, boo = oracle.search(layers, Qn, Xs, Qs, dp, nb, k, nthreads=4)
    np.testing.assert_array_equal(wi.cpu().numpy().view(np.uint32), io)
    np.testing.assert_array_equal(wd.cpu().numpy().astype(np.float64), do)
    np.testing.assert_array_equal(wbo.cpu().numpy(), boo[:, :, 0])
    parts = []
    for r in range(world):
        per, lo, hi = row_slice(nq, r, world)
        if hi > lo:
            parts.append(tuple(t.clone() for t in ReplicaSearcher(h, 0, 1).search(qn[lo:hi].contiguous(), qs[lo:hi].contiguous(), nb, k)))
    torch.cuda.synchronize()
    for j, whole in enumerate((wd, wi, wbo)):
        assert torch.equal(torch.cat([p[j] for p in parts]), whole)
    h.close()


def test_library_rccl_exchange_world1(oracle):
    """lmi_comm_* / lmi_allgather_merge: the exchange step through RCCL inside the library, exercised on the one test
    GPU with a single-rank communicator (a real ncclCommInitRank + ncclAllGather on the handle's stream)."""
    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.sharded import ShardedSearcher

    g = load_golden("G3")
    Xn, Qn, Xs, Qs = inputs_for("G3", g)
    layers = layers_from(g)
    dp = g["data_prediction"]
    L = layers[-1][0].shape[0]
    dev = torch.device("cuda", 0)
    idx = _capi.Index(0)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.set_mlp(layers)
    idx.set_buckets(Xs, dp[:, 0], L)
    comm = idx.comm_init(0, 1, _capi.Index.comm_unique_id())
    qn, qs = torch.from_numpy(Qn).to(dev), torch.from_numpy(Qs).to(dev)
    d, i, bo = ShardedSearcher(idx, 0, 1, lib_comm=comm).search(qn, qs, 4, 10)
    torch.cuda.synchronize()
    do, io, _ = oracle.search(layers, Qn, Xs, Qs, dp, 4, 10, nthreads=4)
    np.testing.assert_array_equal(i.cpu().numpy().view(np.uint32), io)
    np.testing.assert_array_equal(d.cpu().numpy().astype(np.float64), do)
    _capi.Index.comm_destroy(comm)
    idx.close()


def test_bench_self_launch_two_ranks_gloo():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the bench starts its ranks itself (child torchrun, before any
    GPU call in the parent) and runs the whole N > 1 path -- owned-only ingest, both collective layouts, the merge --
    here with the two ranks sharing the one card and the collectives over gloo (LMI_BENCH_BACKEND=gloo)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LMI_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--n", "300000", "--nq", "2000",
                        "--steps", "3", "--warmup", "1", "--epochs", "40", "--train-rows", "50000"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    assert j["sharded_alt_mode"]["value"] > 0 and len(j["per_rank"]) == 2
    assert {p["rank"] for p in j["per_rank"]} == {0, 1}
    assert j["recall_at_10"] is not None and j["recall_at_10"] > 0.9


def test_random_shardings_equal_single():
    """A fuzz of the bucket-sharded identity: random shapes (d 8..800, 3..300 buckets with empty / tiny / heavy ones, top-1..6,
    k 1..20), random world sizes 2..9 and the deterministic bucket assignment; every rank scans the whole batch on the buckets it owns
    (host-pointer C ABI with keys), the blocks are stacked as the all-gather delivers them and lmi_merge_gathered must reproduce the
    single-handle answer bit for bit.  40 cases."""
    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.sharded import assign_buckets

    for case in range(40):
        rs = np.random.RandomState(5000 + case)
        d = int(rs.choice([8, 45, 64, 96, 130, 256, 768, 800]))
        L = int(rs.choice([3, 12, 60, 300]))
        nb = int(min(L, rs.choice([1, 2, 3, 4, 6])))
        k = int(min(10 * nb, rs.choice([1, 5, 10, 10, 15, 20]))) if nb > 1 else 10
        N = int(rs.choice([500, 5000, 30000]))
        w = 1.0 / (1.0 + np.arange(L) / 4.0)
        labels = rs.choice(L, N, p=w / w.sum()).astype(np.int64)
        labels[labels == 1] = 0                                   # an empty bucket
        X = rs.randn(L, d).astype(np.float32)[labels] * 0.7 + rs.randn(N, d).astype(np.float32)
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        nq = int(rs.choice([1, 33, 700]))
        order = np.stack([rs.permutation(L)[:nb] for _ in range(nq)]).astype(np.int32)
        Q = rs.randn(nq, d).astype(np.float32)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
        world = int(rs.randint(2, 10))
        sizes = np.bincount(labels, minlength=L)
        owner = assign_buckets(sizes, world, weights=sizes.astype(np.float64) * (1 + rs.rand(L)))
        single = _capi.Index(0, chunk_rows=256)
        single.set_buckets(X, labels, L)
        sd, si = single.scan_topk(Q, order, k)
        kout = si.shape[1]
        gd = np.empty((world, nq, kout), np.float32)
        gi = np.empty((world, nq, kout), np.uint32)
        gk = np.empty((world, nq, kout), np.uint32)
        for r in range(world):
            h = _capi.Index(0, chunk_rows=256)
            h.set_buckets(X, labels, L, owned=(owner == r).astype(np.uint8))
            gd[r], gi[r], gk[r] = h.scan_topk(Q, order, k, want_keys=True)
            h.close()
        md = np.empty((nq, kout), np.float32)
        mi = np.empty((nq, kout), np.uint32)
        single.merge_gathered(gd, gi, gk, world, nq, kout, md, mi)
        desc = dict(case=case, d=d, L=L, nb=nb, k=k, N=N, nq=nq, world=world)
        np.testing.assert_array_equal(mi, si, err_msg=str(desc))
        np.testing.assert_array_equal(md, sd, err_msg=str(desc))
        single.close()
