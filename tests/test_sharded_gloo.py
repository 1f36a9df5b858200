"""CPU (`-m "not gpu"`): the N>1 path with world_size 2, 4 and 8 over `gloo`.

Covers the host logic of the bucket-sharded mode -- deterministic bucket assignment, the
[dists|ids|keys] block packing, the ONE all-gather, the (dist, key) merge order -- and the identity
requirement of SURVEY section 8e: the merged result equals the single-process result bit for bit.
The per-rank scan itself is played by the CPU oracle here (there is no GPU in this container); on
the GPU the same `sharded.py` code drives lmi_search / lmi_merge_gathered (tests/test_gpu_sharded.py).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def local_block(oracle, layers, Qn, Xs, Qs, dp, nb, kout, owned):
    """What lmi_search(+keys) returns on a rank owning `owned` buckets: oracle restatement."""
    bo = oracle.precompute_bucket_order(layers, Qn, nb)
    groups = [(p, rows) for p, rows in oracle.group_buckets(dp) if owned[p[0]]]
    ids = np.arange(1, Xs.shape[0] + 1)
    nq = Qs.shape[0]
    D = np.empty((nq, nb * 10), dtype=np.float64)
    I = np.empty((nq, nb * 10), dtype=np.uint32)
    K = np.empty((nq, nb * 10), dtype=np.int64)
    for r in range(nb):
        d, n = oracle.search_single_bucket(Xs, ids, groups, Qs, bo[:, r, :])
        D[:, r * 10:(r + 1) * 10], I[:, r * 10:(r + 1) * 10] = d, n
        K[:, r * 10:(r + 1) * 10] = r * 16 + np.arange(10)[None, :]
    order = np.lexsort((K, D), axis=1)[:, :kout]
    return (np.take_along_axis(D, order, 1).astype(np.float32), np.take_along_axis(I, order, 1),
            np.take_along_axis(K, order, 1).astype(np.uint32))


def _worker(rank, world, port, name, nb, k, out_dir):
    sys.path[:0] = [ROOT, HERE, os.path.join(HERE, "golden")]
    from helpers import inputs_for, layers_from, load_golden
    from learnedmetricindex_amd import sharded
    from oracle import lmi_oracle as oracle

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g = load_golden(name)
        Xn, Qn, Xs, Qs = inputs_for(name, g)
        layers = layers_from(g)
        dp = g["data_prediction"]
        L = layers[-1][0].shape[0]
        sizes = np.bincount(dp[:, 0], minlength=L)
        owner = sharded.assign_buckets(sizes, world)
        kout = 10 if nb == 1 else k
        d, i, keys = local_block(oracle, layers, Qn, Xs, Qs, dp, nb, kout, owner == rank)
        blk = torch.from_numpy(sharded.pack_block(np, d, i, keys))
        gathered = sharded.all_gather_blocks(blk, world)            # the one collective
        md, mi = sharded.merge_blocks_numpy(gathered.numpy(), kout)
        # query-sharded inference: this rank routes its slice of the batch only, the bucket order is gathered
        nq = Qn.shape[0]
        per, lo, hi = sharded.row_slice(nq, rank, world)
        loc = torch.full((per, nb), -1, dtype=torch.int32)
        if hi > lo:
            loc[: hi - lo] = torch.from_numpy(oracle.precompute_bucket_order(layers, Qn[lo:hi], nb)[:, :, 0].astype(np.int32))
        bo = sharded.all_gather_rows(loc, nq, world).numpy()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), d=md, i=mi, owner=owner, bo=bo)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name,nb,k,world", [("G1", 3, 10, 2), ("G4", 4, 10, 2), ("G4", 3, 15, 2), ("G4", 1, 5, 2),
                                              ("G1", 3, 10, 4), ("G4", 4, 10, 8), ("G4", 3, 15, 8)])
def test_multi_rank_result_identical_to_single(oracle, tmp_path, name, nb, k, world):
    from helpers import inputs_for, layers_from, load_golden

    mp.spawn(_worker, args=(world, _free_port(), name, nb, k, str(tmp_path)), nprocs=world, join=True)
    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    do, io, boo = oracle.search(layers_from(g), Qn, Xs, Qs, g["data_prediction"], nb, k)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in ranks:
        np.testing.assert_array_equal(r["bo"], boo[:, :, 0])          # gathered bucket order == unsharded routing
        np.testing.assert_array_equal(r["owner"], ranks[0]["owner"])  # same assignment on every rank
    assert set(np.unique(ranks[0]["owner"])) == set(range(world))    # every rank owns a bucket (L >= 12 >= world)
    for r in ranks:
        np.testing.assert_array_equal(r["i"], io)
        np.testing.assert_array_equal(r["d"].astype(np.float64), do)


class _OracleIndex:
    """Stands in for `_capi.Index` under ReplicaSearcher on the CPU: lmi_search played by the oracle on CPU tensors."""

    def __init__(self, oracle, layers, Xs, dp):
        self.oracle, self.layers, self.Xs, self.dp = oracle, layers, Xs, dp

    @staticmethod
    def kout(nb, k):
        return 10 if nb == 1 else k

    def search_device(self, qn_t, qs_t, nb, k, d_t, i_t, keys_t=None, bo_t=None):
        d, i, bo = self.oracle.search(self.layers, qn_t.numpy(), self.Xs, qs_t.numpy(), self.dp, nb, k)
        d_t.copy_(torch.from_numpy(d.astype(np.float32)))
        i_t.copy_(torch.from_numpy(i.view(np.int32)))
        if bo_t is not None:
            bo_t.copy_(torch.from_numpy(bo[:, :, 0].astype(np.int32)))


def _replica_worker(rank, world, port, name, nb, k, out_dir):
    sys.path[:0] = [ROOT, HERE, os.path.join(HERE, "golden")]
    from helpers import inputs_for, layers_from, load_golden
    from learnedmetricindex_amd import sharded
    from oracle import lmi_oracle as oracle

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g = load_golden(name)
        Xn, Qn, Xs, Qs = inputs_for(name, g)
        idx = _OracleIndex(oracle, layers_from(g), Xs, g["data_prediction"])
        d, i, bo = sharded.ReplicaSearcher(idx, rank, world).search(torch.from_numpy(Qn), torch.from_numpy(Qs), nb, k)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), d=d.numpy(), i=i.numpy().view(np.uint32), bo=bo.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,nb,k,world", [("G1", 3, 10, 2), ("G4", 3, 15, 2), ("G4", 1, 5, 2), ("G4", 4, 10, 8)])
def test_query_sharded_replicas_identical_to_single(oracle, tmp_path, name, nb, k, world):
    """SURVEY section 8e's other mode: every rank answers its slice of the batch against the whole index; the one
    all-gather of [dists | ids | bucket order] rows must reproduce the single-process answer (also when the batch does
    not divide by the world size and when a rank's slice is empty)."""
    from helpers import inputs_for, layers_from, load_golden

    mp.spawn(_replica_worker, args=(world, _free_port(), name, nb, k, str(tmp_path)), nprocs=world, join=True)
    g = load_golden(name)
    Xn, Qn, Xs, Qs = inputs_for(name, g)
    do, io, boo = oracle.search(layers_from(g), Qn, Xs, Qs, g["data_prediction"], nb, k)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        np.testing.assert_array_equal(got["i"], io)
        np.testing.assert_array_equal(got["d"].astype(np.float64), do)
        np.testing.assert_array_equal(got["bo"], boo[:, :, 0])


def test_assign_buckets_balances_and_is_deterministic():
    from learnedmetricindex_amd.sharded import assign_buckets

    rs = np.random.RandomState(0)
    sizes = rs.randint(0, 200_000, size=120)
    sizes[[3, 77]] = 0
    for world in (1, 2, 4, 8):
        owner = assign_buckets(sizes, world)
        assert owner.min() >= 0 and owner.max() < world
        np.testing.assert_array_equal(owner, assign_buckets(sizes, world))
        load = np.bincount(owner, weights=sizes, minlength=world)
        assert load.max() <= load.mean() * 1.05 + sizes.max() / world
    w2 = assign_buckets(sizes, 8, weights=sizes.astype(float) ** 2)
    load = np.bincount(w2, weights=sizes.astype(float) ** 2, minlength=8)
    assert load.max() <= load.mean() * 1.15


def test_row_slices_cover_the_batch():
    from learnedmetricindex_amd.sharded import row_slice

    for nq, world in ((10, 3), (10000, 8), (5, 8), (7, 2), (1, 1), (0, 4)):
        seen = []
        for r in range(world):
            per, lo, hi = row_slice(nq, r, world)
            assert 0 <= lo <= hi <= nq and hi - lo <= per and per * world >= nq
            assert lo == min(r * per, nq)
            seen.extend(range(lo, hi))
        assert seen == list(range(nq))
