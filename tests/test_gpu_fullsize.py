"""GPU (`-m gpu`): BASELINE.json's configurations at their full sizes, against the CPU oracle.

* C1 (configs[0]): 100 000 x 768, 120 leaves, top-4, 1 000 queries: every query oracle-checked.
* C2 (configs[1], the headline): 10M x 768, 120 buckets, top-4, 10 000 queries on one MI355X, placement and routing
  through the MLP kernels.  The fp16-prefilter mode and the all-f32 mode (two independently built indexes) must
  return bit-identical ids and distances for the WHOLE batch, and 256 sampled queries are re-computed by the oracle
  (canonical fmaf chain over every row of every visited bucket, bucket by bucket like LearnedIndex.py:107-146 /
  350-371) and must match the GPU's ids and distances exactly.  The same on the HARD generator (overlapping clusters,
  5M x 768) and at k = 15 / top-8 (per-bucket bounds) on 1.5M x 768.
* C4 (configs[3]): ONE 1/8 shard of 100M x 768, 1 024 buckets, top-8: the rank ingests the 100M labels, owns
  128 buckets (12.5M rows, 57.6 GB resident as f32 rows + fp16 fragments) and passes only its own rows in
  (lmi_buckets_add_owned_rows); L = 1 024 routing through the MLP kernels is checked against the oracle and
  sampled queries' neighbours against the oracle on the buckets the rank owns; unowned slots are (inf, 0).
The vectors are generated piecewise on the device and never exist as a whole on the host."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

D = 768


def _need_hbm(gib):
    if torch.cuda.get_device_properties(0).total_memory < gib * (1 << 30):
        pytest.skip(f"needs ~{gib} GiB of HBM")


def _oracle_check(oracle, idx, Qh, order_h, sel, nb, got_d, got_i, nthreads=16, k=10):
    """Per-rank knn over each visited bucket (read back from HBM) + stable merge == the GPU's answer on `sel`."""
    ns = sel.size
    rank_d = np.full((nb, ns, 10), np.inf)
    rank_i = np.zeros((nb, ns, 10), dtype=np.uint32)
    sizes = idx.bucket_sizes()
    for b in np.unique(order_h[sel]):
        if b < 0 or sizes[b] == 0:
            continue
        rows, ids = idx.read_bucket(int(b))
        for r in range(nb):
            rel = np.flatnonzero(order_h[sel, r] == b)
            if rel.size:
                sim, loc = oracle.knn_ip(Qh[sel[rel]], rows, 10, nthreads=nthreads)
                dd = np.float32(1) - sim                     # LearnedIndex.py:368
                ii = ids[loc]                                # -1 -> last label (numpy indexing), SURVEY Q4
                rank_d[r, rel], rank_i[r, rel] = dd, ii
    fd = fi = None
    for r in range(nb):
        fd, fi = oracle.merge_rank(fd, fi, rank_d[r], rank_i[r], k)
    np.testing.assert_array_equal(got_i[sel], fi)
    np.testing.assert_array_equal(got_d[sel].astype(np.float64), fd)


def _random_feature_mlp(centres, H, seed, gain=8.0):
    """An MLP-4 shaped net (d -> H -> L) whose classes follow the clusters loosely: hidden = random ReLU features, output = their
    correlation with the centres.  Not trained (the build container has no GPU to train on): what matters here is that placement
    AND routing go through the HIP MLP kernels end to end, as in the reference (model.py:226-241, LearnedIndexBuilder.py:76)."""
    L, d = centres.shape
    gw = torch.Generator().manual_seed(seed)
    W1 = (torch.randn(H, d, generator=gw) / d ** 0.5).numpy().astype(np.float32)
    b1 = (0.1 * torch.randn(H, generator=gw)).numpy().astype(np.float32)
    hc = np.maximum(torch.nn.functional.normalize(centres, dim=1).cpu().numpy() @ W1.T + b1, 0.0)
    W2 = (gain * (hc - hc.mean(0)) / H ** 0.5).astype(np.float32)
    return [(W1, b1), (W2, np.zeros(L, dtype=np.float32))]


def _mlp_routed_case(oracle, *, d, L, NB, n, nq, k=10, noise=1.0, centre_scale=1.0, zipf=0.0, seed=2023, n_oracle=256, piece=1 << 19, tag=""):
    """One workload END TO END through the MLP on the device: placement = argmax MLP(x) over all n rows, routing + scan + merge by
    lmi_search.  The fp16-prefilter mode and the all-f32 mode (two independently built indexes) must agree on ids and distance bits
    for the WHOLE batch; `n_oracle` sampled queries are re-computed by the oracle (bucket order: forward_logits + rank_classes;
    neighbours: canonical chain over every row of every visited bucket + stable merge to k)."""
    from learnedmetricindex_amd import _capi

    dev = torch.device("cuda", 0)
    g0 = torch.Generator(device=dev).manual_seed(seed)
    centres = torch.randn(L, d, generator=g0, device=dev) * centre_scale
    layers = _random_feature_mlp(centres, 512, seed + 1)
    w = None
    if zipf > 0:   # heavy-tailed cluster weights (bench.py's hard leg)
        w = (1.0 / (1.0 + torch.arange(L, device=dev, dtype=torch.float32) / zipf))

    def draw(g, count):
        a = torch.randint(0, L, (count,), generator=g, device=dev) if w is None else torch.multinomial(w, count, replacement=True, generator=g)
        return torch.nn.functional.normalize(centres[a] + noise * torch.randn(count, d, generator=g, device=dev), dim=1).contiguous()

    def rows(p, count):
        return draw(torch.Generator(device=dev).manual_seed(seed * 7 + 1000 + p), count)

    pieces = [(p, min(piece, n - p * piece)) for p in range((n + piece - 1) // piece)]
    Q = draw(torch.Generator(device=dev).manual_seed(seed * 7 + 77), nq)
    Qh = Q.cpu().numpy()
    sel = np.sort(np.random.RandomState(5).choice(nq, min(n_oracle, nq), replace=False))
    kout = _capi.Index.kout(NB, k)
    out, labels = [], None
    for pf in (True, False):
        idx = _capi.Index(0, prefilter=pf)
        idx.set_stream(torch.cuda.current_stream().cuda_stream)
        idx.set_mlp(layers)
        if labels is None:
            lab = torch.empty(n, dtype=torch.int32, device=dev)
            for p, c in pieces:
                idx.mlp_topk_device(rows(p, c), 1, lab[p * piece: p * piece + c])
            torch.cuda.synchronize()
            labels = lab.cpu().numpy().astype(np.int64)
            del lab
        idx.buckets_begin(labels, d, L)
        for p, c in pieces:
            idx.add_rows(rows(p, c), p * piece)
            torch.cuda.synchronize()
        idx.buckets_end()
        dd = torch.empty((nq, kout), dtype=torch.float32, device=dev)
        ii = torch.empty((nq, kout), dtype=torch.int32, device=dev)
        bo = torch.empty((nq, NB), dtype=torch.int32, device=dev)
        idx.search_device(Q, Q, NB, k, dd, ii, None, bo)
        torch.cuda.synchronize()
        dh, ih, order_h = dd.cpu().numpy(), ii.cpu().numpy().view(np.uint32), bo.cpu().numpy()
        if pf:
            sizes = idx.bucket_sizes()
            active, survivors, fallbacks = idx.prefilter_stats()
            print(f"{tag}: bucket sizes min/median/max {sizes.min()}/{int(np.median(sizes))}/{sizes.max()}, empty {int((sizes == 0).sum())}; "
                  f"{survivors / (nq * NB):.2f} survivors per slot, {fallbacks} fallback slots")
            assert active and survivors >= min(k, 10) * nq
            np.testing.assert_array_equal(order_h[sel], oracle.rank_classes(oracle.forward_logits(layers, Qh[sel], nthreads=16), NB))
            _oracle_check(oracle, idx, Qh, order_h, sel, NB, dh, ih, k=k)   # the default mode against the oracle
        out.append((dh, ih, order_h))
        idx.close()
        torch.cuda.empty_cache()
    (d1, i1, o1), (d0, i0, o0) = out
    np.testing.assert_array_equal(o1, o0)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)      # distance BITS, whole batch
    assert np.all(np.diff(d1, axis=1) >= 0) and np.all(i1 > 0)
    return d1, i1


def test_c2_full_size_modes_and_oracle(oracle):
    """C2 (configs[1], the headline): 10M x 768, 120 leaves, top-4, 10 000 queries, MLP-routed end to end (round 4: placement and
    routing through the MLP kernels like C5's test, no longer by centre similarity)."""
    _need_hbm(150)
    _mlp_routed_case(oracle, d=D, L=120, NB=4, n=10_000_000, nq=10_000, tag="C2")


def test_c1_full_size_every_query_against_the_oracle(oracle):
    """C1 (configs[0], the reference's own CPU-runnable case): 100 000 x 768, 120 leaves, top-4, 1 000 queries, MLP-routed end to
    end -- both scan modes identical for the whole batch and EVERY query re-computed by the oracle (bucket order, ids, distance bits)."""
    _mlp_routed_case(oracle, d=D, L=120, NB=4, n=100_000, nq=1_000, n_oracle=1_000, tag="C1")


def test_hard_generator_prefilter_equals_exact_and_oracle(oracle):
    """The workload where the prefilter's bound matters most (bench.py's hard leg: centres x 0.26, heavy-tailed cluster weights --
    overlapping clusters, small score gaps, recall@10 ~0.9 at top-4), 5M x 768: whole-batch prefilter == all-f32 (ids and distance
    bits) and 256 oracle-checked queries (VERDICT r03 "weak" 1 / "Next" 5a)."""
    _need_hbm(100)
    _mlp_routed_case(oracle, d=D, L=120, NB=4, n=5_000_000, nq=10_000, centre_scale=0.26, zipf=20.0, seed=4242, tag="hard")


def test_k15_top8_per_bucket_bounds_at_scale(oracle):
    """k = 15 > 10 turns the query-level bound off (a query may need more than the 10 best of one bucket's list: every slot keeps its
    own bucket's bound) and nb = 8 walks the longer rank merge: 1.5M x 768, 64 leaves, 4 000 queries; whole batch prefilter ==
    all-f32, 128 oracle-checked queries merged to k = 15 (LearnedIndex.py:125-146; per-bucket k stays 10)."""
    _need_hbm(40)
    d, i = _mlp_routed_case(oracle, d=D, L=64, NB=8, n=1_500_000, nq=4_000, k=15, centre_scale=0.5, seed=1515, n_oracle=128, tag="k15/nb8")
    assert d.shape[1] == 15


@pytest.mark.parametrize("tag,d,L,NB,n,nq", [
    ("d96-wide", 96, 64, 4, 3_000_000, 10_000),       # 625 queries per bucket: the low-dimensional kernels' WIDE form (8-wave blocks, 12-col-block tiles)
    ("d96-narrow", 96, 512, 4, 3_000_000, 10_000),    # 78 per bucket: two 4-wave blocks per CU, 8-col-block tiles, single row-blocks
    ("d128", 128, 256, 8, 3_000_000, 8_000),          # 8 k16-groups, top-8
    ("d1536", 1536, 64, 4, 1_000_000, 4_000),         # rows wider than four waves' re-rank buffers: two waves per block
    ("leaves2000", 768, 2000, 4, 2_000_000, 10_000),  # small buckets (sampling stride < 16), the routing kernels' sort at 2 000 buckets
])
def test_other_shapes_at_scale(oracle, tag, d, L, NB, n, nq):
    """The code paths round 4 added late, at sizes where they carry real work (MLP-routed end to end like C2's test): whole batch
    prefilter == all-f32 (ids and distance bits) + 128 oracle-checked queries each."""
    _need_hbm(40)
    _mlp_routed_case(oracle, d=d, L=L, NB=NB, n=n, nq=nq, seed=900 + d + L, n_oracle=128, tag=tag)


def test_c4_one_eighth_shard(oracle):
    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.sharded import assign_buckets

    _need_hbm(120)
    L, NB, n, nq, world, rank = 1024, 8, 100_000_000, 10_000, 8, 3
    piece = 1 << 19
    dev = torch.device("cuda", 0)
    g0 = torch.Generator(device=dev).manual_seed(404)
    centres = torch.randn(L, D, generator=g0, device=dev)
    # labels of all 100M objects (uneven buckets: a Zipf-ish weighting), generated piecewise on the device
    w = (1.0 / (1.0 + torch.arange(L, device=dev, dtype=torch.float32) / 200.0))
    lab_parts = []
    for p in range((n + (1 << 24) - 1) >> 24):
        g = torch.Generator(device=dev).manual_seed(9000 + p)
        cnt = min(1 << 24, n - (p << 24))
        lab_parts.append(torch.multinomial(w, cnt, replacement=True, generator=g).to(torch.int32).cpu())
    labels = torch.cat(lab_parts).numpy().astype(np.int64)
    del lab_parts
    sizes = np.bincount(labels, minlength=L)
    owner = assign_buckets(sizes, world)
    owned = (owner == rank).astype(np.uint8)
    own_rows = np.flatnonzero(owned[labels].astype(bool)).astype(np.int64)   # ~12.5M original row numbers
    assert 0.11 * n < own_rows.size < 0.14 * n

    def vectors(index_t, p):
        """rows of the objects index_t (a device int64 tensor): centre of their bucket + noise seeded per piece"""
        g = torch.Generator(device=dev).manual_seed(5000 + p)
        lab = torch.from_numpy(labels[index_t.cpu().numpy()]).to(dev)
        return torch.nn.functional.normalize(centres[lab] + torch.randn(index_t.shape[0], D, generator=g, device=dev), dim=1).contiguous()

    # random-weight MLP-4 768 -> 512 -> 1024 (L = 1 024 routing through the MLP kernels)
    rs = np.random.RandomState(17)
    layers = [((rs.randn(512, D) / np.sqrt(D)).astype(np.float32), (0.1 * rs.randn(512)).astype(np.float32)),
              ((rs.randn(L, 512) / np.sqrt(512)).astype(np.float32), (0.1 * rs.randn(L)).astype(np.float32))]
    idx = _capi.Index(0)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.set_mlp(layers)
    idx.buckets_begin(labels, D, L, owned=owned)          # the 100M-label ingest
    for p, r0 in enumerate(range(0, own_rows.size, piece)):
        it = torch.from_numpy(own_rows[r0: r0 + piece]).to(dev)
        idx.add_owned_rows(vectors(it, p), it)             # only the rank's own 12.5M rows travel
        torch.cuda.synchronize()
    idx.buckets_end()
    np.testing.assert_array_equal(idx.bucket_sizes(), np.where(owned.astype(bool), sizes, 0))

    gq = torch.Generator(device=dev).manual_seed(78)
    Q = torch.nn.functional.normalize(centres[torch.randint(0, L, (nq,), generator=gq, device=dev)]
                                      + torch.randn(nq, D, generator=gq, device=dev), dim=1).contiguous()
    Qh = Q.cpu().numpy()
    # (1) L = 1 024 routing: MLP + class ranking on the GPU == oracle, on a sample of the batch
    bo_mlp = torch.empty((nq, NB), dtype=torch.int32, device=dev)
    idx.mlp_topk_device(Q, NB, bo_mlp)
    torch.cuda.synchronize()
    selq = np.sort(np.random.RandomState(6).choice(nq, 512, replace=False))
    np.testing.assert_array_equal(bo_mlp.cpu().numpy()[selq], oracle.precompute_bucket_order(layers, Qh[selq], NB, nthreads=16)[:, :, 0])
    # (2) the scan at C4's shape: top-8 of 1 024 by centre similarity (what a trained index would route to)
    order = (Q @ centres.T).topk(NB, dim=1).indices.to(torch.int32).contiguous()
    order_h = order.cpu().numpy()
    d = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    i = torch.empty((nq, 10), dtype=torch.int32, device=dev)
    keys = torch.empty((nq, 10), dtype=torch.int32, device=dev)
    idx.scan_topk_device(Q, order, NB, 10, d, i, keys)
    torch.cuda.synchronize()
    dh, ih = d.cpu().numpy(), i.cpu().numpy().view(np.uint32)
    active, survivors, fallbacks = idx.prefilter_stats()
    assert active and fallbacks == 0
    visits_owned = owned[order_h].astype(bool)               # [nq, NB]
    none = ~visits_owned.any(axis=1)
    assert none.any() and np.all(np.isinf(dh[none])) and np.all(ih[none] == 0)   # slots of other ranks' buckets: (inf, 0)
    # queries whose owned buckets are among the 6 most visited owned buckets (bounded read-back: ~6 x 300 MB)
    hot = np.argsort(-np.bincount(order_h[visits_owned], minlength=L))[:6]
    cand = np.flatnonzero((np.isin(order_h, hot) | ~visits_owned).all(axis=1) & visits_owned.any(axis=1))
    sel = cand[:96]
    assert sel.size >= 32
    _oracle_check(oracle, idx, Qh, order_h, sel, NB, dh, ih)
    # ingest check: a bucket read back == the regenerated rows of its objects, in ascending original order
    b = int(hot[0])
    rows_b, ids_b = idx.read_bucket(b)
    expect_rows = np.flatnonzero(labels == b)
    np.testing.assert_array_equal(ids_b, (expect_rows + 1).astype(np.uint32))
    pos_in_own = np.searchsorted(own_rows, expect_rows[:50])
    for r_, p_own in zip(range(50), pos_in_own):
        pc, off = divmod(int(p_own), piece)
        it = torch.from_numpy(own_rows[pc * piece: pc * piece + piece]).to(dev)
        np.testing.assert_array_equal(vectors(it, pc)[off].cpu().numpy(), rows_b[r_])
        if r_ >= 2:
            break
    idx.close()


def test_c5_full_size(oracle):
    """C5 (configs[4]): 10M x 45 (AlphaFind protein-embedding shape; 3 k16-groups = 48 wide in the fp16 slab: the low-dimensional
    prefilter kernels, lmi_pass2_small.h), 256 leaves, cosine (= unit-norm rows + 1 - ip), top-4, 10 000 queries --
    END TO END through the MLP: an MLP-4 (45 -> 512 -> 256) places the rows (argmax, LearnedIndexBuilder.py:76) and routes the
    queries (lmi_search: MLP + top-4 + scan + merge in one call).  Both scan modes identical for the whole batch; for 256
    sampled queries the bucket order must equal the oracle's (forward_logits + rank_classes) and the neighbours the oracle's
    per-bucket knn + stable merge (reference call sites model.py:226-241, LearnedIndex.py:107-146, 360-368)."""
    from learnedmetricindex_amd import _capi

    _need_hbm(40)
    d5, L, NB, n, nq, piece, H = 45, 256, 4, 10_000_000, 10_000, 1 << 20, 512
    dev = torch.device("cuda", 0)
    g0 = torch.Generator(device=dev).manual_seed(555)
    centres = torch.randn(L, d5, generator=g0, device=dev)
    # an MLP whose classes follow the clusters loosely: hidden = random features, output = their correlation with the centres
    gw = torch.Generator().manual_seed(556)
    W1 = (torch.randn(H, d5, generator=gw) / d5 ** 0.5).numpy().astype(np.float32)
    b1 = (0.1 * torch.randn(H, generator=gw)).numpy().astype(np.float32)
    hc = np.maximum(torch.nn.functional.normalize(centres, dim=1).cpu().numpy() @ W1.T + b1, 0.0)
    W2 = (8.0 * (hc - hc.mean(0)) / H ** 0.5).astype(np.float32)
    b2 = np.zeros(L, dtype=np.float32)
    layers = [(W1, b1), (W2, b2)]

    def rows(p, count):
        g = torch.Generator(device=dev).manual_seed(3000 + p)
        a = torch.randint(0, L, (count,), generator=g, device=dev)
        return torch.nn.functional.normalize(centres[a] + 0.35 * torch.randn(count, d5, generator=g, device=dev), dim=1).contiguous()

    pieces = [(p, min(piece, n - p * piece)) for p in range((n + piece - 1) // piece)]
    gq = torch.Generator(device=dev).manual_seed(78)
    Q = torch.nn.functional.normalize(centres[torch.randint(0, L, (nq,), generator=gq, device=dev)]
                                      + 0.35 * torch.randn(nq, d5, generator=gq, device=dev), dim=1).contiguous()
    Qh = Q.cpu().numpy()
    sel = np.sort(np.random.RandomState(8).choice(nq, 256, replace=False))
    out = []
    labels = None
    for pf in (True, False):
        idx = _capi.Index(0, prefilter=pf)
        idx.set_stream(torch.cuda.current_stream().cuda_stream)
        idx.set_mlp(layers)
        if labels is None:   # placement: argmax MLP(x) over all N with the HIP MLP kernels
            lab = torch.empty(n, dtype=torch.int32, device=dev)
            for p, c in pieces:
                idx.mlp_topk_device(rows(p, c), 1, lab[p * piece: p * piece + c])
            torch.cuda.synchronize()
            labels = lab.cpu().numpy().astype(np.int64)
            del lab
            # ... and the oracle's placement of a sample agrees
            xs = rows(3, 4096)
            got = torch.empty(4096, dtype=torch.int32, device=dev)
            idx.mlp_topk_device(xs, 1, got)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(got.cpu().numpy(), oracle.rank_classes(oracle.forward_logits(layers, xs.cpu().numpy(), nthreads=16), 1)[:, 0])
        idx.buckets_begin(labels, d5, L)
        for p, c in pieces:
            idx.add_rows(rows(p, c), p * piece)
            torch.cuda.synchronize()
        idx.buckets_end()
        d = torch.empty((nq, 10), dtype=torch.float32, device=dev)
        i = torch.empty((nq, 10), dtype=torch.int32, device=dev)
        bo = torch.empty((nq, NB), dtype=torch.int32, device=dev)
        idx.search_device(Q, Q, NB, 10, d, i, None, bo)
        torch.cuda.synchronize()
        dh, ih, order_h = d.cpu().numpy(), i.cpu().numpy().view(np.uint32), bo.cpu().numpy()
        if pf:
            sizes = idx.bucket_sizes()
            print(f"C5: bucket sizes min/median/max {sizes.min()}/{int(np.median(sizes))}/{sizes.max()}, empty {int((sizes == 0).sum())}")
            np.testing.assert_array_equal(order_h[sel], oracle.rank_classes(oracle.forward_logits(layers, Qh[sel], nthreads=16), NB))
            active, survivors, fallbacks = idx.prefilter_stats()
            assert active and survivors >= 10 * nq
            print(f"C5: {survivors / (nq * NB):.2f} survivors per slot, {fallbacks} fallback slots")
            _oracle_check(oracle, idx, Qh, order_h, sel, NB, dh, ih)
        out.append((dh, ih, order_h))
        idx.close()
        torch.cuda.empty_cache()
    (d1, i1, o1), (d0, i0, o0) = out
    np.testing.assert_array_equal(o1, o0)
    np.testing.assert_array_equal(i1, i0)
    np.testing.assert_array_equal(d1, d0)
    assert np.all(np.diff(d1, axis=1) >= 0) and np.all(i1 > 0)


def test_c3_eight_ranks_full_size():
    """C3 (configs[2]): the 10M x 768 / 120-leaf / top-4 index bucket-sharded over 8 ranks, every rank played on the one card in
    turn: owned-only ingest (lmi_buckets_add_owned_rows), the rank's block of the batch, then lmi_merge_gathered over the 8
    blocks must be byte-identical to the single-handle C2 answer (SURVEY 8e identity requirement; no reference counterpart)."""
    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.sharded import assign_buckets

    _need_hbm(150)
    L, NB, n, nq, piece, world = 120, 4, 10_000_000, 10_000, 1 << 19, 8
    dev = torch.device("cuda", 0)
    g0 = torch.Generator(device=dev).manual_seed(2023)
    centres = torch.randn(L, D, generator=g0, device=dev)

    def rows(p, count):
        g = torch.Generator(device=dev).manual_seed(1000 + p)
        a = torch.randint(0, L, (count,), generator=g, device=dev)
        return a, torch.nn.functional.normalize(centres[a] + torch.randn(count, D, generator=g, device=dev), dim=1).contiguous()

    pieces = [(p, min(piece, n - p * piece)) for p in range((n + piece - 1) // piece)]
    labels_t = torch.cat([rows(p, c)[0] for p, c in pieces])
    labels = labels_t.cpu().numpy().astype(np.int64)
    sizes = np.bincount(labels, minlength=L)
    gq = torch.Generator(device=dev).manual_seed(77)
    Q = torch.nn.functional.normalize(centres[torch.randint(0, L, (nq,), generator=gq, device=dev)]
                                      + torch.randn(nq, D, generator=gq, device=dev), dim=1).contiguous()
    order = (Q @ centres.T).topk(NB, dim=1).indices.to(torch.int32).contiguous()
    stream = torch.cuda.current_stream().cuda_stream

    single = _capi.Index(0)
    single.set_stream(stream)
    single.buckets_begin(labels, D, L)
    for p, c in pieces:
        single.add_rows(rows(p, c)[1], p * piece)
        torch.cuda.synchronize()
    single.buckets_end()
    sd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    si = torch.empty((nq, 10), dtype=torch.int32, device=dev)
    single.scan_topk_device(Q, order, NB, 10, sd, si)
    torch.cuda.synchronize()
    single.close()
    torch.cuda.empty_cache()

    owner = assign_buckets(sizes, world)
    blocks = torch.empty((world, 3, nq, 10), dtype=torch.int32, device=dev)
    rows_seen = 0
    for r in range(world):
        own = (owner == r)
        h = _capi.Index(0)
        h.set_stream(stream)
        h.buckets_begin(labels, D, L, owned=own.astype(np.uint8))
        own_mask = torch.from_numpy(own).to(dev)
        for p, c in pieces:
            a, x = rows(p, c)
            keep = torch.nonzero(own_mask[a]).flatten()
            if keep.numel():
                h.add_owned_rows(x[keep].contiguous(), (keep + p * piece).contiguous())
            torch.cuda.synchronize()
        h.buckets_end()
        assert h.bucket_sizes().sum() == sizes[own].sum()
        rows_seen += int(sizes[own].sum())
        h.scan_topk_device(Q, order, NB, 10, blocks[r, 0], blocks[r, 1], blocks[r, 2])
        torch.cuda.synchronize()
        h.close()
        torch.cuda.empty_cache()
    assert rows_seen == n
    merger = _capi.Index(0)
    merger.set_stream(stream)
    md = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    mi = torch.empty((nq, 10), dtype=torch.int32, device=dev)
    merger.merge_gathered(blocks[0, 0], blocks[0, 1], blocks[0, 2], world, nq, 10, md, mi, world_stride=3 * nq * 10)
    torch.cuda.synchronize()
    assert torch.equal(mi, si) and torch.equal(md.view(torch.int32), sd.view(torch.int32))
    merger.close()
