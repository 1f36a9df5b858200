#!/usr/bin/env python3
"""Developer aid (GPU box): random small shapes, prefilter mode against the all-f32 mode (both on the GPU), bit for bit.

  python tools/fuzz_modes.py [--cases 200 --seed 1]

Every case draws d (1..2100, weighted towards the kernels' boundaries), the number of buckets, their sizes (empty, tiny, ragged,
one heavy), top-n, k and a batch routed at random (so that buckets receive 0 .. thousands of queries), builds two indexes and
compares ids and distances of the whole batch.  The low-dimensional kernels' wide form is forced on / off / left automatic in turn
(LMI_PS_WIDE, read when a handle is created).  Prints the failing case's parameters and exits 1 on the first mismatch.
tests/test_gpu_fuzz.py runs the same cases with five queries of each also re-computed by the CPU oracle (LMI_FUZZ_CASES / LMI_FUZZ_SEED
lengthen it)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

D_CHOICES = [1, 3, 8, 15, 16, 17, 31, 32, 33, 45, 48, 63, 64, 65, 77, 80, 81, 95, 96, 97, 100, 111, 112, 113, 127, 128, 129, 130, 159, 160,
             161, 191, 192, 200, 255, 256, 257, 300, 384, 500, 512, 640, 767, 768, 769, 1000, 1024, 1025, 1100, 1127, 1300, 1536, 2048, 2100]


def one_case(capi, rs, case, oracle=None):
    d = int(rs.choice(D_CHOICES)) if rs.rand() < 0.8 else int(rs.randint(1, 700))
    L = int(rs.choice([1, 2, 3, 7, 16, 40, 120, 300, 1000, 2500]))
    nb = int(min(L, rs.choice([1, 2, 3, 4, 5, 8])))
    k = int(rs.choice([1, 5, 10, 10, 10, 15, 20])) if nb > 1 else int(rs.choice([1, 5, 10]))
    k = min(k, 10 * nb)
    budget = int(2e7 // max(d, 16))                      # rows: keep a case under ~80 MB of f32
    N = int(min(budget, rs.choice([200, 3000, 20000, 60000])))
    kind = rs.randint(4)
    if kind == 0:
        labels = rs.randint(0, L, N)
    elif kind == 1:                                        # one heavy bucket, many empty ones
        labels = np.where(rs.rand(N) < 0.6, rs.randint(0, L), rs.randint(0, max(1, L // 3), N))
    elif kind == 2:                                        # sizes around the row-block / tile boundaries
        sizes = rs.choice([0, 1, 9, 10, 31, 32, 33, 255, 256, 257, 511, 513, 2047, 2049], L)
        labels = np.repeat(np.arange(L), sizes)[:N]
        N = labels.size
        if N == 0:
            labels, N = np.zeros(50, dtype=np.int64), 50
    else:
        w = 1.0 / (1.0 + np.arange(L) / 3.0)
        labels = rs.choice(L, N, p=w / w.sum())
    labels = labels.astype(np.int64)
    rs.shuffle(labels)
    centres = rs.randn(L, d).astype(np.float32)
    X = centres[labels] * rs.choice([0.0, 0.3, 1.0]) + rs.randn(N, d).astype(np.float32)
    if rs.rand() < 0.3:                                    # clusters of near-copies
        src = rs.randint(0, N, max(1, N // 50))
        X[rs.randint(0, N, src.size * 8)] = np.repeat(X[src], 8, axis=0) + 1e-5 * rs.randn(src.size * 8, d).astype(np.float32)
    X /= np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-20)
    nq = int(rs.choice([1, 7, 64, 500, 3000]))
    hot = rs.rand() < 0.5                                   # half of the cases: most queries on a few buckets (several query tiles)
    order = np.empty((nq, nb), dtype=np.int32)
    for i in range(nq):
        pool = rs.permutation(min(L, 4))[:nb] if hot and L >= nb and min(L, 4) >= nb and rs.rand() < 0.8 else rs.permutation(L)[:nb]
        order[i] = pool
    if rs.rand() < 0.2:
        order[rs.rand(nq, nb) < 0.1] = -1                   # unvisited slots
    Q = centres[np.maximum(order[:, 0], 0)] * 0.5 + rs.randn(nq, d).astype(np.float32)
    Q /= np.maximum(np.linalg.norm(Q, axis=1, keepdims=True), 1e-20)
    chunk = rs.choice([None, 256, 512, 2048])
    wide = ("", "0", "1")[case % 3]
    metric = "l2" if rs.rand() < 0.25 else "ip"           # L2: vectors of any norm, one stored column more
    if metric == "l2":
        X *= rs.uniform(0.2, 3.0, (N, 1)).astype(np.float32)
        Q *= np.float32(rs.uniform(0.5, 2.0))
    res = []
    for pf in (True, False):
        if wide:
            os.environ["LMI_PS_WIDE"] = wide
        else:
            os.environ.pop("LMI_PS_WIDE", None)
        idx = capi.Index(0, chunk_rows=None if chunk is None else int(chunk), prefilter=pf, metric=metric)
        idx.set_buckets(X, labels, L)
        res.append(idx.scan_topk(Q, order, k))
        idx.close()
    desc = dict(case=case, d=d, L=L, nb=nb, k=k, N=N, kind=int(kind), nq=nq, hot=bool(hot), chunk=chunk, wide=wide, metric=metric)
    (d1, i1), (d0, i0) = res
    if oracle is not None:   # a few queries against the CPU oracle as well (the routing kernels serve both GPU modes)
        sub = np.sort(rs.choice(nq, min(nq, 5), replace=False))
        do, io, _ = oracle.search(None, Q[sub], X, Q[sub], labels, nb, k, bucket_order=order[sub][:, :, None], metric=metric)
        if not (np.array_equal(i1[sub].view(np.uint32), io) and np.array_equal(d1[sub].astype(np.float64), do)):
            print("ORACLE MISMATCH", desc, "queries", sub, flush=True)
            return False, desc
    if not (np.array_equal(i1, i0) and np.array_equal(d1.view(np.uint64) if d1.dtype == np.float64 else d1, d0.view(np.uint64) if d0.dtype == np.float64 else d0)):
        bad = np.flatnonzero((i1 != i0).any(axis=1) | (d1 != d0).any(axis=1))
        print("MISMATCH", desc, "queries", bad[:10], flush=True)
        return False, desc
    return True, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from learnedmetricindex_amd import _capi
    t0 = time.time()
    for case in range(a.cases):
        rs = np.random.RandomState(a.seed * 100003 + case)
        ok, desc = one_case(_capi, rs, case)   # (GPU modes only; tests/test_gpu_fuzz.py passes the CPU oracle in as well)
        if not ok:
            sys.exit(1)
        if case % 20 == 0:
            print(f"case {case}: ok {desc} ({time.time() - t0:.0f} s)", flush=True)
    print(f"{a.cases} cases identical in both modes ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
