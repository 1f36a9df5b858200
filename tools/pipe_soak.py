#!/usr/bin/env python3
"""Developer aid (GPU box): soak of the host pipeline's stream logic -- many batches of DIFFERENT queries through
HostPipeline (navigation-stream overlap, side-stream MLP tail, side-stream work queues), every batch compared with a
synchronous lmi_search of the same queries.

  python tools/pipe_soak.py [--batches 300 --n 400000 --nq 9000]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=300)
    ap.add_argument("--n", type=int, default=400_000)
    ap.add_argument("--nq", type=int, default=9000)   # 282 MLP blocks: 256 + a thin tail -> the split path
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--leaves", type=int, default=64)
    ap.add_argument("--mode", default="nav", choices=["nav", "plain", "twin"])
    a = ap.parse_args()
    import torch
    from learnedmetricindex_amd import _capi
    from learnedmetricindex_amd.pipeline import HostPipeline
    rs = np.random.RandomState(3)
    d, L, nb, k = a.d, a.leaves, 4, 10
    layers = [((rs.randn(256, d) / np.sqrt(d)).astype(np.float32), rs.randn(256).astype(np.float32) * 0.1),
              ((rs.randn(L, 256) / 16).astype(np.float32), rs.randn(L).astype(np.float32) * 0.1)]
    X = rs.randn(a.n, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    idx = _capi.Index(0)
    idx.set_mlp(layers)
    labels = idx.mlp_topk(X, 1)[:, 0].astype(np.int64)
    idx.set_buckets(X, labels, L)
    nsets = 6
    Q = [rs.randn(a.nq, d).astype(np.float32) for _ in range(nsets)]
    for q in Q:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    idx.set_stream(0)
    want = [idx.search(q, q, nb, k) for q in Q]
    pipe = HostPipeline(idx, a.nq, d, d, nb, k, depth=2, same_queries=True, want_bucket_order=True,
                        overlap_inference=a.mode == "nav", two_handles=a.mode == "twin")
    order = rs.randint(0, nsets, size=a.batches)
    tickets = []
    bad = 0
    for bi, s in enumerate(order):
        tickets.append((pipe.submit(Q[s]), s))
        if len(tickets) >= 2:   # check the batch before the newest while the newest runs
            t, ss = tickets[-2]
            dd, ii = pipe.result(t)
            bo = pipe.bucket_order(t)
            ok = np.array_equal(ii, want[ss][1]) and np.array_equal(dd, want[ss][0]) and np.array_equal(bo, want[ss][2])
            bad += 0 if ok else 1
    pipe.drain()
    print(f"mode {a.mode}: {a.batches} batches, {bad} differing from the synchronous search")
    idx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
