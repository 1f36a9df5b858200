#!/usr/bin/env python3
"""How many candidates would pass 2 emit if pass 1 were FOLDED into it?  (CPU simulation, numpy; no GPU needed.)

Today (lmi_pass2.h): pass 1 scores every 16th 256-row tile of a bucket, the column's bound is the 10th largest of the sampled
tiles' 16 slot maxima each, pass 2 emits every row whose score reaches it.  Folded (the alternative): no pass 1; a pass-2 item (a
2048-row chunk = 8 tiles of one bucket and query tile) starts from whatever bound earlier items of its bucket have PUBLISHED for the
column, tightens it with the slot maxima of its own tiles as it goes, and publishes it when it ends.  Items of a bucket are taken
from the queue in order by the ~32 blocks of an XCD, so about `concurrent` of them run at the same time and see no bound of each
other.  The simulation draws one bucket of the bench's generator (unit-norm rows of one Gaussian cluster, sigma 1, d = 768, exact f32
scores; the fp16 error term is left out of both), `m` queries of the same cluster (their primary bucket) and counts, per column,
the rows each scheme emits.

  python3 tools/fold_sim.py [--rows 83328 --m 128 --d 768 --concurrent 32]"""
import argparse

import numpy as np


def kth_of_slot_maxima(slot_max, k=10):
    """slot_max [slots, m] -> per column the k-th largest slot maximum (-inf with fewer than k slots)."""
    if slot_max.shape[0] < k:
        return np.full(slot_max.shape[1], -np.inf, dtype=np.float32)
    return np.partition(slot_max, slot_max.shape[0] - k, axis=0)[slot_max.shape[0] - k]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=83_328)
    ap.add_argument("--m", type=int, default=128)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--concurrent", type=int, default=32)
    ap.add_argument("--seed", type=int, default=2023)
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    c = rs.randn(args.d).astype(np.float32)
    X = c + rs.randn(args.rows, args.d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = c + rs.randn(args.m, args.d).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    S = X @ Q.T                                            # [rows, m]
    n = args.rows // 256 * 256
    S = S[:n]
    slots = S.reshape(n // 16, 16, args.m).max(axis=1)     # 16-row slot maxima, [n/16, m]; a 256-row tile = 16 consecutive slots
    ntiles = n // 256
    top10 = np.partition(S, n - 10, axis=0)[n - 10]        # the column's true 10th best
    # --- today: every 16th tile sampled
    samp = np.concatenate([slots[t * 16:(t + 1) * 16] for t in range(0, ntiles, 16)])
    b_now = kth_of_slot_maxima(samp)
    emit_now = (S >= b_now).sum(axis=0)
    # --- folded: chunks of 8 tiles, `concurrent` of them in flight
    chunk_tiles = 8
    nchunks = (ntiles + chunk_tiles - 1) // chunk_tiles
    published = np.full(args.m, -np.inf, dtype=np.float32)
    emit_fold = np.zeros(args.m, dtype=np.int64)
    pending = []                                           # (end wave, bound) of chunks in flight
    for c0 in range(0, nchunks, args.concurrent):
        start_bound = published.copy()                     # what a wave of concurrent chunks sees
        ends = []
        for ch in range(c0, min(nchunks, c0 + args.concurrent)):
            seen = np.empty((0, args.m), dtype=np.float32)
            thr = start_bound.copy()
            for t in range(ch * chunk_tiles, min(ntiles, (ch + 1) * chunk_tiles)):
                tile = S[t * 256:(t + 1) * 256]
                # the tile is scored, its slot maxima join the chunk's, THEN its rows are tested (the best the fold can do)
                seen = np.concatenate([seen, slots[t * 16:(t + 1) * 16]])
                thr = np.maximum(thr, kth_of_slot_maxima(seen))
                emit_fold += (tile >= thr).sum(axis=0)
            ends.append(thr)
        for thr in ends:
            published = np.maximum(published, thr)
    print(f"bucket of {n} rows, {args.m} primary columns, d = {args.d}; true 10th best per column: mean {top10.mean():.4f}")
    print(f"  today  (1/16 of the tiles sampled first):        {emit_now.mean():8.1f} candidates per column (max {emit_now.max()})")
    print(f"  folded ({args.concurrent:2d} chunks of 8 tiles in flight at a time): {emit_fold.mean():8.1f} candidates per column (max {emit_fold.max()})")


if __name__ == "__main__":
    main()
