"""Developer aid (GPU box): every vector copied `dup` times (1e-7 apart) -- a slot's top ten and everything within 2 eps' of it are
>= dup rows.  Time of a 2 000-query batch and the slots that took the exact fallback (round 4: candidates past a column's buffer go
to the shared overflow log and are re-scored from there; "brute" = slots that still scanned their whole bucket)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learnedmetricindex_amd import _capi
rs = np.random.RandomState(0)
d, L, nb = 128, 16, 2
for dup in (1, 20, 60, 100, 300, 600, 1000):
    U = 200_000 // dup
    base = rs.randn(U, d).astype(np.float32); base /= np.linalg.norm(base, axis=1, keepdims=True)
    X = np.repeat(base, dup, axis=0)
    X = X + (1e-7 * rs.randn(*X.shape)).astype(np.float32) if dup > 1 else X
    labels = rs.randint(0, L, size=U).repeat(dup).astype(np.int64)
    Q = base[rs.choice(U, 2000)] + 0.05 * rs.randn(2000, d).astype(np.float32)
    Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    order = np.stack([rs.permutation(L)[:nb] for _ in range(2000)]).astype(np.int32)
    idx = _capi.Index(0)
    idx.set_buckets(X, labels, L)
    idx.scan_topk(Q, order, 10)
    t0 = time.time(); dd, ii = idx.scan_topk(Q, order, 10); t1 = time.time()
    act, sv, fb = idx.prefilter_stats()
    st = idx.debug_peek("pf_fallback", 32).view(np.uint32)
    print(f"dup x{dup}: {1e3*(t1-t0):8.1f} ms for 2000 queries x {nb} buckets, survivors/slot {sv/4000:.1f}, flagged slots {fb}, "
          f"brute-forced {st[5]}, overflow-log entries {st[3]} (fail flags {st[1]} {st[2]}), phases ms {np.round(idx.timings()[5:9], 3).tolist()}")
    idx.close()
