import os, sys, numpy as np
sys.path[:0] = [os.getcwd()]
from learnedmetricindex_amd import _capi
rs = np.random.RandomState(0)
d = 768
for n in (5, 9, 12, 20, 33, 40):
    base = rs.randn(d).astype(np.float32)
    X = base[None, :] + 1e-3 * rs.randn(n, d).astype(np.float32)
    Q = rs.randn(8, d).astype(np.float32)
    lab = np.zeros(n, np.int64)
    order = np.zeros((8, 1), np.int32)
    out = {}
    for mode in ("1", "0"):
        os.environ["LMI_RESCORE_SIMPLE"] = mode
        idx = _capi.Index(0)
        idx.set_buckets(X, lab, 1)
        out[mode] = idx.scan_topk(Q, order, 10)
        st = idx.prefilter_stats()
        idx.close()
    d1, i1 = out["1"]; d0, i0 = out["0"]
    ex = 1 - (Q.astype(np.float64) @ X.astype(np.float64).T)
    print("n", n, "survivors/slot", st[1] / 8, "fallbacks", st[2], "equal", np.array_equal(d1, d0))
    if not np.array_equal(d1, d0):
        q = 0
        got = {int(i): float(v) for i, v in zip(i0[q], d0[q]) if i > 0}
        err = {i: round(v - ex[q, i - 1], 4) for i, v in got.items()}
        print("   streamed errors by id (row+1):", err)
