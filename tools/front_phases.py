#!/usr/bin/env python3
"""GPU box: where front_kernel (lmi_front.h) spends its time.  The kernel's cost depends on the batch (nq, n_buckets, L, d), not on the
index size, so a small index is enough.  LMI_FR_DEBUG=1 makes block 0, the first bucket block and the last one stamp the chip's 100 MHz
clock at their phase boundaries; this prints the phases (us) and the launch's duration from the handle's device stamps.

  python3 tools/front_phases.py [d L nq nb [parts]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(d=768, L=120, nq=10000, nb=4, parts=0):
    os.environ["LMI_FR_DEBUG"] = "1"
    if parts:
        os.environ["LMI_FR_PARTS"] = str(parts)
    from learnedmetricindex_amd import _capi

    rs = np.random.RandomState(1)
    N = 200 * L
    X = rs.randn(N, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    lab = (np.arange(N) % L).astype(np.int64)
    Q = rs.randn(nq, d).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    order = np.stack([rs.permutation(L)[:nb] for _ in range(nq)]).astype(np.int32)
    idx = _capi.Index(0, chunk_rows=256)
    idx.set_buckets(X, lab, L)
    for _ in range(3):
        idx.scan_topk(Q, order, 10)
    idx.timings_reset()
    for _ in range(5):
        idx.scan_topk(Q, order, 10)
    tm, n = idx.timings_mean()
    st = idx.debug_peek("fr_dbg", 24 * 8).view(np.uint64).astype(np.int64)
    names0 = ["walk A", "global copies", "fills", "queue sort"]
    namesb = ["walk A", "prefix", "walk B", "first col-block", "rest"]
    print(f"d={d} L={L} nq={nq} nb={nb} parts={parts or 'auto'}: route phase {tm[_capi.T_ROUTE] * 1e3:.1f} us (mean of {n})")
    t = st[0:5]
    print("  block 0      :", ", ".join(f"{nm} {(t[i + 1] - t[i]) / 100:.1f}" for i, nm in enumerate(names0)), f"| total {(t[4] - t[0]) / 100:.1f} us")
    for base, who in ((8, "first bucket"), (16, "last bucket ")):
        t = st[base:base + 6]
        if t[0] == 0:
            continue
        print(f"  {who} :", ", ".join(f"{nm} {(t[i + 1] - t[i]) / 100:.1f}" for i, nm in enumerate(namesb)), f"| total {(t[5] - t[0]) / 100:.1f} us; "
              f"starts {(t[0] - st[0]) / 100:.1f} us after block 0")
    idx.close()


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
