#!/usr/bin/env python3
"""GPU box: where front_kernel (lmi_front.h) spends its time.  The kernel's cost depends on the batch (nq, n_buckets, L, d), not on the
index size, so a small index is enough.  LMI_FR_DEBUG=1 makes block 0, the first bucket block and the last one stamp the chip's 100 MHz
clock at their phase boundaries; this prints the phases (us) and the launch's duration from the handle's device stamps.

  python3 tools/front_phases.py [d L nq nb]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(d=768, L=120, nq=10000, nb=4):
    os.environ["LMI_FR_DEBUG"] = "1"
    os.environ.setdefault("LMI_FRONT", "1")   # (LMI_FRONT=0: the separate preparation kernels, for the route phase's baseline)
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
    print(f"d={d} L={L} nq={nq} nb={nb}: route phase {tm[_capi.T_ROUTE] * 1e3:.1f} us (mean of {n})")
    t = st[0:5]
    if t[0]:
        print(f"  pack_kernel block 0 (queues): statistics {(t[1] - t[0]) / 100:.1f}, keys {(t[2] - t[1]) / 100:.1f}, sort {(t[3] - t[2]) / 100:.1f}, "
              f"queues {(t[4] - t[3]) / 100:.1f} us")
    for base, who in ((8, "route_kernel first block"), (16, "route_kernel last block ")):
        t = st[base:base + 3]
        if t[0] and t[2]:
            print(f"  {who}: counting walk {(t[1] - t[0]) / 100:.1f}, allocation + placing walk {(t[2] - t[1]) / 100:.1f} us"
                  + (f"; pack block 0 starts {(st[0] - t[0]) / 100:.1f} us after it" if base == 8 and st[0] else ""))
    idx.close()


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
