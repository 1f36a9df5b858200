#!/usr/bin/env python3
"""Times the device-side multi-level navigation (lmi_nav_order) on a random [c0, c1] tree (developer aid).
  python tools/nav_bench.py --d 32 --cats 10 10 --nq 10000 --nb 10"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=32)
    ap.add_argument("--cats", type=int, nargs="+", default=[10, 10])
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--nb", type=int, default=10)
    ap.add_argument("--check", type=int, default=200, help="queries compared with the oracle's walk")
    a = ap.parse_args()
    from learnedmetricindex_amd import _capi
    from oracle import lmi_oracle

    rs = np.random.RandomState(0)

    def mlp(out):
        return [((rs.randn(a.hidden, a.d) * 2 / np.sqrt(a.d)).astype(np.float32), (0.1 * rs.randn(a.hidden)).astype(np.float32)),
                ((rs.randn(out, a.hidden) * 2 / np.sqrt(a.hidden)).astype(np.float32), (0.1 * rs.randn(out)).astype(np.float32))]

    c0, c1 = a.cats
    root = mlp(c0)
    internal = [((i, -1), mlp(c1)) for i in range(c0)]
    bucket_paths = [(i, j) for i in range(c0) for j in range(c1)]
    eng = _capi.Index(0)
    eng.set_mlp(root)
    for i, (_, l) in enumerate(internal):
        eng.nav_set_model(i + 1, l)
    offset, cm, cb = [0], [], []
    for i in range(c0):
        cm.append(i + 1); cb.append(-2)
    offset.append(len(cm))
    for i in range(c0):
        for j in range(c1):
            cm.append(-1); cb.append(i * c1 + j)
        offset.append(len(cm))
    eng.nav_set_tree(offset, cm, cb)
    q = rs.randn(a.nq, a.d).astype(np.float32)
    eng.nav_order(q[:64], a.nb)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        slab, ent = eng.nav_order(q, a.nb)
        best = min(best, time.perf_counter() - t0)
    print(f"d={a.d} tree={a.cats} nq={a.nq} nb={a.nb}: device {float(eng.timings()[_capi.T_INFERENCE]):.3f} ms, "
          f"host call (pageable in/out) {best * 1e3:.3f} ms")
    if a.check:
        bo = lmi_oracle.precompute_bucket_order_multilevel(root, internal, bucket_paths, q[:a.check], a.nb, a.cats, nthreads=8)
        got = slab[:a.check]
        exp = bo[:, :, 0] * c1 + bo[:, :, 1]
        print("identical to the oracle's walk:", bool(np.array_equal(got, exp)), f"({(got == exp).mean():.4f} of the slots)")


if __name__ == "__main__":
    main()
