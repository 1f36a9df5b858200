#!/usr/bin/env python3
"""Developer aid (GPU box): pass-2 time per col-block as a function of the queries per bucket (how well the query tiles are
filled), for one or more builds of the library (e.g. the default NG 2 build and a -DLMI_PF_NG=1 build).

  python tools/tile_fill.py lib1.so [lib2.so ..] [--n 10000000 --leaves 120]

Every bucket receives exactly m (query, rank) slots for m in --m; prints pass-2 ms and ns per (col-block x 1000 rows)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from scan_ab import load_capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--leaves", type=int, default=120)
    ap.add_argument("--m", type=int, nargs="*", default=[64, 96, 128, 160, 192, 256, 352, 512])
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    N, d, L = args.n, args.d, args.leaves
    rs = np.random.RandomState(1)
    labels = rs.randint(0, L, size=N).astype(np.int64)   # equal buckets
    g = torch.Generator(device=dev).manual_seed(5)
    for li, path in enumerate(args.libs):
        capi = load_capi(path, li)
        idx = capi.Index(0)
        idx.set_stream(torch.cuda.current_stream().cuda_stream)
        idx.buckets_begin(labels, d, L)
        CH = 1 << 19
        for p in range((N + CH - 1) // CH):
            n = min(CH, N - p * CH)
            gg = torch.Generator(device=dev).manual_seed(100 + p)
            idx.add_rows(torch.nn.functional.normalize(torch.randn(n, d, generator=gg, device=dev), dim=1).contiguous(), p * CH)
            torch.cuda.synchronize()
        idx.buckets_end()
        for m in args.m:
            nb = 4
            nq = m * L // nb
            # query i visits buckets (i*nb + r) % L: every bucket gets exactly m slots
            order = ((np.arange(nq)[:, None] * nb + np.arange(nb)[None, :]) % L).astype(np.int32)
            q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g, device=dev), dim=1).contiguous()
            bo = torch.from_numpy(order).to(dev)
            out_d = torch.empty((nq, 10), dtype=torch.float32, device=dev)
            out_i = torch.empty((nq, 10), dtype=torch.int32, device=dev)
            ts = []
            for r in range(6):
                idx.scan_topk_device(q, bo, nb, 10, out_d, out_i)
                t = idx.timings()
                if r:
                    ts.append(float(t[capi.T_PF_EMIT]))
            ms = float(np.median(ts))
            cbs = (m + 31) // 32
            print(f"{os.path.basename(path):20s} m {m:4d} ({cbs:2d} col-blocks) pass2 {ms:7.3f} ms  = {ms * 1e6 / (cbs * L * (N / L) / 1000):7.2f} ns per col-block x 1000 rows"
                  f"   ({2.0 * m * N * d / ms / 1e9:7.1f} TFLOP/s useful)", flush=True)
        idx.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
