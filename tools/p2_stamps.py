#!/usr/bin/env python3
"""Developer aid (GPU box): phase timing of pass2_kernel from a -DLMI_P2_STAMPS build (lmi_pass2.h).

  python tools/p2_stamps.py learnedmetricindex_amd/variants/p2_stamps.so [--n 10000000 --mq 345]

Index: 120 equal buckets of unit-norm gaussian rows; every bucket receives exactly --mq queries (the benchmark's
mean is 345 = 11 col-blocks).  Per wave: share of its time waiting for a stage to land (vmcnt), at the stage barrier, in the stage
(fragment reads + MFMAs + DMA issue), in the epilogue, at item start / end; cycles per tile."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from scan_ab import load_capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--leaves", type=int, default=120)
    ap.add_argument("--mq", type=int, default=345)
    ap.add_argument("--nb", type=int, default=4)
    ap.add_argument("--clustered", action="store_true",
                    help="L gaussian clusters (unit-norm rows, sigma 1), queries drawn the same way, routed to their nb nearest "
                         "centres: the benchmark's kind of data (the query-level bound then prunes ranks 1..nb-1)")
    args = ap.parse_args()
    import torch

    capi = load_capi(args.lib, 0)
    dev = torch.device("cuda", 0)
    N, d, L, nb = args.n, args.d, args.leaves, args.nb
    nq = args.mq * L // nb
    labels = (np.arange(N) % L).astype(np.int64)
    rs = np.random.RandomState(3)
    order = np.stack([(rs.randint(L) + np.arange(nb) * (L // nb)) % L for _ in range(nq)]).astype(np.int32)
    # exactly mq per bucket: query i, rank r -> bucket (i + r * L/nb) % L
    order = np.stack([(np.arange(nq) + r * (L // nb)) % L for r in range(nb)], axis=1).astype(np.int32)
    idx = capi.Index(0)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.buckets_begin(labels, d, L)
    CH = 1 << 19
    gc = torch.Generator(device=dev).manual_seed(9)
    centres = torch.randn(L, d, generator=gc, device=dev)
    for p in range((N + CH - 1) // CH):
        n = min(CH, N - p * CH)
        g = torch.Generator(device=dev).manual_seed(100 + p)
        x = torch.randn(n, d, generator=g, device=dev)
        if args.clustered:
            x = x + centres[torch.from_numpy(labels[p * CH: p * CH + n]).to(dev)]
        idx.add_rows(torch.nn.functional.normalize(x, dim=1).contiguous(), p * CH)
        torch.cuda.synchronize()
    idx.buckets_end()
    g = torch.Generator(device=dev).manual_seed(5)
    q = torch.randn(nq, d, generator=g, device=dev)
    if args.clustered:   # query i belongs to cluster order[i, 0]; its other ranks are the deterministic ones above (far clusters)
        q = q + centres[torch.from_numpy(order[:, 0].astype(np.int64)).to(dev)]
    q = torch.nn.functional.normalize(q, dim=1).contiguous()
    bo = torch.from_numpy(order).to(dev)
    out_d = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, 10), dtype=torch.int32, device=dev)
    for _ in range(3):
        idx.scan_topk_device(q, bo, nb, 10, out_d, out_i)
        t = idx.timings()
    print("pass 2:", float(t[capi.T_PF_EMIT]), "ms; pass 1:", float(t[capi.T_PF_SAMPLE]), "ms; rescore:", float(t[capi.T_RESCORE]),
          "ms; prefilter stats", idx.prefilter_stats(), "scan stats", idx.scan_stats())
    raw = idx.debug_peek("pf_stamps", 8 * 12 * 8).view(np.uint64).reshape(8, 12).astype(np.float64)
    names = ["wait", "bar", "stage", "epi", "start", "end"]
    if args.d <= 128:   # the low-dimensional kernel's phases (lmi_pass2_small.h, PS_STAMP)
        names = ["start", "ldwait", "blocks", "drain", "end", "-"]
    tiles = raw[:, 7]
    print(f"-- {tiles[0]:.0f} tiles per wave (all CUs), {raw[:, :6].sum(axis=1).mean() / max(1.0, tiles[0]):.0f} cycles per tile and wave")
    print("   wave " + " ".join(f"{n:>7s}" for n in names) + "   cycles/tile: " + " ".join(f"{n:>7s}" for n in names))
    for wv in range(8):
        r = raw[wv, :6]
        print(f"   {wv:4d} " + " ".join(f"{100 * v / max(1.0, r.sum()):6.1f}%" for v in r) + "                " + " ".join(f"{v / max(1.0, tiles[wv]):7.0f}" for v in r))


if __name__ == "__main__":
    main()
