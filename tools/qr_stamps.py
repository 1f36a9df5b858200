#!/usr/bin/env python3
"""Developer aid (GPU box): phase timing of pass2_qr_kernel from a -DLMI_QR_STAMPS build of the library.

  python tools/qr_stamps.py learnedmetricindex_amd/variants/qr_stamps.so [--n 10000000 --nq 10000]

Prints, per mode (full tiles / split tiles) and wave, the share of the wave's time in each phase of the half-step loop
(wait = vmcnt wait for the row-block's pieces, bar = barrier, epi = epilogue, comp = MFMA half, idle = half-steps without
MFMAs incl. their DMA issue, start = item prologue incl. the resident fragment load, end = drain)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from scan_ab import load_capi  # noqa: E402


def main(report=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--leaves", type=int, default=120)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--nb", type=int, default=4)
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    N, d, L, nq, nb = args.n, args.d, args.leaves, args.nq, args.nb
    rs = np.random.RandomState(1)
    w = rs.gamma(4.0, 1.0, size=L); w /= w.sum()
    labels = rs.choice(L, size=N, p=w).astype(np.int64)
    order = np.stack([rs.choice(L, size=nb, replace=False, p=w) for _ in range(nq)]).astype(np.int32)
    g = torch.Generator(device=dev).manual_seed(5)
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g, device=dev), dim=1).contiguous()
    bo = torch.from_numpy(order).to(dev)
    capi = load_capi(args.lib, 0)
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
    out_d = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, 10), dtype=torch.int32, device=dev)
    for _ in range(3):
        idx.scan_topk_device(q, bo, nb, 10, out_d, out_i)
    t = idx.timings()
    if report is not None:
        return report(capi, idx, t)
    print("pass 2:", float(t[capi.T_PF_EMIT]), "ms")
    raw = idx.debug_peek("pf_stamps", 2 * 8 * 12 * 8).view(np.uint64).reshape(2, 8, 12).astype(np.float64)
    # epi = rest of the epilogue after its two inner stamps (atomic issue); flush = waiting for / storing the previous
    # epilogue's candidates; cmpct = threshold test + compaction
    names = ["wait", "bar", "epi", "comp", "idle", "start", "end", "flush", "cmpct"]
    raw = np.concatenate([raw[:, :, :7], raw[:, :, 8:10], raw[:, :, 7:8]], axis=2)  # .. , flush, cmpct, nrb
    for m, mode in enumerate(["full", "split"]):
        tot = raw[m, :, :9].sum()
        if tot == 0:
            continue
        print(f"-- {mode} tiles: {raw[m, 0, 9]:.0f} row-blocks per wave, {raw[m, :, :9].sum(axis=1).mean() / max(1.0, raw[m, 0, 9]):.0f} cycles per row-block and wave")
        print("   wave " + " ".join(f"{n:>7s}" for n in names))
        for wv in range(8):
            r = raw[m, wv, :9]
            print(f"   {wv:4d} " + " ".join(f"{100 * v / max(1.0, r.sum()):6.1f}%" for v in r))


if __name__ == "__main__":
    main()
