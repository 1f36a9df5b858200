#!/usr/bin/env python3
"""A/B timing of liblmi_hip.so build variants in ONE process (interleaved rounds, guide rule 24).

  python tools/scan_ab.py learnedmetricindex_amd/liblmi_hip.so learnedmetricindex_amd/variants/x.so ...
      [--n 10000000 --d 768 --leaves 120 --nq 10000 --nb 4 --rounds 8 --chunk-rows 2048]

Synthetic routing (no MLP training): unit-norm gaussian rows, labels drawn from a skewed
distribution close to a trained LMI's bucket sizes, each query visits nb distinct random buckets
with probability proportional to bucket size.  Every variant gets its own index and must return
identical ids; prints per-variant scan-kernel time (median/min) and TFLOP/s.
"""
import argparse
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_capi(path, tag):
    spec = importlib.util.spec_from_file_location(f"_capi_{tag}", os.path.join(ROOT, "learnedmetricindex_amd", "_capi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.LIB_PATH = os.path.abspath(path)
    mod.lib()
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--leaves", type=int, default=120)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--nb", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--chunk-rows", type=int, nargs="*", default=[])
    ap.add_argument("--no-check", action="store_true", help="ablation builds: do not compare ids")
    args = ap.parse_args()
    import torch

    dev = torch.device("cuda", 0)
    N, d, L, nq, nb = args.n, args.d, args.leaves, args.nq, args.nb
    rs = np.random.RandomState(1)
    w = rs.gamma(4.0, 1.0, size=L)
    w /= w.sum()
    labels = rs.choice(L, size=N, p=w).astype(np.int64)
    sizes = np.bincount(labels, minlength=L)
    order = np.stack([rs.choice(L, size=nb, replace=False, p=w) for _ in range(nq)]).astype(np.int32)
    g = torch.Generator(device=dev).manual_seed(5)
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g, device=dev), dim=1).contiguous()
    bo = torch.from_numpy(order).to(dev)
    CH = 1 << 19
    variants = []
    for i, path in enumerate(args.libs):
        capi = load_capi(path, i)
        cr = args.chunk_rows[i] if i < len(args.chunk_rows) else None
        idx = capi.Index(0, chunk_rows=cr)
        idx.set_stream(torch.cuda.current_stream().cuda_stream)
        idx.buckets_begin(labels, d, L)
        for p in range((N + CH - 1) // CH):
            n = min(CH, N - p * CH)
            gg = torch.Generator(device=dev).manual_seed(100 + p)
            x = torch.nn.functional.normalize(torch.randn(n, d, generator=gg, device=dev), dim=1).contiguous()
            idx.add_rows(x, p * CH)
            torch.cuda.synchronize()
        idx.buckets_end()
        out_d = torch.empty((nq, 10), dtype=torch.float32, device=dev)
        out_i = torch.empty((nq, 10), dtype=torch.int32, device=dev)
        variants.append((path, capi, idx, out_d, out_i, []))
    ref = None
    emit = {}
    for r in range(args.rounds + 1):
        for path, capi, idx, out_d, out_i, times in variants:
            idx.scan_topk_device(q, bo, nb, 10, out_d, out_i)
            t = idx.timings()
            if r:
                times.append(float(t[capi.T_SCAN]))
                emit.setdefault(path, []).append(float(t[capi.T_PF_EMIT]))
                emit.setdefault(path + "#rs", []).append(float(t[capi.T_RESCORE]))
                emit.setdefault(path + "#p1", []).append(float(t[capi.T_PF_SAMPLE]))
            elif ref is None:
                ref = out_i.clone()
            else:
                assert args.no_check or torch.equal(ref, out_i), f"{path}: ids differ from {args.libs[0]}"
    for path, capi, idx, out_d, out_i, times in variants:
        fl, pairs, items = idx.scan_stats()
        med, mn = float(np.median(times)), float(np.min(times))
        e = emit.get(path, [0.0])
        rs_ = emit.get(path + "#rs", [0.0])
        p1_ = float(np.median(emit.get(path + "#p1", [0.0])))
        print(f"{os.path.basename(path):28s} pass2 median {float(np.median(e)):7.3f} min {float(np.min(e)):7.3f} ms | rescore median {float(np.median(rs_)):6.3f} | pass1 median {p1_:6.3f} | scan median {med:8.3f} ms  min {mn:8.3f} ms  "
              f"{fl / med / 1e9:7.2f} TFLOP/s (median)  items {items}  prefilter {idx.prefilter_stats()}  "
              f"rounds {[round(t, 1) for t in times]}  last phases {[round(float(v), 2) for v in idx.timings()]}", flush=True)


if __name__ == "__main__":
    main()
