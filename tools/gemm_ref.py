#!/usr/bin/env python3
"""Developer aid (GPU box): what the vendor GEMM (torch.matmul -> hipBLASLt, fp16 in / fp32 accumulate, fp16 out) reaches on the
shape of the prefilter's pass 2 -- per bucket (rows x 768) . (768 x queries) -- as a calibration point for its roofline
fraction.  The library kernel WRITES its scores (83 k x 352 halfs per bucket) and tests nothing; pass 2 tests every score
against a per-query threshold and emits the candidates instead.

  python tools/gemm_ref.py [--rows 83333 --queries 345 --d 768 --buckets 120]"""
import argparse
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=83333)
    ap.add_argument("--queries", type=int, default=345)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--buckets", type=int, default=120)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    X = [torch.randn(a.rows, a.d, generator=g, device=dev, dtype=torch.float16) for _ in range(a.buckets)]
    for nq in (a.queries, 352, 256, 512):
        Q = torch.randn(nq, a.d, generator=g, device=dev, dtype=torch.float16)
        out = torch.empty(a.rows, nq, device=dev, dtype=torch.float16)
        for x in X[:8]:
            torch.matmul(x, Q.T, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(a.reps):
            e0.record()
            for x in X:
                torch.matmul(x, Q.T, out=out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out2 = torch.empty(nq, a.rows, device=dev, dtype=torch.float16)
        for x in X[:8]:
            torch.matmul(Q, x.T, out=out2)
        torch.cuda.synchronize()
        best2 = 1e9
        for _ in range(a.reps):
            e0.record()
            for x in X:
                torch.matmul(Q, x.T, out=out2)
            e1.record()
            torch.cuda.synchronize()
            best2 = min(best2, e0.elapsed_time(e1))
        print(f"   (Q . X^T form: {best2:7.3f} ms)")
        best = min(best, best2)
        fl = 2.0 * a.rows * nq * a.d * a.buckets
        by = a.buckets * a.rows * a.d * 2.0
        print(f"queries/bucket {nq:4d}: {best:7.3f} ms for {a.buckets} buckets = {fl / best / 1e9:7.1f} TFLOP/s, "
              f"{by / best / 1e6:6.0f} GB/s of vectors = {by / best / 1e6 / 8000:.3f} of the 8 TB/s roofline")


if __name__ == "__main__":
    main()
