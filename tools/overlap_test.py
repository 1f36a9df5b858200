#!/usr/bin/env python3
"""Does a pinned H2D copy on a side stream overlap a long-running kernel on another stream? (developer aid)"""
import time
import torch

dev = torch.device("cuda", 0)
h = torch.empty((10000, 768), dtype=torch.float32).pin_memory()
d = torch.empty_like(h, device=dev)
a = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def copy_only():
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)


def mm_only():
    with torch.cuda.stream(s2):
        for _ in range(4):
            torch.mm(a, a)


def both():
    copy_only(); mm_only()


print(f"copy {t(copy_only):.3f} ms  mm {t(mm_only):.3f} ms  both {t(both):.3f} ms")
