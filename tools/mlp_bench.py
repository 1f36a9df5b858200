#!/usr/bin/env python3
"""Times the navigation MLP (fused one-launch kernel vs per-layer kernels) for several batch sizes (developer aid)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from learnedmetricindex_amd import _capi  # noqa: E402

d, H, L, nb = 768, 512, 120, 4
rs = np.random.RandomState(0)
layers = [((rs.randn(H, d) / np.sqrt(d)).astype(np.float32), np.zeros(H, np.float32)),
          ((rs.randn(L, H) / np.sqrt(H)).astype(np.float32), np.zeros(L, np.float32))]
dev = torch.device("cuda", 0)
for fused in (1, 2, 0):
    idx = _capi.Index(0)
    idx.set_fused_mlp(fused)
    idx.set_mlp(layers)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    for nq in (2048, 8192, 10000, 16384, 32768):
        q = torch.randn(nq, d, device=dev)
        bo = torch.empty((nq, nb), dtype=torch.int32, device=dev)
        ts = []
        for _ in range(8):
            idx.mlp_topk_device(q, nb, bo)
            ts.append(float(idx.timings()[_capi.T_INFERENCE]))
        print(f"fused={fused} nq={nq:6d}: {np.median(ts[2:]) * 1e3:8.1f} us  ({2 * nq * (d * H + H * L) / np.median(ts[2:]) / 1e9:.1f} TFLOP/s)")
    idx.close()
