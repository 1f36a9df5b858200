#!/usr/bin/env python3
"""Developer aid (GPU box): phase timing of the streamed pass-2 kernel from a -DLMI_PF_STAMPS build.

  python tools/pf_stamps.py learnedmetricindex_amd/variants/pf_stamps.so

Per wave: share of its time in wait (vmcnt: the stage has landed), bar (barrier), stage (fragment reads + MFMAs + DMA
issue), epi (epilogue of a 256 x 256 tile), start / end of an item; cycles per tile."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import qr_stamps  # noqa: E402  (shares the synthetic index)


def report(capi, idx, t):
    print("pass 2:", float(t[capi.T_PF_EMIT]), "ms; pass 1:", float(t[capi.T_PF_SAMPLE]), "ms (a -DLMI_PF_STAMPS_SAMPLE=1 build stamps pass 1)")
    raw = idx.debug_peek("pf_stamps", 8 * 12 * 8).view(np.uint64).reshape(8, 12).astype(np.float64)
    names = ["wait", "bar", "stage", "epi", "start", "end", "flush", "thrpass", "atomics", "merge"]
    tiles = raw[:, 7:8].copy()
    raw = np.concatenate([raw[:, :6], raw[:, 8:11], raw[:, 6:7], tiles], axis=1)
    print(f"-- {raw[0, 10]:.0f} tiles per wave, {raw[:, :10].sum(axis=1).mean() / max(1.0, raw[0, 10]):.0f} cycles per tile and wave")
    print("   wave " + " ".join(f"{n:>7s}" for n in names) + "   cycles/tile: " + " ".join(f"{n:>7s}" for n in names))
    for wv in range(8):
        r = raw[wv, :10]
        print(f"   {wv:4d} " + " ".join(f"{100 * v / max(1.0, r.sum()):6.1f}%" for v in r) + "                " + " ".join(f"{v / max(1.0, raw[wv, 10]):7.0f}" for v in r))


if __name__ == "__main__":
    qr_stamps.main(report)
