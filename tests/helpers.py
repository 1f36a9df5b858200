"""Shared helpers for the parity tests."""
import os

import numpy as np

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))
    return g


def layers_from(g, prefix=""):
    n = int(g[f"{prefix}n_layers"])
    return [(g[f"{prefix}W{i}"], g[f"{prefix}b{i}"]) for i in range(n)]


def inputs_for(name, g):
    """Regenerates the fixture's inputs from its seed and verifies the stored checksums.
    Returns (X_nav, Q_nav, X_search, Q_search)."""
    X, Q = synth.mixture(int(g["seed"]), int(g["N"]), int(g["d"]), int(g["n_centres"]), int(g["nq"]))
    if name == "G4":
        X = X.copy()
        X[g["dup_rows"]] = X[int(g["dup_src"])]
    np.testing.assert_array_equal(synth.checksum(X), g["x_checksum"])
    np.testing.assert_array_equal(synth.checksum(Q), g["q_checksum"])
    if "P" in g:
        P = g["P"]
        return (X @ P).astype(np.float32), (Q @ P).astype(np.float32), X, Q
    return X, Q, X, Q


def compare_modulo_near_ties(ref_d, ref_i, got_d, got_i, tol=2e-6, rtol=1e-4):
    """ids must agree except inside groups of reference distances closer than `tol` (two fp32
    summation orders may order such near-ties differently); distances within `rtol` relative
    (north_star: 1e-4).  Returns the number of positions whose id differs."""
    ref_d = np.asarray(ref_d, dtype=np.float64)
    got_d = np.asarray(got_d, dtype=np.float64)
    assert ref_d.shape == got_d.shape == ref_i.shape == got_i.shape
    fin = np.isfinite(ref_d)
    assert np.array_equal(fin, np.isfinite(got_d))
    big = fin & (np.abs(ref_d) > 1e30)  # faiss -FLT_MAX padding
    np.testing.assert_array_equal(ref_d[big], got_d[big])
    ok = fin & ~big
    np.testing.assert_allclose(got_d[ok], ref_d[ok], rtol=rtol, atol=tol)
    diff = 0
    for r in range(ref_d.shape[0]):
        for j in range(ref_d.shape[1]):
            if ref_i[r, j] == got_i[r, j]:
                continue
            diff += 1
            assert ok[r, j], f"row {r} pos {j}: id differs at a padded/unvisited slot"
            near = np.abs(ref_d[r] - ref_d[r, j]) <= tol
            last = abs(ref_d[r, j] - ref_d[r, -1]) <= tol  # k-th/(k+1)-th boundary flip
            assert (got_i[r, j] in ref_i[r][near]) or last, (
                f"row {r} pos {j}: ref id {ref_i[r, j]} (d={ref_d[r, j]!r}) vs got {got_i[r, j]} "
                f"(d={got_d[r, j]!r}) is not a near-tie")
    return diff
