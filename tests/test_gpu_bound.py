"""GPU (`-m gpu`): the fp16 prefilter's error bound, MEASURED on the scores the pass-2 kernel really computes.

lmi_prefilter.h keeps every row with shat >= That - 2 eps' and claims |shat - s'_canonical| <= eps' for every
(query, row), with eps' from measured rounding-error norms + `4 d 2^-24 ||q^|| ||x^||` for the two binary32
summations (the MFMA's internal accumulation order/rounding of v_mfma_f32_32x32x16_f16 is not documented).
A violated bound would silently drop a true neighbour, and mode-equality tests on friendly data cannot see it.
Here the test hook `lmi_debug_emit_all` makes pass 2 emit EVERY row of the visited buckets (<= 1024 rows each),
`lmi_debug_read_candidates` returns the kernel's shat per (query, row), and the oracle's canonical chain gives
s_c; s' = xscale*qscale*s_c is exact (powers of two).  Asserted: max |shat - s'| / eps' < 1, on

  * unit-norm Gaussian rows (the benchmark's distribution),
  * ALL-POSITIVE vectors (sum|terms| = |sum terms|: the worst case for accumulation error),
  * magnitudes that straddle binades inside one vector (partial sums cross many exponents, both growing
    and shrinking term order), and rows/queries spread over four orders of magnitude,
  * d in {768, 2048, 4096}.
The max ratio per case is printed (pytest -s) and returned in the assertion message."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

L, NB, NQ = 4, 4, 96
SIZES = (1000, 640, 997, 33)  # rows per bucket: full candidate buffers, ragged tiles, one below a tile


def make(kind, d, seed):
    rs = np.random.RandomState(seed)
    n = sum(SIZES)
    X = rs.randn(n, d).astype(np.float32)
    Q = rs.randn(NQ, d).astype(np.float32)
    if kind == "gauss":
        pass
    elif kind == "positive":
        X, Q = np.abs(X), np.abs(Q)
    elif kind == "binade_up":      # term magnitudes grow along k: every partial sum is dwarfed by the next terms
        ramp = np.exp2(np.linspace(-14, 0, d)).astype(np.float32)
        X, Q = np.abs(X) * ramp, np.abs(Q) * ramp
    elif kind == "binade_down":    # shrink along k: late terms fall below the running sum's ulp
        ramp = np.exp2(np.linspace(0, -14, d)).astype(np.float32)
        X, Q = np.abs(X) * ramp, np.abs(Q) * ramp
    elif kind == "binade_mix":     # per-element random exponents, random signs: cancellation across binades
        X = X * np.exp2(rs.randint(-12, 1, size=X.shape)).astype(np.float32)
        Q = Q * np.exp2(rs.randint(-12, 1, size=Q.shape)).astype(np.float32)
    elif kind == "scales":         # un-normalised rows and queries over four orders of magnitude
        X = X * (10.0 ** rs.uniform(-2, 2, size=(n, 1))).astype(np.float32)
        Q = Q * (10.0 ** rs.uniform(-2, 2, size=(NQ, 1))).astype(np.float32)
    else:
        raise ValueError(kind)
    if kind in ("gauss", "positive"):
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    labels = np.repeat(np.arange(L), SIZES).astype(np.int64)
    perm = rs.permutation(n)
    return np.ascontiguousarray(X[perm]), np.ascontiguousarray(Q), labels[perm]


CASES = [(k, d) for d in (768, 2048, 4096) for k in ("gauss", "positive", "binade_up", "binade_down", "binade_mix", "scales")]


@pytest.mark.parametrize("kind,d", CASES)
def test_prefilter_bound_holds(oracle, kind, d):
    from learnedmetricindex_amd import _capi

    X, Q, labels = make(kind, d, seed=1000 + d)
    order = np.tile(np.arange(L, dtype=np.int32), (NQ, 1))
    idx = _capi.Index(0, prefilter=True)
    idx.set_buckets(X, labels, L)
    idx.debug_emit_all(True)
    dd, ii = idx.scan_topk(Q, order, 10)
    active, _, fallbacks = idx.prefilter_stats()
    assert active
    worst, worst_at, checked = 0.0, None, 0
    for b in range(L):
        rows_b = np.flatnonzero(labels == b)                      # bucket order = ascending original row
        s_c = oracle.forward_logits([(X[rows_b], np.zeros(rows_b.size, np.float32))], Q, nthreads=8)  # chain from 0
        for q in range(NQ):
            r, shat, cnt, eps2, qs, xs = idx.debug_read_candidates(q * NB + b)
            assert cnt == rows_b.size, f"bucket {b} query {q}: {cnt} of {rows_b.size} rows emitted"
            assert eps2 > 0 and np.isfinite(eps2)
            sp = s_c[q, r].astype(np.float64) * float(qs) * float(xs)   # exact: powers of two
            ratio = np.abs(shat.astype(np.float64) - sp) / (0.5 * eps2)
            checked += r.size
            j = int(np.argmax(ratio))
            if ratio[j] > worst:
                worst, worst_at = float(ratio[j]), (b, q, int(r[j]), float(shat[j]), float(sp[j]), 0.5 * eps2)
    print(f"[bound] {kind:11s} d={d:4d}: max |shat - s'|/eps' = {worst:.4f} over {checked} (query,row) pairs "
          f"(fallback slots {fallbacks})")
    assert worst < 1.0, f"bound violated: ratio {worst} at (bucket, query, row, shat, s', eps') = {worst_at}"
    # and the answers are the exact mode's / the oracle's, whatever path (fallback) produced them
    idx.close()
    ex = _capi.Index(0, prefilter=False)
    ex.set_buckets(X, labels, L)
    d0, i0 = ex.scan_topk(Q, order, 10)
    ex.close()
    np.testing.assert_array_equal(ii, i0)
    np.testing.assert_array_equal(dd, d0)
    do, io, _ = oracle.search(None, None, X, Q[:16], labels[:, None], NB, 10, nthreads=8, bucket_order=order[:16][:, :, None])
    np.testing.assert_array_equal(ii[:16], io)
    np.testing.assert_array_equal(dd[:16].astype(np.float64), do)
