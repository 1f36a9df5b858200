"""CPU (`-m "not gpu"`): li.LearnedIndex's content fingerprints of the caller's frames (host logic of the drop-in boundary, SURVEY Q1 /
section 8b): the xxhash path and the zlib fallback see the same edits; the full fingerprint catches an in-place edit of any row, the
sampled one only of sampled rows."""
import builtins
import sys

import numpy as np
import pandas as pd


def _frames():
    rs = np.random.RandomState(0)
    a = rs.randn(20_000, 16).astype(np.float32)
    return a, pd.DataFrame(a, copy=False)


def test_full_and_sampled_fingerprints():
    from learnedmetricindex_amd.li import LearnedIndex as L

    a, df = _frames()
    full0, samp0 = L._frame_fingerprint(df, True), L._frame_fingerprint(df, False)
    assert L._frame_fingerprint(df.copy(), True) == full0          # equal content, another object
    a[12_345, 3] += 1.0                                            # an in-place edit of one row (not among the 4 096 sampled ones)
    assert L._frame_fingerprint(df, True) != full0
    rows = np.unique(np.linspace(0, df.shape[0] - 1, num=4096, dtype=np.int64))
    assert 12_345 not in rows and L._frame_fingerprint(df, False) == samp0
    a[rows[7], 0] += 1.0                                           # a sampled row
    assert L._frame_fingerprint(df, False) != samp0


def test_zlib_fallback_when_xxhash_is_missing(monkeypatch):
    from learnedmetricindex_amd.li import LearnedIndex as L

    real_import = builtins.__import__

    def no_xxhash(name, *args, **kw):
        if name == "xxhash":
            raise ImportError("xxhash hidden by the test")
        return real_import(name, *args, **kw)

    monkeypatch.setattr(builtins, "__import__", no_xxhash)
    monkeypatch.delitem(sys.modules, "xxhash", raising=False)
    h = L._Hasher()
    assert h._h is None                                            # the zlib pair is in use
    a, df = _frames()
    f0 = L._frame_fingerprint(df, True)
    assert f0 == L._frame_fingerprint(df.copy(), True)
    a[5, 5] -= 2.0
    assert L._frame_fingerprint(df, True) != f0
    assert L._array_fingerprint(np.arange(10)) != L._array_fingerprint(np.arange(1, 11))
