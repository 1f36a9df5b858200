"""CPU (`-m "not gpu"`): liblmi_hip.so loads and exports every symbol include/lmi_hip.h declares.
No compute is called (there is no GPU in the build container)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "lmi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lmi_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from learnedmetricindex_amd import _capi

    names = _declared()
    assert len(names) >= 18
    assert os.path.exists(_capi.LIB_PATH), "build liblmi_hip.so first (__graft_entry__.build())"
    L = ctypes.CDLL(_capi.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in lmi_hip.h but not exported"
    assert sorted(_capi.SIGNATURES) == names  # the Python binding covers exactly the header
    assert L.lmi_abi_version() == 1


def test_missing_library_fails_loudly(monkeypatch):
    from learnedmetricindex_amd import _capi

    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/liblmi_hip.so")
    try:
        _capi.lib()
    except _capi.LmiError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected LmiError")
