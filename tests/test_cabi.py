"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU and
exports every symbol include/bcnn_hip.h declares (no compute calls here)."""
import ctypes
import os

from bcnn_amd import _lib


def test_library_builds_and_exports_every_declared_symbol():
    _lib.build()
    assert os.path.exists(_lib.LIB_PATH)
    declared = _lib.declared_symbols()
    assert len(declared) >= 40
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(raw, s)]
    assert not missing, missing
    # the Python signature table covers the header exactly
    assert set(declared) == set(_lib.SIGNATURES)


def test_no_cpu_fallback_when_library_missing(tmp_path, monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must fail loudly without the HIP library")


def test_product_does_not_import_oracle():
    """The product (bcnn_amd/, include/) must never import, link or dlopen anything under oracle/."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|libbcnn_oracle|libbcnn_ref|oracle/_ref|#include\s+\"[^\"]*oracle", re.M)
    for sub in ("bcnn_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(root, sub)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".h", ".c", ".cpp")) or fn == "Makefile":
                    text = open(os.path.join(dirpath, fn), errors="ignore").read()
                    assert not bad.search(text), os.path.join(dirpath, fn)


def test_host_library_builds_and_exports_the_public_api():
    """libbcnn.so (C99 host) exports every BCNN_API function declared in include/bcnn/bcnn.h."""
    import re
    from bcnn_amd import capi
    capi.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "bcnn", "bcnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"BCNN_API\s+[A-Za-z_ \*]+?\b(bcnn_[a-z0-9_]+)\s*\(", text)))
    assert len(declared) >= 52, len(declared)
    raw = ctypes.CDLL(capi.LIB_PATH)
    missing = [s for s in declared if not hasattr(raw, s)]
    assert not missing, missing
