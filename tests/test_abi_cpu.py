"""CPU-only: the C-ABI library builds, loads and exports every symbol that
include/bgamd.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from bias_gan_amd import _lib
    return _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bgamd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 30
    cdll = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(cdll, n), f"{n} declared in include/bgamd.h but not exported"
    # and the Python binding covers exactly the header
    assert sorted(lib.EXPORTS) == names


def test_abi_version_and_error_string(lib):
    l = lib.load()
    assert l.bg_abi_version() == lib.ABI_VERSION
    assert isinstance(l.bg_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    # descriptor checks run on the host before any launch
    l = lib.load()
    d = lib.ConvDesc(lib.BF16, 1, 8, 8, 12, 8, 8, 16, 1, 1, 1, 0, 1, 12, 16)
    rc = l.bg_conv2d_fwd(ctypes.byref(d), 16, 16, None, 16, None)
    assert rc == -1 and b"multiples of 8" in l.bg_last_error()


def test_missing_library_fails_loudly(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libbgamd.so")
    with pytest.raises(RuntimeError, match="no PyTorch/CPU fallback"):
        lib.load()
