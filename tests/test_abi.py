"""The C-ABI library loads on a machine without a GPU and exports every symbol include/fmx.h
declares; argument validation and error reporting work without touching a device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from findex_amd import _lib


def declared_functions():
    text = open(os.path.join(ROOT, "include", "fmx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fmx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_binds():
    assert declared_functions() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(L, name), name


def test_abi_version_and_device_count():
    L = _lib.load()
    assert L.fmx_abi_version() == 1
    n = ctypes.c_int(-1)
    assert L.fmx_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.fmx_result) == 24
    assert ctypes.sizeof(_lib.fmx_limits) == 24
    assert ctypes.sizeof(_lib.fmx_stats_t) == 80


def test_open_errors_are_statuses_with_messages(tmp_path, testdata):
    L = _lib.load()
    h = ctypes.c_void_p()
    # missing file -> FMX_ERR_IO (reference: "File %s does not exists", bwtmerger.scala:430)
    rc = L.fmx_open(b"/nonexistent/x.bwt", b"/nonexistent/x.aux", 1, 0, ctypes.byref(h))
    assert rc == 1 and b"does not exists" in L.fmx_last_error()
    # wrong endianness -> the size check fails (bwtmerger.scala:153)
    rc = L.fmx_open(os.path.join(testdata, "words.bwt").encode(), os.path.join(testdata, "words.aux").encode(), 0, 0,
                    ctypes.byref(h))
    assert rc == 2 and b"bad size" in L.fmx_last_error()
    # truncated aux
    bad = tmp_path / "t.aux"
    bad.write_bytes(b"\0" * 100)
    rc = L.fmx_open(os.path.join(testdata, "test1024.cmp.bwt").encode(), str(bad).encode(), 0, 0, ctypes.byref(h))
    assert rc == 2
    # null arguments
    assert L.fmx_open(None, None, 1, 0, ctypes.byref(h)) == 3
    assert L.fmx_n(None, None) == 3


def test_no_cpu_fallback_without_device(testdata):
    """Without a HIP device open() must fail loudly (FMX_ERR_HIP), never compute on the CPU."""
    L = _lib.load()
    n = ctypes.c_int()
    L.fmx_device_count(ctypes.byref(n))
    if n.value:
        pytest.skip("a HIP device is present")
    h = ctypes.c_void_p()
    rc = L.fmx_open(os.path.join(testdata, "words.bwt").encode(), os.path.join(testdata, "words.aux").encode(), 1, 0,
                    ctypes.byref(h))
    assert rc == 5 and b"no CPU fallback" in L.fmx_last_error()
    import findex_amd
    with pytest.raises(findex_amd.FmxError):
        findex_amd.HipFMSearcher(os.path.join(testdata, "words.bwt"))


def test_product_never_touches_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "findex_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower(), os.path.join(dirpath, f)
