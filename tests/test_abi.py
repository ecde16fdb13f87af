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
    assert L.fmx_abi_version() == 5
    n = ctypes.c_int(-1)
    assert L.fmx_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.fmx_result) == 24
    assert ctypes.sizeof(_lib.fmx_limits) == 24
    assert ctypes.sizeof(_lib.fmx_stats_t) == 232


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


def test_fm_sibling_is_held_to_the_reference_checks(tmp_path, testdata):
    """NaiveFMSearcher cannot open an index without X.fm and FMLoader throws on a bad one (bwtmerger.scala:259-262,339:
    element size 4, size * 4 + 9 == file length, n = fm.size).  The engine never reads the lists, so a missing .fm is
    fine -- but one that is there must pass those checks and describe the .bwt's n rows; a bad one fails the open with
    FMX_ERR_FORMAT before any device is touched (so this runs without a GPU), a good one lets the open go on."""
    import shutil
    import struct
    L = _lib.load()
    h = ctypes.c_void_p()
    for name, be in (("test1024.cmp", False), ("words", True)):
        for ext in (".bwt", ".aux"):
            shutil.copy(os.path.join(testdata, name + ext), tmp_path / (name + ext))
        bwt, aux, fm = tmp_path / (name + ".bwt"), tmp_path / (name + ".aux"), tmp_path / (name + ".fm")
        n = struct.unpack(">q" if be else "<q", bwt.read_bytes()[:8])[0]
        fmt = ">q" if be else "<q"

        def rc_of():
            return L.fmx_open(str(bwt).encode(), str(aux).encode(), 1 if be else 0, 0, ctypes.byref(h))
        fm.write_bytes(b"\x08" + struct.pack(fmt, n) + b"\0" * (8 * n))
        assert rc_of() == 2 and b"bad elSize 8" in L.fmx_last_error()
        fm.write_bytes(b"\x04" + struct.pack(fmt, n) + b"\0" * (4 * n - 4))
        assert rc_of() == 2 and b"bad size" in L.fmx_last_error() and b"0x9" in L.fmx_last_error()
        fm.write_bytes(b"\x04" + struct.pack(fmt, n - 1) + b"\0" * (4 * (n - 1)))
        assert rc_of() == 2 and b"not one index" in L.fmx_last_error()
        fm.write_bytes(b"\x04\0\0")
        assert rc_of() == 2
        fm.write_bytes(b"\x04" + struct.pack(fmt, n) + b"\0" * (4 * n))       # well-formed: the open goes on to the device
        rc = rc_of()
        assert rc in (0, 5), L.fmx_last_error()
        if rc == 0:
            L.fmx_close(h)


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


def test_header_is_plain_c_and_links(tmp_path):
    """include/fmx.h is the boundary a JNI / cgo / FFI stub compiles against: it must be valid strict C99 and
    C++11 on its own, and a C program using only it must link against libfmx.so (no torch, no HIP headers)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "use_fmx.c"
    src.write_text(
        "#include <fmx.h>\n#include <stdio.h>\n"
        "int main(void) {\n"
        "  fmx_stats_t s; fmx_limits l; fmx_result r; int n = -1;\n"
        "  (void)s; (void)l; (void)r;\n"
        "  if (fmx_abi_version() <= 0) return 2;\n"
        "  if (fmx_device_count(&n) != FMX_OK || n < 0) return 3;\n"
        "  printf(\"%d %d %d\\n\", (int)sizeof(fmx_stats_t), (int)sizeof(fmx_limits), (int)sizeof(fmx_result));\n"
        "  return 0;\n}\n")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + inc, "-c", str(src),
                           "-o", str(tmp_path / "a.o")])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I" + inc, "-x", "c++", "-c", str(src),
                           "-o", str(tmp_path / "b.o")])
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "use_fmx"
    subprocess.check_call(["gcc", str(tmp_path / "a.o"), "-L" + libdir, "-lfmx", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stderr[-500:])
    assert out.stdout.split() == [str(ctypes.sizeof(_lib.fmx_stats_t)), str(ctypes.sizeof(_lib.fmx_limits)),
                                  str(ctypes.sizeof(_lib.fmx_result))]


def test_product_library_has_no_fault_injection_switch():
    """fmx_comm.cpp's FMX_COMM_FAIL_INIT switch exists in the tests' twin of the library only."""
    from findex_amd import build
    assert b"FMX_COMM_FAIL_INIT" not in open(_lib.LIB_PATH, "rb").read()
    assert b"FMX_COMM_FAIL_INIT" in open(build.OUT_FAULTS, "rb").read()
    twin = ctypes.CDLL(build.OUT_FAULTS)
    for name in declared_functions():
        assert hasattr(twin, name), name


def test_build_lists_cover_the_sources():
    """Every source and header under findex_amd/csrc is named in findex_amd/build.py (the staleness check and
    the compile list are driven by those lists)."""
    from findex_amd import build
    have = set(os.listdir(build.CSRC))
    assert {f for f in have if f.endswith((".hip", ".cpp"))} == set(build.SOURCES)
    assert {f for f in have if f.endswith(".h")} == set(build.HEADERS)


def test_config_keys_and_values():
    """fmx_config_set (include/fmx.h): every documented key takes its documented values and refuses others with
    FMX_ERR_ARG and a message -- no device needed."""
    L = _lib.load()
    ok = {b"layout": [b"onehot", b"bytes", b"auto"], b"checkpoints": [b"superblock", b"auto"], b"ktab": [b"off", b"auto"],
          b"jump": [b"off", b"rows", b"rows3", b"jumps", b"auto"], b"pipeline": [b"on", b"off"], b"validate": [b"1", b"0"],
          b"threads": [b"3", b"0"], b"jump_chars": [b"8", b"11", b"9"], b"tables_after": [b"auto", b"100000", b"0"],
          b"jump_pairs": [b"on", b"off", b"auto"], b"table_budget": [b"100000000000", b"0.5", b"1.0", b"0", b"auto"]}
    for key, values in ok.items():
        for v in values:                        # the last value of each list is the default: left in place
            assert L.fmx_config_set(key, v) == 0, (key, v)
        if key != b"validate":                  # ("validate" reads anything but "0" as on)
            assert L.fmx_config_set(key, b"no-such-value") == 3, key
            assert L.fmx_last_error().decode()
    for key, bad in ((b"jump_chars", b"7"), (b"jump_chars", b"12"), (b"tables_after", b"-1"), (b"table_budget", b"1.5"),
                     (b"table_budget", b"0.0"), (b"table_budget", b"-5")):
        assert L.fmx_config_set(key, bad) == 3, (key, bad)
    # the 8-byte interval form's host-side decode needs no device either
    import numpy as np
    pk = np.array([5 | (3 << 40), 7 | (0xFFFFFF << 40), 9, 1, 1, 7 + (1 << 30)], dtype=np.uint64)      # k = 3: a hit, a wide one, a miss; one escape
    sp, ep = np.zeros(3, dtype=np.uint64), np.zeros(3, dtype=np.uint64)
    vp = ctypes.c_void_p
    assert L.fmx_packed_words(3, 1) == 6
    assert L.fmx_unpack_intervals(pk.ctypes.data_as(vp), 3, 1, sp.ctypes.data_as(vp), ep.ctypes.data_as(vp)) == 0
    assert sp.tolist() == [5, 7, 9] and ep.tolist() == [8, 7 + (1 << 30), 9]
    assert L.fmx_unpack_intervals(pk.ctypes.data_as(vp), 3, 0, sp.ctypes.data_as(vp), ep.ctypes.data_as(vp)) == 9       # FMX_ERR_OVERFLOW
    pk[4] = 3                                                                                                             # an escape entry that names pattern 3 of 3
    assert L.fmx_unpack_intervals(pk.ctypes.data_as(vp), 3, 1, sp.ctypes.data_as(vp), ep.ctypes.data_as(vp)) == 2       # FMX_ERR_FORMAT
    assert L.fmx_config_set(b"no-such-key", b"1") == 3
    assert L.fmx_config_set(None, b"1") == 3 and L.fmx_config_set(b"jump", None) == 3
    # the per-handle form refuses a null handle before it looks at the key; closing nothing is not an error (fmx.h)
    assert L.fmx_index_config_set(None, b"jump", b"off") == 3
    assert L.fmx_close(None) == 0
    assert L.fmx_prepare_ex(None, 1, 0) == 3
