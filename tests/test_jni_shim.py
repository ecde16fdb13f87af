"""bindings/fmx_jni.c and bindings/hipfm.scala cannot be built here (no JDK, no scalac), so they are kept honest at
the text / type level: the C shim must type-check against include/fmx.h as it is NOW (a stand-in jni.h with the JNI
specification's signatures stands for the JDK's), and the Scala natives and the C definitions must pair up one to one
with matching arities and JNI types."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI_C = os.path.join(ROOT, "bindings", "fmx_jni.c")
SCALA = os.path.join(ROOT, "bindings", "hipfm.scala")
STANDIN = os.path.join(ROOT, "tests", "native", "jni_standin")

SCALA_TO_JNI = {"Long": "jlong", "Int": "jint", "Boolean": "jboolean", "String": "jstring", "Unit": "void", "Double": "jdouble",
                "Array[Byte]": "jbyteArray", "Array[Long]": "jlongArray", "Array[Int]": "jintArray",
                "Array[Double]": "jdoubleArray", "ByteBuffer": "jobject"}


def scala_natives():
    src = open(SCALA).read()
    out = {}
    for m in re.finditer(r"@native\s+def\s+(\w+)\s*\(([^)]*)\)\s*:\s*([\w\[\]]+)", src, re.S):
        params = [p.split(":")[1].strip() for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = (params, m.group(3))
    return out


def c_natives():
    src = open(JNI_C).read()
    out = {}
    for m in re.finditer(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+FN\((\w+)\)\s*\(([^)]*)\)", src, re.S):
        params = [" ".join(p.split()[:-1]) for p in m.group(3).split(",")]
        out[m.group(2)] = (params, m.group(1))
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_jni_shim_type_checks_against_the_header():
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        "-I" + STANDIN, JNI_C], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_scala_native_has_its_c_definition():
    sc, cc = scala_natives(), c_natives()
    assert len(sc) >= 30 and len(cc) >= 30
    assert sorted(sc) == sorted(cc), "natives without a partner: %s" % sorted(set(sc) ^ set(cc))
    for name, (sparams, sret) in sc.items():
        cparams, cret = cc[name]
        assert cparams[:2] == ["JNIEnv", "jobject"] or cparams[0].startswith("JNIEnv"), name
        want = [SCALA_TO_JNI[p] for p in sparams]
        got = cparams[2:]
        assert got == want, "%s: Scala %s -> %s, C has %s" % (name, sparams, want, got)
        if sret == "Array[Byte]":
            assert cret == "jbyteArray", name
        elif sret == "ByteBuffer":
            assert cret == "jobject", name
        else:
            assert cret == SCALA_TO_JNI[sret], "%s returns %s in Scala, %s in C" % (name, sret, cret)


def test_jni_symbol_prefix_matches_the_scala_object():
    # object HipFM in package org.fmindex -> JVM class org.fmindex.HipFM$ -> Java_org_fmindex_HipFM_00024_<name>
    src = open(SCALA).read()
    assert re.search(r"^package org\.fmindex", src, re.M)
    assert re.search(r"object HipFM\b", src)
    assert "#define FN(name) Java_org_fmindex_HipFM_00024_##name" in open(JNI_C).read()


def test_close0_is_reachable_from_exactly_one_guarded_site():
    """`def close() = close0(h)` and `override def finalize() = close0(h)` on an immutable handle freed an
    index twice whenever a caller closed and the GC later finalized.  Text-level (no scalac here): besides its @native
    declaration, close0 is called at ONE place; that place takes the handle out of an AtomicLong with getAndSet(0) and skips
    0; close() and finalize() both go through it; every other native sees the handle through the accessor that refuses 0."""
    src = open(SCALA).read()
    calls = [m.start() for m in re.finditer(r"\bclose0\s*\(", src)]
    decl = [m.start() for m in re.finditer(r"@native\s+def\s+close0\s*\(", src)]
    assert len(decl) == 1 and len(calls) == 2, "close0 must be declared once and called once: %d uses" % len(calls)
    site = src[src.rfind("\n", 0, max(calls)) + 1: src.find("\n", max(calls))]
    assert "getAndSet(0L)" in site and "!= 0L" in site and "private def release" in site, site
    assert re.search(r"def close\(\): Unit = release\(\)", src) and re.search(r"override def finalize\(\): Unit = release\(\)", src)
    assert re.search(r"protected def h: Long = \{\s*val v = hbox\.get\s*if \(v == 0L\) throw", src), "natives must refuse a closed handle"
    assert "protected val h: Long" not in src, "a searcher keeps its handle in the AtomicLong box, not in an immutable val"
    # the resident regex batch frees its natives once, too
    assert re.search(r"def close\(\): Unit = if \(closed\.compareAndSet\(false, true\)\)", src)
    # and fmx_close(NULL) is a no-op on the C side (tests/test_abi.py::test_config_keys_and_values calls it)
    api = open(os.path.join(ROOT, "findex_amd", "csrc", "fmx_api.cpp")).read()
    assert re.search(r"int fmx_close\(fmx_index \*idx\) \{\s*if \(!idx\) return FMX_OK;", api)


def test_packed_decode_checks_the_escape_entrys_pattern():
    """HipFMSearcher.unpack narrowed an escape entry's pattern id with .toInt and indexed with it unchecked; the C
    decoder answers FMX_ERR_FORMAT there.  The Scala decode now compares the Long with k before narrowing, and the JNI entry
    point refuses escapeCap > k (which also keeps its capacity arithmetic from wrapping)."""
    src = open(SCALA).read()
    assert re.search(r"if \(ql < 0L \|\| ql >= k\.toLong\) throw", src)
    c = open(JNI_C).read()
    assert "escapeCap > k" in c and "GetDirectBufferCapacity(e, out) / 8 <" in c
