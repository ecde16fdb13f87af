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
