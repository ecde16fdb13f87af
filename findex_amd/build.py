"""Builds libfmx.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m findex_amd.build [--force]

hipcc cross-compiles without a GPU; the .so lands in findex_amd/lib/ (git-ignored,
but shipped to the GPU box with the tree).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libfmx.so")
# the same library with fmx_comm.cpp's fault-injection switch compiled in (-DFMX_FAULT_INJECTION): loaded by one test only
OUT_FAULTS = os.path.join(OUT_DIR, "libfmx_faults.so")
SOURCES = ["fmx_api.cpp", "fmx_hostpar.cpp", "fmx_comm.cpp", "fmx_hostrank.cpp", "fmx_regex.cpp", "fmx_build.hip", "fmx_kernels.hip", "fmx_search.hip", "fmx_ktab.hip", "fmx_jump.hip", "fmx_select.hip", "fmx_frontier.hip", "fmx_refmatch.hip"]
HEADERS = ["fmx_device.h", "fmx_host.h", "fmx_hostpar.h", "fmx_nfa.h", "fmx_regex.h"]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps():
    return [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "fmx.h")]


def _source_stamp():
    """Hash of every file a translation unit can see.  The library is up to date when the stamp written beside it is
    this one -- file times say nothing on a box the tree was copied to, and a source edited WHILE a build runs (hipcc
    reads a .hip file twice, for the device and for the host) must not leave objects that disagree about a struct."""
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


STAMP = OUT + ".stamp"


def _stale():
    if not os.path.exists(OUT) or not os.path.exists(STAMP) or not os.path.exists(OUT_FAULTS):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_stamp()


def _unit_stamp(src, flags):
    """Hash of what ONE translation unit can see: its source, every header, the flags."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for d in [os.path.join(CSRC, src)] + [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(ROOT, "include", "fmx.h")]:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-I" + os.path.join(ROOT, "include"),
             "-I" + CSRC, "-Wall", "-Wno-unused-result"] + os.environ.get("FMX_CXXFLAGS", "").split()
    for attempt in range(3):
        stamp = _source_stamp()
        objs = []
        procs = []
        for src in SOURCES:
            obj = os.path.join(OUT_DIR, src + ".o")
            objs.append(obj)
            # an object is kept when the stamp beside it names exactly what this unit would be compiled from
            ustamp = _unit_stamp(src, flags)
            try:
                with open(obj + ".stamp") as f:
                    fresh = os.path.exists(obj) and f.read().strip() == ustamp
            except OSError:
                fresh = False
            if fresh and not force:
                continue
            cmd = [_hipcc()] + flags + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, obj, ustamp, subprocess.Popen(cmd)))
        for src, obj, ustamp, p in procs:
            if p.wait() != 0:
                raise RuntimeError("hipcc failed on " + src)
            with open(obj + ".stamp", "w") as f:
                f.write(ustamp + "\n")
        if _source_stamp() == stamp:
            break           # (else: the sources changed under the compilers -- again, from what they are now)
    else:
        raise RuntimeError("the sources keep changing while they are compiled")
    tmp = OUT + ".%d.tmp" % os.getpid()
    cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", tmp] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, OUT)
    # ... and the tests' twin: fmx_comm.cpp once more with the fault-injection switch, everything else as it is
    fobj = os.path.join(OUT_DIR, "fmx_comm.cpp.faults.o")
    subprocess.check_call([_hipcc()] + flags + ["-DFMX_FAULT_INJECTION", "-x", "hip", "-c", os.path.join(CSRC, "fmx_comm.cpp"), "-o", fobj])
    ftmp = OUT_FAULTS + ".%d.tmp" % os.getpid()
    subprocess.check_call([_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", ftmp] +
                          [fobj if o.endswith("fmx_comm.cpp.o") else o for o in objs] + ["-ldl"])
    os.replace(ftmp, OUT_FAULTS)
    with open(STAMP, "w") as f:
        f.write(stamp + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
