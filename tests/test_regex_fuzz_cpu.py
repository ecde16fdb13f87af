"""The C++ regex front-end under AddressSanitizer + UBSan on the CPU (the GPU boxes have no sanitizer runs):
random and malformed regexes must end in tables with in-range indexes or in RegexError, nothing else."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.timeout(600)
def test_regex_front_end_fuzz_under_sanitizers(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = tmp_path / "fuzz_regex"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "findex_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "fuzz_regex.cpp"), os.path.join(ROOT, "findex_amd", "csrc", "fmx_regex.cpp"),
           "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not available: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    for seed in (11, 12):
        run = subprocess.run([str(exe), str(seed), "20000"], capture_output=True, text=True, timeout=400)
        assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
        assert "compiled" in run.stdout
