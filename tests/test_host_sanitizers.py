"""CPU: the threaded host C++ (csrc/textio.cpp, csrc/juncio.cpp) under AddressSanitizer + UBSan and under
ThreadSanitizer (SURVEY section 5 asks for a race / memory-error check of the native host code).  The
driver (tests/host_sanitize/driver.cpp) pushes every host entry point through its threaded path;
sanitizers belong on the CPU build only."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "splicedice_amd", "csrc")
INPUTS = os.path.join(REPO, "tests", "golden", "quant_c1", "inputs")
FILES = [("s0.junc.bed", 1), ("s1.junc.bed", 1), ("s3.SJ.out.tab", 2), ("s4.plain.bed", 0)]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_host_cpp_under_sanitizer(kind, tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    b = subprocess.run(["make", "-C", CSRC, kind], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout + b.stderr
    exe = os.path.join(REPO, "build", "sanitize", f"host_{kind}")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    args = [exe, str(tmp_path)] + [f"{os.path.join(INPUTS, f)}:{t}" for f, t in FILES]
    r = subprocess.run(args, capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "driver: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
