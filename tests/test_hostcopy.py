"""The host-side copy threads of fdoct_process's pinned staging (fdoct_amd/csrc/fdoct_hostcopy.h) -- pure host code, so it is
checked here on the CPU under ThreadSanitizer and AddressSanitizer / UBSan (sanitizers run on the CPU build only)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "hostcopy_check.cpp")


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_copy_threads_move_every_byte_under_sanitizers(tmp_path, sanitizer):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "hostcopy_check")
    build = subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", SRC, "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed: " + build.stderr.strip().splitlines()[-1])
    assert build.returncode == 0, build.stderr
    # ThreadSanitizer needs an address-space layout it can map (the kernel's default ASLR entropy may not be): setarch -R where present
    cmd = [exe]
    if sanitizer == "thread" and shutil.which("setarch"):
        cmd = ["setarch", os.uname().machine, "-R", exe]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if run.returncode != 0 and "unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.startswith("ok ")
