"""The host worker pool behind wepp_place_batch (wepp_amd/csrc/host_pool.hpp): a C++ stress test, plain and under
ThreadSanitizer (CPU build only)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cxx", "host_pool_stress.cpp")
INC = os.path.join(ROOT, "wepp_amd", "csrc")


@pytest.mark.parametrize("flags,rounds", [(["-O2"], 20000), (["-O1", "-g", "-fsanitize=thread"], 4000)])
def test_host_pool_stress(tmp_path, flags, rounds):
    exe = str(tmp_path / "host_pool_stress")
    build = subprocess.run(["g++", "-std=c++17", "-pthread", *flags, "-I", INC, SRC, "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "-fsanitize=thread" in flags:
        pytest.skip("no ThreadSanitizer runtime: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe, str(rounds)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.startswith("ok "), (run.stdout[-500:], run.stderr[-2000:])
    assert "ThreadSanitizer" not in run.stderr, run.stderr[-2000:]
