"""The flattener under AddressSanitizer + UBSan on the CPU (window-stream builder, position index, pre-test bytes):
tests/cxx/flatten_sanitized.cpp linked with the host-only sources of wepp_amd/csrc."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "wepp_amd", "csrc")
SRCS = [os.path.join(ROOT, "tests", "cxx", "flatten_sanitized.cpp")] + [os.path.join(CS, f) for f in ("flatmat.cpp", "gen.cpp", "flat_debug.cpp", "errors.cpp")]


@pytest.mark.parametrize("pre_min", ["0", "8192"])
def test_flattener_under_sanitizers(tmp_path, pre_min):
    exe = str(tmp_path / "flatten_sanitized")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            *SRCS, "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("no sanitizer runtime: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, WEPP_IX_PRE_MIN_NODES=pre_min, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0 and run.stdout.strip().startswith("ok "), (run.stdout[-500:], run.stderr[-3000:])
