"""Mutation fuzz of the .pb loader of the C++ host mirror (wepp_amd/host/mat.cpp) under AddressSanitizer + UBSan:
damaged files load or raise MAT::mat_error, nothing else."""
import os
import subprocess

import numpy as np
import pytest

import pb_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "wepp_amd", "host")


def test_pb_loader_survives_damaged_files(tmp_path):
    rng = np.random.default_rng(7)
    n = 60
    parent = [-1] + [int(rng.integers(0, i)) for i in range(1, n)]
    names = [f"n{i}" for i in range(n)]
    muts = [[(int(rng.integers(1, 200)), int(1 << rng.integers(0, 4)), int(1 << rng.integers(0, 4)), int(1 << rng.integers(0, 4)))
             for _ in range(int(rng.integers(0, 3)))] for _ in range(n)]
    src = str(tmp_path / "tree.pb")
    try:
        pb_fixture.write_pb(src, parent, names, muts)
    except Exception as e:  # noqa: BLE001  (python-protobuf missing or too old)
        pytest.skip(f"no .pb fixture: {e!r}")
    exe = str(tmp_path / "pb_fuzz")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-I", HOST, os.path.join(ROOT, "tests", "cxx", "pb_fuzz.cpp"), os.path.join(HOST, "mat.cpp"),
                            "-lz", "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe, src, str(tmp_path / "damaged.pb"), "3000"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=2048"))
    assert run.returncode == 0 and run.stdout.startswith("ok "), (run.stdout[-300:], run.stderr[-3000:])
    loaded, rejected = int(run.stdout.split()[1]), int(run.stdout.split()[3])
    assert loaded + rejected == 3000 and rejected > 100
