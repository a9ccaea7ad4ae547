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


def _build(tmp_path, name, sources, extra=()):
    exe = str(tmp_path / name)
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-I", HOST, *sources, *extra, "-lz", "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-3000:]
    return exe


def _fixture_tree(tmp_path, rng, n=40):
    parent = [-1] + [int(rng.integers(0, i)) for i in range(1, n)]
    names = [f"n{i}" for i in range(n)]
    muts = [[(int(rng.integers(1, 200)), 1, 1, int(1 << rng.integers(1, 4)))] if i else [] for i in range(n)]
    path = str(tmp_path / "tree.pb")
    pb_fixture.write_pb(path, parent, names, muts)
    return path


@pytest.mark.parametrize("mode", ["vcf", "reads"])
def test_vcf_and_reads_loaders_survive_damaged_files(tmp_path, mode):
    rng = np.random.default_rng(11)
    lib = os.path.join(ROOT, "wepp_amd")
    try:
        if mode == "vcf":
            aux = _fixture_tree(tmp_path, rng)
            src = str(tmp_path / "samples.vcf")
            samples = [[(int(p), 1, int(1 << rng.integers(1, 4)), 0) for p in sorted(rng.choice(np.arange(1, 200), 5, replace=False))]
                       for _ in range(6)]
            samples[2].append((199, 1, 15, 1))
            pb_fixture.write_vcf(src, [f"s{i}" for i in range(6)], samples)
        else:
            aux = str(tmp_path / "ref.fa")
            ref = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 400))
            open(aux, "w").write(">ref\n" + ref + "\n")
            src = str(tmp_path / "reads.pb")
            reads = []
            for i in range(30):
                st = int(rng.integers(1, 300))
                content = list(ref[st - 1:st + 79])
                for k in rng.choice(len(content), 3, replace=False):
                    content[int(k)] = "ACGTN_"[int(rng.integers(0, 6))]
                reads.append((f"r{i}", st, "".join(content), int(rng.integers(1, 4))))
            pb_fixture.write_reads_pb(src, reads)
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"no fixture: {e!r}")
    srcs = [os.path.join(ROOT, "tests", "cxx", "loader_fuzz.cpp"), os.path.join(HOST, "mat.cpp"), os.path.join(HOST, "wepp_filter.cpp")]
    exe = _build(tmp_path, "loader_fuzz", srcs, extra=["-L", lib, "-lwepp_place", f"-Wl,-rpath,{lib}"])
    run = subprocess.run([exe, mode, aux, src, str(tmp_path / "damaged"), "2000"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=2048"))
    assert run.returncode == 0 and run.stdout.startswith("ok "), (run.stdout[-300:], run.stderr[-3000:])
    loaded, rejected = int(run.stdout.split()[1]), int(run.stdout.split()[3])
    assert loaded + rejected == 2000


def test_tree_description_check_survives_damage(tmp_path):
    """wepp_tree_desc through the flattener's validation (the same code wepp_mat_create runs), ASan + UBSan."""
    cs = os.path.join(ROOT, "wepp_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "cxx", "tree_desc_fuzz.cpp")] + [os.path.join(cs, f) for f in ("flatmat.cpp", "gen.cpp", "flat_debug.cpp", "errors.cpp")]
    exe = _build(tmp_path, "tree_desc_fuzz", srcs)
    run = subprocess.run([exe, "600"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=4096"))
    assert run.returncode == 0 and run.stdout.startswith("ok "), (run.stdout[-300:], run.stderr[-3000:])
    assert int(run.stdout.split()[3]) > 50          # most damage is caught
