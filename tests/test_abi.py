"""The C-ABI library loads and exports every symbol include/wepp_place.h
declares; error reporting works without a GPU."""
import ctypes
import os
import re

import pytest

import wepp_amd as w
from wepp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "wepp_place.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(wepp_[a-z_0-9]+)\s*\(", src))
    names.discard("wepp_pack_read_word")  # static inline
    return sorted(names)


def test_every_declared_symbol_is_exported():
    names = _declared()
    assert len(names) >= 18
    L = ctypes.CDLL(w.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/wepp_place.h but not exported"
    assert sorted(names) == sorted(_lib.EXPORTED), "python binding and header disagree"


def test_header_compiles_as_c_and_cpp(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "wepp_place.h"\nint main(void){return (int)wepp_pack_read_word(1,1,2,0)==0;}\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-x", "c++", "-I", inc, "-c", str(src), "-o",
                           str(tmp_path / "t2.o")])


def test_pack_read_word_matches_c_macro():
    assert int(w.pack_read_word(29903, 8, 15, 1)) == (29903 | (8 << 20) | (15 << 24) | (1 << 28))
    p, r, m, ms = w.unpack_read_word(w.pack_read_word([5, 7], [1, 2], [4, 15], [0, 1]))
    assert p.tolist() == [5, 7] and r.tolist() == [1, 2] and m.tolist() == [4, 15] and ms.tolist() == [0, 1]


def test_errors_are_codes_and_messages_not_exits():
    t = w.Tree.from_lists([-1, -1], [[], []])
    with pytest.raises(w.WeppError) as ei:
        w.FlatView(t)
    assert ei.value.code == 1
    assert _lib.lib.wepp_last_error().decode() != ""
    # null handles
    assert _lib.lib.wepp_mat_get_stats(None, None) == 1
    assert _lib.lib.wepp_place_batch(None, None, None, 0, None, None, None, None, None) == 1


def test_mat_create_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device wepp_mat_create returns WEPP_EDEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    g = w.generate_tree(1, 100)
    with pytest.raises(w.WeppError) as ei:
        w.Mat(g.tree)
    assert ei.value.code == 3 and "no CPU fallback" in str(ei.value)
