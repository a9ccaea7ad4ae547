#!/usr/bin/env python3
"""Clock ticks by phase of k_route (block 0, wave 0) on the default step; needs a -DWEPP_ROUTE_STATS build:
  tools/build_variant.sh rstats -DWEPP_ROUTE_STATS && WEPP_PLACE_LIB=variants/rstats/libwepp_place.so python tools/diag/route_stats.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import wepp_amd as w
from wepp_amd import _lib
nodes = int(os.environ.get("PROBE_NODES", 4_000_000))
g = w.generate_tree(21, nodes)
reads = g.reads(52, 1_000_000, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001, p_n=0.005)
mat = w.Mat(g.tree)
for _ in range(5):
    mat.place_batch(reads)
lib = _lib.lib
out = (ctypes.c_ulonglong * 16)()
lib.wepp_debug_route_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
rc = lib.wepp_debug_route_stats(out)
v = list(out)
n = max(v[9], 1)
names = ["prologue", "offsets+words", "A: counts, root", "A: tree-wide stream", "A: window crown", "A: first loads", "B: class, plan, counters", "lists", "epilogue"]
print("launches", v[9], {nm: round(v[i] / n, 1) for i, nm in enumerate(names)}, "ticks per launch (s_memtime)")
