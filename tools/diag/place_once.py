#!/usr/bin/env python3
"""A few placement calls of the default batch (PROBE_NODES nodes, 1 M reads, host buffers) and nothing else: for kernel
traces of variant builds (WEPP_PLACE_LIB) whose results may be wrong on purpose."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import wepp_amd as w
g = w.generate_tree(21, int(os.environ.get("PROBE_NODES", 4_000_000)))
reads = g.reads(52, 1_000_000, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001, p_n=0.005)
mat = w.Mat(g.tree)
for _ in range(6):
    mat.place_batch(reads)
mat.close()
