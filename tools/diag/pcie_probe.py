#!/usr/bin/env python3
"""wepp_place_batch from host buffers (pageable and pinned), 1 M short reads, under several settings:
tools/diag/pcie_probe.py "" "WEPP_PIPE_SUB_BATCHES=2" ...   (PROBE_NODES, PROBE_READS; WEPP_DEBUG_TIMING=1 prints the call's timeline)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import wepp_amd as w
nodes = int(os.environ.get("PROBE_NODES", 16_000_000))
n_reads = int(os.environ.get("PROBE_READS", 1_000_000))
g = w.generate_tree(21, nodes)
flat = w.FlatView(g.tree)
batches = [g.reads(52 + i, n_reads, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001, p_n=0.005) for i in range(3)]
for setting in (sys.argv[1:] or [""]):
    keys = []
    for kv in setting.split():
        k, v = kv.split("=", 1)
        os.environ[k] = v
        keys.append(k)
    mat = w.Mat(None, device=0, flat=flat)
    res = mat.place_batch(batches[0])
    for i in range(3):
        res = mat.place_batch(batches[i % 3], out=res)
    t0 = time.perf_counter()
    steps = 12
    for i in range(steps):
        res = mat.place_batch(batches[i % 3], out=res)
    dt = (time.perf_counter() - t0) / steps
    print(f"{setting or 'defaults':50s} {dt * 1e3:8.3f} ms/call {n_reads / dt:10.4g} reads/s", flush=True)
    mat.close()
    for k in keys:
        del os.environ[k]
