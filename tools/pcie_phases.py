#!/usr/bin/env python3
"""One default batch through wepp_place_batch (host buffers) with WEPP_DEBUG_TIMING=1: the library prints when the
results' D2H was enqueued and the call's total (stderr); pageable and caller-pinned buffers, one device call."""
import os, sys, time
import numpy as np
os.environ["WEPP_DEBUG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wepp_amd as w
g = w.generate_tree(21, int(os.environ.get("PROBE_NODES", "16000000")))
batches = [g.reads(22 + i, 1_000_000) for i in range(2)]
mat = w.Mat(g.tree, device=0)
pin = lambda a: torch.from_numpy(a.copy()).pin_memory().numpy()
n = batches[0].n_reads
pinned = []
for b in batches:
    r = w.Reads.__new__(w.Reads); r.read_off, r.read_word = pin(b.read_off), pin(b.read_word); pinned.append(r)
pout = w.PlacementResult(pin(np.zeros(n, np.uint32)), pin(np.zeros(n, np.int32)), pin(np.zeros(n, np.uint32)), pin(np.zeros(n, np.uint32)))
for label, bs, o in (("pageable", batches, None), ("pinned", pinned, pout)):
    res = mat.place_batch(bs[0], out=o)
    for i in range(4):
        t0 = time.perf_counter()
        res = mat.place_batch(bs[i % 2], out=res)
        print(label, "python-side ms %.3f" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
mat.close()
