#!/usr/bin/env python3
"""PCIe-inclusive rate: wepp_place_batch with HOST buffers (validation, H2D of the reads, kernels, D2H of the results)
on the bench workload, for every sub-batch count of the pipeline (wepp_mat_set_pipeline) with pageable and with
caller-pinned buffers; WEPP_DEBUG_TIMING=1 in the environment adds the library's own print per call (when each
sub-batch's results were enqueued, the call's total).  PCIE_SUBBATCHES="1,2,4" restricts the sweep."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wepp_amd as w
g = w.generate_tree(21, int(os.environ.get("PROBE_NODES", "16000000")))
batches = [g.reads(22 + i, 1_000_000) for i in range(4)]
mat = w.Mat(g.tree, device=0)
out = None
pin = lambda a: torch.from_numpy(a.copy()).pin_memory().numpy()
pinned = []
for b in batches:
    r = w.Reads.__new__(w.Reads)
    r.read_off, r.read_word = pin(b.read_off), pin(b.read_word)
    pinned.append(r)
n = batches[0].n_reads
pout = w.PlacementResult(pin(np.zeros(n, np.uint32)), pin(np.zeros(n, np.int32)), pin(np.zeros(n, np.uint32)), pin(np.zeros(n, np.uint32)))
for S in [int(x) for x in os.environ.get("PCIE_SUBBATCHES", "1,2,3,4,8").split(",")]:
    mat.set_pipeline(S)
    for label, bs, o in (("pageable", batches, None), ("pinned", pinned, pout)):
        res = mat.place_batch(bs[0], out=o)
        ts = []
        for i in range(8):
            t0 = time.perf_counter()
            res = mat.place_batch(bs[i % 4], out=res)
            ts.append(time.perf_counter() - t0)
        print(json.dumps({"sub_batches": S, "buffers": label, "ms_median": round(float(np.median(ts)) * 1e3, 3),
                          "ms_min": round(min(ts) * 1e3, 3), "reads_per_s": round(n / float(np.median(ts)))}), flush=True)
mat.close()
