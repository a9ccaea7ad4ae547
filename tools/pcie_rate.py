#!/usr/bin/env python3
"""PCIe-inclusive rate: wepp_place_batch with HOST buffers (H2D of the reads, kernels, D2H of the
results, device allocations included) on the bench workload."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wepp_amd as w
g = w.generate_tree(21, 16_000_000)
reads = g.reads(22, 1_000_000, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001, p_n=0.005)
mat = w.Mat(g.tree, device=0)
mat.place_batch(reads.slice(0, 1000))
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    res = mat.place_batch(reads)
    best = min(best, time.perf_counter() - t0)
print(json.dumps({"reads": reads.n_reads, "best_s": best, "reads_per_s_host_buffers": reads.n_reads / best,
                  "note": "includes the C-ABI's validation of the read words and the staged H2D / D2H copies"}))
mat.close()
