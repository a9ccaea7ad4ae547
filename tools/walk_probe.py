#!/usr/bin/env python3
"""Times placement legs (walk on / off) on the bench MAT and prints the walks' iteration counts."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wepp_amd as w
from bench import DeviceBatch, truncate_reads

N = int(os.environ.get("PROBE_NODES", "16000000"))
g = w.generate_tree(21, N)
mat = w.Mat(g.tree)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
pool = g.reads(123, 1_000_000, p_n=0.06)
legs = [("default", g.reads(22, 1_000_000))] + [("k=%d" % k, truncate_reads(w, pool, k)) for k in (0, 1, 2, 4, 8)]
legs.append(("p_n=0.02", g.reads(122, 1_000_000, p_n=0.02)))
legs.append(("p_n=0.05", g.reads(122, 1_000_000, p_n=0.05)))
only = os.environ.get("PROBE_LEGS")
if only:
    legs = [l for l in legs if l[0] in only.split(",")]
for name, rd in legs:
    b = DeviceBatch(torch, rd, dev)
    for walk in (True, False):
        mat.set_use_walk(walk)
        b.place(mat, stream); torch.cuda.synchronize()
        mat.timing_reset()
        t0 = time.perf_counter()
        for _ in range(3):
            b.place(mat, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        ms, n, passes, nbytes = mat.last_timing()
        rw, it = mat.last_walk()
        tr = np.bincount(mat.last_tiers(rd.n_reads), minlength=15)
        print("%-10s walk=%d  %.3f ms/step  kernel %.3f ms  walked %d  wave-iterations/step %d  tiers %s" %
              (name, walk, dt * 1e3, ms, rw, it // 3, tr.tolist()), flush=True)
