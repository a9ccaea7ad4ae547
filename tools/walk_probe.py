#!/usr/bin/env python3
"""Times placement legs on the bench MAT and prints what the walks did (kernel time, wave iterations, algorithmic
bytes, reads per plan class).  Environment: PROBE_NODES (16000000), PROBE_LEGS (comma list of default, k=0, k=1, k=2,
k=4, k=8, p_n=0.02, p_n=0.05, long; default: all but long), PROBE_WALK (1 | 0 | both), PROBE_STEPS (3),
PROBE_READS (1000000).  With WEPP_PLACE_LIB pointing at a -DWEPP_WALK_STATS build and WEPP_WALK_DEBUG=1 the library
also prints the walks' cycles by phase."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wepp_amd as w
from bench import DeviceBatch, truncate_reads

N = int(os.environ.get("PROBE_NODES", "16000000"))
R = int(os.environ.get("PROBE_READS", "1000000"))
STEPS = int(os.environ.get("PROBE_STEPS", "3"))
only = os.environ.get("PROBE_LEGS", "default,k=0,k=1,k=2,k=4,k=8,p_n=0.02,p_n=0.05").split(",")
walks = {"1": (True,), "0": (False,), "both": (True, False)}[os.environ.get("PROBE_WALK", "1")]
g = w.generate_tree(21, N)
mat = w.Mat(g.tree)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
pool = None
for name in only:
    if name == "default":
        rd = g.reads(22, R)
    elif name.startswith("k="):
        pool = pool if pool is not None else g.reads(123, R, p_n=0.06)
        rd = truncate_reads(w, pool, int(name[2:]))
    elif name.startswith("p_n="):
        rd = g.reads(122, R, p_n=float(name[4:]))
    elif name == "long":
        rd = g.reads(24, min(R, 200000), read_len=1200, amplicon_len=1200, amplicon_step=1020, p_substitution=0.03, p_n=0.02)
    else:
        raise SystemExit("unknown leg " + name)
    b = DeviceBatch(torch, rd, dev)
    for walk in walks:
        mat.set_use_walk(walk)
        b.place(mat, stream); torch.cuda.synchronize()
        mat.timing_reset()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            b.place(mat, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / STEPS
        ms, n, passes, nbytes = mat.last_timing()
        rw, it = mat.last_walk()
        cls, st = mat.last_plans(rd.n_reads)
        plans = {w.PLAN_NAMES[c]: int(k) for c, k in enumerate(np.bincount(cls, minlength=6)) if k}
        print("%-10s walk=%d reads %d entries/read %.2f  %.3f ms/step  kernel %.3f ms  %.3g reads/s  walked %d  wave-iterations/step %d  "
              "alg bytes/step %.4g (%.0f GB/s)  plans %s  streams %s" %
              (name, walk, rd.n_reads, b.nw / max(1, rd.n_reads), dt * 1e3, ms, rd.n_reads / dt, rw, it // STEPS, nbytes, nbytes / (ms * 1e-3) / 1e9,
               plans, np.bincount(st, minlength=15).tolist()), flush=True)
