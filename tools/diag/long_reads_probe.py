#!/usr/bin/env python3
"""configs[4]'s shard (125 000 reads of 1.2 kb, device-resident) under several settings of the environment knobs of
one handle each on the same flat image: tools/diag/long_reads_probe.py "WEPP_TARGET_WAVES_DENSE=8192" "WEPP_WINDOWS_UNFUSED=1" ...
(an empty string = the defaults).  PROBE_NODES (16000000), PROBE_READS (125000)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import wepp_amd as w
from bench import DeviceBatch, long_reads_batch
nodes = int(os.environ.get("PROBE_NODES", 16_000_000))
n_reads = int(os.environ.get("PROBE_READS", 125_000))
g = w.generate_tree(21, nodes)
flat = w.FlatView(g.tree)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
batches = [DeviceBatch(torch, long_reads_batch(g, 24 + i, n_reads), dev) for i in range(4)]
ref = None
for setting in (sys.argv[1:] or [""]):
    keys = []
    for kv in setting.split():
        k, v = kv.split("=", 1)
        os.environ[k] = v
        keys.append(k)
    mat = w.Mat(None, device=0, flat=flat)
    for b in batches:
        b.place(mat, stream)
    torch.cuda.synchronize()
    got = [t.cpu().numpy().copy() for t in batches[0].out]
    if ref is None:
        ref = got
    same = all((a == b).all() for a, b in zip(ref, got))
    t0 = time.perf_counter()
    steps = 12
    for i in range(steps):
        batches[i % 4].place(mat, stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{setting or 'defaults':50s} {dt * 1e3:8.3f} ms/step {n_reads / dt:10.4g} reads/s  same_as_first={same}", flush=True)
    mat.close()
    for k in keys:
        del os.environ[k]
