#!/usr/bin/env python3
"""Device-resident default step (1 M reads, 8 batches in rotation) under several environment settings, one handle each,
interleaved and repeated (boxes and clocks drift): tools/diag/step_probe.py "" "WEPP_WW_BLOCK_MAX_SMALL=100000 WEPP_WW_BLOCK_MAX_BIG=100000" ...
PROBE_NODES (16000000), PROBE_SHAPE (json of generator kwargs, e.g. {"p_hub": 0.5})."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import wepp_amd as w
from bench import DeviceBatch
nodes = int(os.environ.get("PROBE_NODES", 16_000_000))
shape = json.loads(os.environ.get("PROBE_SHAPE", "{}"))
g = w.generate_tree(21, nodes, **shape)
flat = w.FlatView(g.tree)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
batches = [DeviceBatch(torch, g.reads(22 + i, 1_000_000, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001, p_n=0.005), dev) for i in range(8)]
settings = sys.argv[1:] or [""]
mats = []
for setting in settings:
    keys = []
    for kv in setting.split():
        k, v = kv.split("=", 1)
        os.environ[k] = v
        keys.append(k)
    mats.append(w.Mat(None, device=0, flat=flat))
    for k in keys:
        del os.environ[k]
    for b in batches:
        b.place(mats[-1], stream)
torch.cuda.synchronize()
best = [1e9] * len(mats)
for rep in range(4):
    for i, mat in enumerate(mats):
        t0 = time.perf_counter()
        for s in range(40):
            batches[s % 8].place(mat, stream)
        torch.cuda.synchronize()
        best[i] = min(best[i], (time.perf_counter() - t0) / 40)
for setting, b in zip(settings, best):
    print(f"{setting or 'defaults':70s} {b * 1e3:8.4f} ms/step (best of 4 x 40)", flush=True)
