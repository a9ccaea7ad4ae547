#!/usr/bin/env python3
"""Does grouping reads of similar work into the same wave help the walks?  Places the bench batches in their own
order and sorted by number of entries (a proxy for the events a read walks); results are order-independent."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wepp_amd as w
from wepp_amd import Reads
from bench import DeviceBatch

g = w.generate_tree(21, 16_000_000)
mat = w.Mat(g.tree)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
legs = [("default", g.reads(22, 1_000_000)), ("p_n=0.02", g.reads(122, 1_000_000, p_n=0.02)), ("p_n=0.05", g.reads(122, 1_000_000, p_n=0.05))]


def reorder(rd, order):
    k = np.diff(rd.read_off).astype(np.int64)
    off = np.zeros(len(order) + 1, np.int64)
    off[1:] = np.cumsum(k[order])
    idx = np.concatenate([np.arange(rd.read_off[q], rd.read_off[q + 1]) for q in order[:0]]) if False else None
    starts = rd.read_off[:-1].astype(np.int64)[order]
    take = np.repeat(starts - off[:-1], k[order]) + np.arange(off[-1])
    return Reads(off.astype(np.uint32), rd.read_word[take])


for name, rd in legs:
    k = np.diff(rd.read_off)
    for label, order in (("batch order", np.arange(rd.n_reads)), ("sorted by entries", np.argsort(k, kind="stable"))):
        b = DeviceBatch(torch, reorder(rd, order), dev)
        b.place(mat, stream); torch.cuda.synchronize()
        mat.timing_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            b.place(mat, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        ms, n, passes, nbytes = mat.last_timing()
        rw, it = mat.last_walk()
        print("%-10s %-18s %.3f ms/step  kernel %.3f ms  wave-iterations/step %d" % (name, label, dt * 1e3, ms, it // 5), flush=True)
