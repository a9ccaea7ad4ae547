#!/usr/bin/env python3
"""tools/seed_probe.py [nodes] [samples]: whole-genome samples on the bench MAT, seeded (seed_kernels.hip) against the
tile sweeps: device-resident time per batch, chunks evaluated per sample, equality of the two paths and with the
incremental checker on a subsample.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import wepp_amd as w
    import oracle_bridge as ob
    from bench import DeviceBatch
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    n_legs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    L = 29903
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    g = w.generate_tree(21, nodes)
    mat = w.Mat(g.tree)
    st = mat.stats
    out = {"nodes": nodes, "setup_s": round(time.perf_counter() - t0, 1), "seed_chunks": int(st.seed_chunks),
           "seed_chunk_blocks": int(st.seed_chunk_blocks), "seed_sig_mb": round(st.seed_sig_bytes / 2**20, 1), "legs": []}
    stream = torch.cuda.current_stream().cuda_stream
    inc = None
    for label, p_sub, p_n in (("~60 entries (p_sub 1e-3, p_n 5e-4)", 0.001, 0.0005), ("~30 entries (p_sub 1e-4, p_n 2e-4)", 0.0001, 0.0002),
                              ("~200 entries (p_sub 1e-3, p_n 5e-3)", 0.001, 0.005))[:n_legs]:
        reads = g.reads(900, n, read_len=L, amplicon_len=L, amplicon_step=L, p_substitution=p_sub, p_n=p_n)
        b = DeviceBatch(torch, reads, dev)
        mat.set_use_seeds(True)
        b.place(mat, stream)
        torch.cuda.synchronize()
        mat.timing_reset()
        t0 = time.perf_counter()
        for _ in range(3):
            b.place(mat, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        ms, _, _, alg = mat.last_timing()
        samples, ev, tot, most, hist = mat.last_seeds(detail=True)
        cls, _ = mat.last_plans(n)
        seeded = [t.clone() for t in b.out]
        leg = {"samples": label, "n": n, "mean_entries": b.nw / n, "seeded_ms": dt * 1e3, "kernel_ms": ms, "samples_per_s": n / dt,
               "algorithmic_bytes": alg, "plan_classes": np.bincount(cls, minlength=7).tolist(),
               "chunks_evaluated_per_sample": ev / max(1, samples), "most_chunks_one_sample": most, "samples_by_chunks_le_1_4_16_64_256_1024_4096_more": hist,
               "chunks": int(st.seed_chunks)}
        # tile sweeps on a subsample
        m = min(n, 1000)
        sub = DeviceBatch(torch, reads.slice(0, m), dev)
        mat.set_use_seeds(False)
        sub.place(mat, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sub.place(mat, stream)
        torch.cuda.synchronize()
        leg["tile_sweep_ms_per_1000"] = (time.perf_counter() - t0) * 1e3 * 1000 / m
        leg["seeded_equals_tile_sweeps"] = all(bool((a[:m] == c).all()) for a, c in zip(seeded, sub.out))
        if nodes <= 4_000_000 or label.startswith("~60"):
            if inc is None:
                inc = ob.IncrementalTree(ob.OracleTree(g.tree))
            k = 64
            want = inc.place_batch(reads.slice(0, k), nthreads=16)
            leg["matches_incremental_checker"] = bool((want["score"] == seeded[1][:k].cpu().numpy()).all() and
                                                      (want["best_j"] == seeded[0][:k].cpu().numpy().view(np.uint32)).all() and
                                                      (want["num_best"] == seeded[2][:k].cpu().numpy().view(np.uint32)).all())
        out["legs"].append(leg)
        print(json.dumps(leg), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
