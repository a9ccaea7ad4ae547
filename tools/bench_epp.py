#!/usr/bin/env python3
"""Measurement of WEPP's own read placement (wepp_epp_map = wepp_filter::cartesian_map,
src/WEPP/initial_filter.cpp:140-239): N-node synthetic MAT, R amplicon reads with windows.
Prints one JSON line: reads/s over the whole call (host sort, H2D, kernels, D2H), the device
time by phase, the event-steps swept, and the oracle's serial rate on a bounded sample of the
same reads.  Run under rocprofv3 --kernel-trace --stats for per-kernel durations."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wepp_amd as w

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--cpu-reads", type=int, default=2000)
ap.add_argument("--no-counts", action="store_true")
a = ap.parse_args()
t0 = time.perf_counter()
g = w.generate_tree(21, a.nodes)
amp = max(400, a.read_len)
reads = g.reads(22, a.reads, read_len=a.read_len, amplicon_len=amp, amplicon_step=300 if a.read_len < 400 else 1000,
                windows=True, max_degree=5)
t_gen = time.perf_counter() - t0
t0 = time.perf_counter()
mat = w.Mat(g.tree)
t_mat = time.perf_counter() - t0
kw = dict(want_counts=not a.no_counts, want_divergence=not a.no_counts)
mat.epp_map(reads.__class__(reads.read_off[:65], reads.read_word[: int(reads.read_off[64])], reads.start[:64], reads.end[:64],
                            reads.degree[:64]), 29903, **kw)           # warm-up
best = None
for _ in range(a.steps):
    t0 = time.perf_counter()
    out = mat.epp_map(reads, 29903, **kw)
    dt = time.perf_counter() - t0
    tm = w.epp_last_timing()
    if best is None or dt < best[0]:
        best = (dt, tm)
dt, tm = best
dev_ms = tm["select_ms"] + tm["sweep1_ms"] + tm["sweep2_ms"] + tm["finish_ms"]
res = {"row": "epp_map (cartesian_map)", "nodes": mat.n_nodes, "reads": a.reads, "read_len": a.read_len,
       "events": int(mat.stats.n_mutations) * 2, "wall_s": dt, "reads_per_s_wall": a.reads / dt,
       "device_ms": dev_ms, "reads_per_s_device": a.reads / (dev_ms / 1e3), "phases": tm,
       "event_steps_per_s": 2 * tm["events_swept"] / ((tm["sweep1_ms"] + tm["sweep2_ms"]) / 1e3),
       "mean_multiplicity": float(out["multiplicity"].mean()), "mean_parsimony": float(out["max_parsimony"].mean()),
       "gen_s": t_gen, "mat_create_s": t_mat}
if a.cpu_reads:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_bridge
    n = min(a.cpu_reads, a.reads)
    sub = reads.__class__(reads.read_off[: n + 1], reads.read_word[: int(reads.read_off[n])], reads.start[:n], reads.end[:n],
                          reads.degree[:n])
    ot = oracle_bridge.OracleTree(g.tree)
    t0 = time.perf_counter()
    want = ot.epp_map(sub, genome_size=29903)
    cdt = time.perf_counter() - t0
    ok = bool((want["max_parsimony"] == out["max_parsimony"][:n]).all() and (want["multiplicity"] == out["multiplicity"][:n]).all())
    res["cpu_baseline"] = {"value": n / cdt, "unit": "reads/s", "cores": 1, "kind": "port",
                           "sample": f"first {n} reads incl. arena + range-tree build, oracle_epp_map, 1 thread",
                           "matches_gpu": ok}
print(json.dumps(res))
