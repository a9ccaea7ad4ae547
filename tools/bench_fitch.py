#!/usr/bin/env python3
"""Measurement of the Fitch-Sankoff row (mapper_body): N-node topology, `rows` synthetic VCF
rows in which a random 0.2 % of the leaves carry an alternate allele.  Prints one JSON line;
run under rocprofv3 --kernel-trace --stats for the per-kernel durations
(algorithmic bytes = 2 * N * rows: the decision-table byte written by the forward
pass and read by the backward pass)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wepp_amd as w

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--rows", type=int, default=8192)
ap.add_argument("--cpu-rows", type=int, default=4)
a = ap.parse_args()
g = w.generate_tree(21, a.nodes)
tree = g.tree
n = tree.n_nodes
has_child = np.zeros(n, bool); has_child[tree.parent[tree.parent >= 0]] = True
leaves = np.flatnonzero(~has_child).astype(np.uint32)
rng = np.random.default_rng(5)
per = max(1, int(len(leaves) * 0.002))
site_ref = (1 << rng.integers(0, 4, a.rows)).astype(np.uint8)
var_off = (np.arange(a.rows + 1, dtype=np.uint64) * per).astype(np.uint32)
var_node = np.concatenate([rng.choice(leaves, per, replace=False) for _ in range(a.rows)]).astype(np.uint32)
var_nuc = (1 << rng.integers(0, 4, a.rows * per)).astype(np.uint8)
bare = w.Tree(tree.parent, np.zeros(n + 1, np.uint32), [], [], [])
t0 = time.perf_counter()
plan = w.FitchPlan(bare)
t_plan = time.perf_counter() - t0
plan.run(site_ref[:64], var_off[:65], var_node[:64 * per], var_nuc[:64 * per])   # warm-up (uploads the topology)
cap = int(1.2 * a.rows * per + a.rows)
t0 = time.perf_counter()
s, nd, par, mut = plan.run(site_ref, var_off, var_node, var_nuc, capacity=cap)     # (sizes the plan's decision tables)
dt_first = time.perf_counter() - t0
t0 = time.perf_counter()
s, nd, par, mut = plan.run(site_ref, var_off, var_node, var_nuc, capacity=cap)
dt = time.perf_counter() - t0
phases = w.fitch_last_timing()
out = {"row": "fitch_sankoff (mapper_body)", "nodes": n, "rows": a.rows, "variants_per_row": per,
       "mutations_out": int(len(s)), "plan_create_s": t_plan, "wall_s_first_run_on_the_plan": dt_first, "wall_s_of_one_run_on_the_plan": dt, "rows_per_s_wall": a.rows / dt,
       "phases_ms": phases, "rows_per_s_kernels": a.rows / (phases["kernels_ms"] * 1e-3),
       "algorithmic_bytes": 2 * n * a.rows}
if a.cpu_rows:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_bridge
    ot = oracle_bridge.OracleTree(bare)
    t0 = time.perf_counter()
    k = 0
    ok = True
    for r in range(a.cpu_rows):
        x, y = int(var_off[r]), int(var_off[r + 1])
        want = ot.mapper_body(int(site_ref[r]), var_node[x:y].astype(np.int32), var_nuc[x:y])
        got = [(int(nd[i]), int(par[i]), int(mut[i])) for i in range(k, k + len(want))]
        ok = ok and got == want
        k += len(want)
    cdt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": a.cpu_rows / cdt, "unit": "rows/s", "cores": 1, "kind": "port",
                           "sample": f"first {a.cpu_rows} rows, oracle_mapper_body, 1 thread", "matches_gpu": ok}
print(json.dumps(out))
