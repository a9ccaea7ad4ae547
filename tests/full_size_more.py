#!/usr/bin/env python3
"""Run by tests/test_gpu_parity.py::test_full_size_genome_samples_epp_fitch in a CHILD process (product defaults: no
WEPP_IX_PRE_MIN_NODES=0): the three paths whose speeds are quoted at 16 M nodes, oracle-checked AT 16 M nodes.
 (a) whole-genome samples (what read_vcf makes of consensus genomes, src/mutation_annotated_tree.cpp:2033-2130)
     through the seeded path (seed_kernels.hip): every 4th against the incremental checker, 4 against the faithful
     restatement of mapper2_body (src/usher_mapper.cpp:168-506), best_j_vec of 50, seeds off / work skipping off;
 (b) WEPP's own placer (wepp_epp_map, src/WEPP/initial_filter.cpp:41-239): per-read outputs of a 20 000-read call for
     300+ reads against oracle_epp_map, every output of a call on those reads, run-to-run identity of the big call;
 (c) per-site Fitch-Sankoff (wepp_fitch_plan_run, src/usher_mapper.cpp:7-162) on the 16 M-node topology: 64 rows
     against oracle_mapper_body.
Prints one JSON line; exits non-zero on a mismatch."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import oracle_bridge  # noqa: E402
import wepp_amd as w  # noqa: E402
from nrich_full_size import gather, same  # noqa: E402

L = 29903


def main():
    assert "WEPP_IX_PRE_MIN_NODES" not in os.environ, "this check is about the product's defaults"
    n_nodes = int(os.environ.get("FULL_NODES", "16000000"))
    which = os.environ.get("FULL_PARTS", "abc")
    nthr = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.perf_counter()
    g = w.generate_tree(21, n_nodes)
    mat = w.Mat(g.tree)
    ot = oracle_bridge.OracleTree(g.tree)
    out = {"nodes": n_nodes, "host_threads": nthr, "setup_s": round(time.perf_counter() - t0, 1)}

    if "a" in which:
        t1 = time.perf_counter()
        inc = ot.incremental()
        legs = []
        for seed, n_s, p_sub, p_n in ((901, 1600, 0.001, 0.0015), (902, 800, 0.0002, 0.003)):
            rd = g.reads(seed, n_s, read_len=L, amplicon_len=L, amplicon_step=L, p_substitution=p_sub, p_n=p_n)
            mat.timing_reset()
            res = mat.place_batch(rd)
            cls, _ = mat.last_plans(n_s)
            seeded, evaluated, total = mat.last_seeds()
            assert (cls == w.PLAN_SEED).mean() > 0.95, np.bincount(cls).tolist()
            every = np.arange(0, n_s, 4)
            same(res, every, inc.place_batch(gather(rd, every), nthreads=nthr), f"whole-genome samples (seed {seed})")
            few = np.array([0, n_s // 2])
            same(res, few, ot.place_batch(gather(rd, few), nthreads=nthr, node_parallel=True), f"whole-genome samples vs the faithful oracle (seed {seed})")
            vec = mat.best_nodes(rd.slice(0, 25), w.PlacementResult(res.best_bfs_j[:25], res.score[:25], res.num_best[:25], res.flags[:25]))
            for q in range(25):
                want = inc.place_sample(*rd.entries(q), want_best_vec=True)["best_j_vec"]
                if vec[q].tolist() != want.tolist():
                    raise SystemExit(f"best_j_vec of whole-genome sample {q}: {vec[q].tolist()[:8]} vs {want.tolist()[:8]}")
            mat.set_use_seeds(False)
            r1 = mat.place_batch(rd)
            mat.set_use_seeds(True)
            mat.set_use_crowns(False)
            mat.set_use_walk(False)
            r0 = mat.place_batch(rd.slice(0, 256))
            mat.set_use_crowns(True)
            mat.set_use_walk(True)
            for f in ("score", "best_bfs_j", "num_best", "flags"):
                assert (getattr(r1, f) == getattr(res, f)).all(), ("seeds on / off", f)
                assert (getattr(r0, f) == getattr(res, f)[:256]).all(), ("work skipping on / off", f)
            legs.append({"samples": n_s, "mean_entries": float(rd.read_off[-1]) / n_s, "seeded": int(seeded), "checked_incremental": int(len(every)),
                         "checked_faithful": 2, "best_j_vec_checked": 25, "chunks_evaluated_per_sample": evaluated / max(1, seeded),
                         "chunks": int(mat.stats.seed_chunks)})
        inc.close()
        out["whole_genome_samples"] = {"legs": legs, "seconds": round(time.perf_counter() - t1, 1)}

    if "b" in which:
        t1 = time.perf_counter()
        n_e = int(os.environ.get("FULL_EPP_READS", "20000"))
        rd = g.reads(77, n_e, windows=True, max_degree=3, p_substitution=0.003, p_n=0.01)
        a = mat.epp_map(rd, L, want_counts=False)
        b = mat.epp_map(rd, L, want_counts=False)
        for k in ("max_parsimony", "multiplicity", "epp_off", "epp_nodes", "score", "divergence"):
            assert np.array_equal(a[k], b[k], equal_nan=True), ("EPP run to run", k)
        idx = np.arange(0, n_e, max(1, n_e // 300))
        base = gather(rd, idx)
        sub = w.EppReads(base.read_off, base.read_word, rd.start[idx].copy(), rd.end[idx].copy(), rd.degree[idx].copy())
        gs = mat.epp_map(sub, L)
        t2 = time.perf_counter()
        os_ = ot.epp_map(sub, L, nthreads=nthr)
        t_or = time.perf_counter() - t2
        for k in ("max_parsimony", "multiplicity", "epp_off", "epp_nodes", "counts"):
            assert np.array_equal(gs[k], os_[k]), ("EPP vs oracle", k)
        assert np.all(np.abs(gs["score"] - os_["score"]) <= 1e-9 * (1 + np.abs(os_["score"]))), "EPP scores"
        assert ((gs["score"] == 0) == (os_["score"] == 0)).all(), "EPP exact zeros"
        assert np.array_equal(gs["divergence"], os_["divergence"], equal_nan=True), "EPP divergence"
        # the per-read outputs of the big call for the same reads (a read's outputs do not depend on its batch)
        assert (a["max_parsimony"][idx] == os_["max_parsimony"]).all() and (a["multiplicity"][idx] == os_["multiplicity"]).all()
        for i, q in enumerate(idx):
            x = a["epp_nodes"][int(a["epp_off"][q]):int(a["epp_off"][q + 1])]
            y = os_["epp_nodes"][int(os_["epp_off"][i]):int(os_["epp_off"][i + 1])]
            assert np.array_equal(x, y), ("EPP list of read", int(q))
        out["epp"] = {"reads": n_e, "checked_against_oracle": int(len(idx)), "oracle_s": round(t_or, 1),
                      "mean_multiplicity": float(a["multiplicity"].mean()), "seconds": round(time.perf_counter() - t1, 1)}
        del a, b, gs, os_

    if "c" in which:
        t1 = time.perf_counter()
        tree = g.tree
        n = tree.n_nodes
        has_child = np.zeros(n, bool)
        has_child[tree.parent[tree.parent >= 0]] = True
        leaves = np.flatnonzero(~has_child).astype(np.uint32)
        rng = np.random.default_rng(5)
        rows = int(os.environ.get("FULL_FITCH_ROWS", "64"))
        per = max(1, int(len(leaves) * 0.002))
        site_ref = (1 << rng.integers(0, 4, rows)).astype(np.uint8)
        var_off = (np.arange(rows + 1, dtype=np.uint64) * per).astype(np.uint32)
        var_node = np.concatenate([rng.choice(leaves, per, replace=False) for _ in range(rows)]).astype(np.uint32)
        var_nuc = (1 << rng.integers(0, 4, rows * per)).astype(np.uint8)
        amb = rng.random(rows * per) < 0.05
        var_nuc[amb] |= (1 << rng.integers(0, 4, int(amb.sum()))).astype(np.uint8)
        bare = w.Tree(tree.parent, np.zeros(n + 1, np.uint32), [], [], [])
        plan = w.FitchPlan(bare)
        s, nd, par, mut = plan.run(site_ref, var_off, var_node, var_nuc, capacity=int(1.3 * rows * per + rows))
        plan.close()
        ob = oracle_bridge.OracleTree(bare)

        def one(r):
            x, y = int(var_off[r]), int(var_off[r + 1])
            return ob.mapper_body(int(site_ref[r]), var_node[x:y].astype(np.int32), var_nuc[x:y])

        t2 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=min(nthr, 16)) as ex:
            wants = list(ex.map(one, range(rows)))
        t_or = time.perf_counter() - t2
        k = 0
        for r, want in enumerate(wants):
            got = list(zip(nd[k:k + len(want)].tolist(), par[k:k + len(want)].tolist(), mut[k:k + len(want)].tolist()))
            if not ((s[k:k + len(want)] == r).all() and got == want):
                raise SystemExit(f"Fitch-Sankoff row {r} differs from oracle_mapper_body")
            k += len(want)
        assert k == len(s), (k, len(s))
        ob.close()
        out["fitch"] = {"rows": rows, "observations_per_row": per, "mutations": int(len(s)), "oracle_s": round(t_or, 1),
                        "seconds": round(time.perf_counter() - t1, 1)}
    ot.close()
    mat.close()
    out["total_s"] = round(time.perf_counter() - t0, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
