#!/usr/bin/env python3
"""Run by tests/test_gpu_parity.py::test_window_bound_on_fuzz_trees in a CHILD process with WEPP_PLACE_LIB pointing at the
test-only build `win64` (tools/build_variant.sh win64 -DWEPP_WIN_SIZE=64 -DWEPP_WIN_STRIDE=32): genome windows of 64
positions every 32, so that the adversarial fuzz trees (masked nodes, multi-allelic alleles, repeated positions along
a path, back-mutations; genomes of 60-600 positions) straddle window edges and the window-candidate bound
`out_W - in_W <= base(root)` decides on the GPU which nodes a read may be placed on: window-crown
walks, per-read sweeps of a window crown (17-32 entries), window tiles (more than 32 entries), against the faithful
oracle.  Prints one JSON line; exits non-zero on a mismatch."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import fuzz_trees as ft  # noqa: E402
import oracle_bridge  # noqa: E402
import wepp_amd as w  # noqa: E402


def windowed_sample(rng, ref, genome, size, stride, k):
    """k entries inside one genome window (or spilling one position over its edge)"""
    wi = int(rng.integers(0, max(1, (genome - 1) // stride + 1)))
    lo = max(1, wi * stride)
    hi = min(genome, wi * stride + size - 1 + (1 if rng.random() < 0.2 else 0))
    if hi < lo:
        return []
    poss = sorted(set(int(x) for x in rng.integers(lo, hi + 1, size=k)))
    ents = []
    for p in poss:
        u = rng.random()
        if u < 0.55:
            ents.append((p, ref[p], 1 << int(rng.integers(0, 4)), 0))
        elif u < 0.75:
            a = int(rng.integers(1, 16))
            ents.append((p, ref[p], a, 1 if (a == 15 and rng.random() < 0.5) else 0))
        else:
            ents.append((p, ref[p], 15, 1))
    return ents


def main():
    st0 = None
    rng = np.random.default_rng(2026)
    seen = np.zeros(8, np.int64)
    crowned = trees_with_crowns = n_reads = 0
    for it in range(int(os.environ.get("WINFUZZ_TREES", "120"))):
        genome = int(rng.choice([60, 150, 300, 600]))
        n_nodes = int(rng.integers(5, 65)) if it % 3 == 0 else int(rng.integers(200, 2500))
        tree, ref = ft.random_tree(rng, n_nodes=n_nodes, genome=genome, p_masked=0.03, p_ambig=0.1, max_muts=3 if n_nodes > 64 else 4)
        samples = []
        for _ in range(int(rng.integers(20, 200))):
            u = rng.random()
            if u < 0.5:
                samples.append(windowed_sample(rng, ref, genome, 64, 32, int(rng.integers(0, 9))))
            elif u < 0.7:
                samples.append(windowed_sample(rng, ref, genome, 64, 32, int(rng.integers(17, 33))))
            elif u < 0.85:
                samples.append(windowed_sample(rng, ref, genome, 64, 32, int(rng.integers(33, 64))))
            else:
                samples.append(ft.random_sample(rng, ref, genome=genome, max_k=int(rng.integers(0, 12))))
        reads = ft.reads_from_samples(samples)
        mat = w.Mat(tree)
        if st0 is None:
            st0 = mat.stats
            assert st0.window_size == 64 and st0.window_stride == 32, "this script needs the win64 build (WEPP_PLACE_LIB)"
        trees_with_crowns += int(mat.stats.n_window_crowns > 0)
        want = oracle_bridge.OracleTree(tree).place_batch(reads, 8)
        for walk in (True, False):
            mat.set_use_walk(walk)
            res = mat.place_batch(reads)
            for f, k in (("score", "score"), ("best_bfs_j", "best_j"), ("num_best", "num_best"), ("has_unique", "has_unique")):
                bad = np.flatnonzero(getattr(res, f) != want[k])
                if bad.size:
                    raise SystemExit(f"tree {it} (walk {walk}): {f} differs at reads {bad[:5].tolist()}: {getattr(res, f)[bad[:5]].tolist()} vs "
                                     f"{want[k][bad[:5]].tolist()}; samples {[samples[i] for i in bad[:2]]}")
            cls, pst = mat.last_plans(reads.n_reads)
            if walk:
                seen += np.bincount(cls, minlength=8)
                crowned += int(((pst == w.WINDOW_CROWN_SLOT) & (cls != w.PLAN_WIN)).sum())
        n_reads += reads.n_reads
        mat.close()
    out = {"trees": it + 1, "reads": n_reads, "trees_with_window_crowns": trees_with_crowns, "reads_on_window_crowns": crowned,
           "reads_by_plan_class": {w.PLAN_NAMES[c]: int(n) for c, n in enumerate(seen[:7]) if n}}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
