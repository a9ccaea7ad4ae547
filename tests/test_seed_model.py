"""The seed bound (seed_kernels.hip) on the CPU: for every sample and every node, the value mapper2_body leaves for
the node (oracle, -p mode) is at least the bound of the node's chunk -- so a chunk whose bound exceeds an achieved
score holds no node that wins or ties.  Fuzz trees (masked nodes, multi-allelic alleles, back-mutations, repeated
positions, IUPAC / N entries, entries beyond the tree's last mutated position) with one block per chunk, and
generated trees with whole-genome samples."""
import os

import numpy as np

import fuzz_trees as ft
import seed_model as sdm
import wepp_amd as w


def _cols(S):
    return list(zip(*S)) if S else ([], [], [], [])


def _check(oracle, tree, fv, samples):
    m = sdm.SeedModel(fv)
    assert m.built
    ot = oracle.OracleTree(tree)
    pruned = total = 0
    for S in samples:
        o = ot.place_sample(*_cols(S), per_node_scores=True)
        lb = m.lower_bounds(S)
        vals = o["node_scores"]                       # BFS order; ineligible nodes carry score + 1
        for d in range(tree.n_nodes):
            c = m.chunk_of_dfs(d)
            assert vals[m.dfs2bfs[d]] >= lb[c], (S, d, c, int(vals[m.dfs2bfs[d]]), int(lb[c]))
        # what the kernel relies on: no chunk ruled out by the optimum holds a node that attains it
        pruned += int((lb > o["score"]).sum())
        total += m.nch
    return pruned, total


def test_seed_bound_fuzz(oracle, monkeypatch):
    monkeypatch.setenv("WEPP_SEED_CHUNK_BLOCKS", "1")
    rng = np.random.default_rng(41)
    pruned = total = 0
    for it in range(150):
        genome = int(rng.choice([20, 200, 1500]))
        tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(2, 400)), genome=genome, p_masked=0.04, p_ambig=0.12)
        fv = w.FlatView(tree)
        samples = []
        for _ in range(4):
            S = ft.random_sample(rng, ref, genome=genome, max_k=int(rng.integers(0, 24)))
            if rng.random() < 0.3:       # an entry beyond every mutated position of the tree
                S = S + [(genome + 5, 1, 2, 0)]
            samples.append(S)
        p, t = _check(oracle, tree, fv, samples)
        pruned += p
        total += t
    assert total > 2000 and pruned > 0      # (random samples sit near the root: few chunks are ruled out; the generated trees below show the power)


def test_seed_bound_generated_trees(oracle, monkeypatch):
    monkeypatch.setenv("WEPP_SEED_CHUNK_BLOCKS", "2")
    pruned = total = 0
    for seed in (3, 4):
        g = w.generate_tree(seed, 4000, genome_len=3000, p_ambiguous=0.02, p_masked_node=0.005, root_mutations=1)
        fv = w.FlatView(g.tree)
        reads = g.reads(seed + 10, 6, read_len=3000, amplicon_len=3000, amplicon_step=3000, p_substitution=0.003,
                        p_n=0.004, p_iupac=0.1)
        samples = [[(int(p), int(r), int(a), int(ms)) for p, r, a, ms in zip(*reads.entries(i))]
                   for i in range(reads.n_reads)]
        p, t = _check(oracle, g.tree, fv, samples)
        pruned += p
        total += t
    assert pruned > total // 2                         # whole-genome samples: most chunks cannot hold the placement
