"""Whole-genome samples through the seeded path (wepp_amd/csrc/seed_kernels.hip) against the oracle: generated trees
at the product's thresholds, and the adversarial fuzz trees with every read forced through the seed kernel (one block
per chunk, no minimum of hard entries, walks off)."""
import numpy as np
import pytest

import fuzz_trees as ft
import wepp_amd as w
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu
PLAN_SEED = 6


def genome_samples(g, seed, n, genome_len, p_sub=0.001, p_n=0.002, p_iupac=0.0):
    return g.reads(seed, n, read_len=genome_len, amplicon_len=genome_len, amplicon_step=genome_len, p_substitution=p_sub,
                   p_n=p_n, p_iupac=p_iupac)


def test_seeded_samples_vs_oracle(oracle):
    """200 K-node tree, whole-genome samples of three noise levels: all of them seeded, every result equal to the
    faithful oracle's (a share) and the incremental checker's (all), to the tile sweeps' (seeds off) and to the
    placement without any work skipping; nearly all chunks are ruled out."""
    L = 29903
    g = w.generate_tree(31, 200000, p_ambiguous=0.01, p_masked_node=0.002, root_mutations=1)
    mat = w.Mat(g.tree)
    st = mat.stats
    assert st.seed_chunks > 100 and st.seed_sig_bytes > 0
    ot = oracle.OracleTree(g.tree)
    inc = oracle.IncrementalTree(ot)
    for i, (p_sub, p_n, p_iupac, n) in enumerate(((0.001, 0.002, 0.0, 300), (0.0001, 0.0005, 0.2, 300), (0.003, 0.01, 0.05, 200))):
        reads = genome_samples(g, 500 + i, n, L, p_sub, p_n, p_iupac)
        mat.timing_reset()
        res = mat.place_batch(reads)
        cls, _ = mat.last_plans(reads.n_reads)
        assert (cls == PLAN_SEED).mean() > 0.95, np.bincount(cls)
        samples, evaluated, total = mat.last_seeds()
        assert samples == int((cls == PLAN_SEED).sum()) and total == samples * st.seed_chunks
        assert evaluated < total // 20, (evaluated, total)          # the signatures rule out at least 95 % of the chunks
        assert_same(res, inc.place_batch(reads, nthreads=8), f"seeded batch {i} vs the incremental checker")
        few = reads.slice(0, 6)
        assert_same(mat.place_batch(few), ot.place_batch(few, 8), f"seeded batch {i} vs the faithful oracle")
        mat.set_use_seeds(False)
        assert_same(res, _as_dict(mat.place_batch(reads)), f"seeded batch {i} vs tile sweeps")
        mat.set_use_seeds(True)
    mat.set_use_crowns(False)
    assert_same(res, _as_dict(mat.place_batch(reads)), "seeded batch vs no work skipping")
    mat.close()


def _as_dict(res):
    return dict(score=res.score, best_j=res.best_bfs_j, num_best=res.num_best, has_unique=res.has_unique)


def test_samples_that_rule_nothing_out(oracle):
    """Samples the bound cannot help: far from every node (random alleles: the best score stays near the root's),
    identical to the reference but for Ns, empty -- the seed kernel evaluates every chunk it must; same results."""
    L = 29903
    g = w.generate_tree(32, 120000)
    rng = np.random.default_rng(5)
    samples = []
    for i in range(40):
        k = int(rng.integers(20, 90))
        pos = np.sort(rng.choice(np.arange(1, L + 1), size=k, replace=False))
        ents = []
        for p in pos:
            u = rng.random()
            ref = 1 << int(rng.integers(0, 4))
            if i % 4 == 3 or u < 0.3:
                ents.append((int(p), ref, 15, 1))                                   # N
            else:
                a = 1 << int(rng.integers(0, 4))
                ents.append((int(p), ref, a if a != ref else (ref << 1 if ref < 8 else 1), 0))
        samples.append(ents)
    reads = w.Reads.from_lists(samples)
    mat = w.Mat(g.tree)
    res = mat.place_batch(reads)
    cls, _ = mat.last_plans(reads.n_reads)
    assert (cls == PLAN_SEED).sum() >= 20
    assert_same(res, oracle.IncrementalTree(oracle.OracleTree(g.tree)).place_batch(reads, nthreads=8), "samples far from the tree")
    mat.close()


def test_second_pass_for_samples_with_many_chunks_left(oracle, monkeypatch):
    """A sample far from every node faces levels of hundreds of chunks that can all tie: from a level of 256 chunks on
    it is handed to the second pass (32 workgroups per sample, the bound shared, partials combined).  One block per
    chunk makes a 120 K-node tree 1 900 chunks; the same samples with the second pass off and against the checker."""
    monkeypatch.setenv("WEPP_SEED_CHUNK_BLOCKS", "1")
    L = 29903
    g = w.generate_tree(33, 120000)
    rng = np.random.default_rng(6)
    samples = []
    for i in range(300):                       # (more than the second pass's table holds: the overflow stays with the first)
        k = int(rng.integers(25, 70))
        pos = np.sort(rng.choice(np.arange(1, L + 1), size=k, replace=False))
        ents = []
        for p in pos:
            ref = 1 << int(rng.integers(0, 4))
            if rng.random() < 0.2:
                ents.append((int(p), ref, 15, 1))
            else:
                a = 1 << int(rng.integers(0, 4))
                ents.append((int(p), ref, a if a != ref else (ref << 1 if ref < 8 else 1), 0))
        samples.append(ents)
    reads = w.Reads.from_lists(samples)
    mat = w.Mat(g.tree)
    assert mat.stats.seed_chunks > 1500
    res = mat.place_batch(reads)
    cls, _ = mat.last_plans(reads.n_reads)
    assert (cls == PLAN_SEED).sum() >= 250
    _, evaluated, _ = mat.last_seeds()
    assert evaluated > 256 * 100                # (they do face such levels)
    assert_same(res, oracle.IncrementalTree(oracle.OracleTree(g.tree)).place_batch(reads, nthreads=8), "second pass vs the checker")
    res2 = mat.place_batch(reads)               # (the table is cleared between calls)
    assert_same(res2, _as_dict(res), "second call")
    mat.close()
    monkeypatch.setenv("WEPP_SEED_HEAVY", "0")
    m1 = w.Mat(g.tree)
    assert_same(m1.place_batch(reads), _as_dict(res), "one pass")
    m1.close()


@pytest.mark.parametrize("blocks", [1, 3])
def test_fuzz_trees_through_the_seed_kernel(oracle, monkeypatch, blocks):
    """Every read of the adversarial fuzz (masked nodes, multi-allelic alleles, repeated positions, back-mutations,
    IUPAC / N entries, empty reads) forced through the seed kernel: no minimum of hard entries or nodes, one / three
    blocks per chunk, walks off (a read that can walk never reaches the sweeps' plans)."""
    monkeypatch.setenv("WEPP_SEED_MIN_HARD", "0")
    monkeypatch.setenv("WEPP_SEED_MIN_NODES", "0")
    monkeypatch.setenv("WEPP_SEED_CHUNK_BLOCKS", str(blocks))
    rng = np.random.default_rng(77 + blocks)
    seeded = 0
    for it in range(80):
        if it % 4 == 3:
            tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(200, 1500)), genome=int(rng.choice([30, 300])), p_masked=0.04, p_ambig=0.12)
            genome = 30
        else:
            tree, ref = ft.random_tree(rng)
            genome = 60
        reads = ft.reads_from_samples([ft.random_sample(rng, ref, genome=genome, max_k=int(rng.integers(0, 12)))
                                       for _ in range(int(rng.integers(1, 120)))])
        mat = w.Mat(tree)
        mat.set_use_walk(False)
        res = mat.place_batch(reads)
        cls, _ = mat.last_plans(reads.n_reads)
        seeded += int((cls == PLAN_SEED).sum())
        assert_same(res, oracle.OracleTree(tree).place_batch(reads, 8), f"seed fuzz {it}")
        mat.close()
    assert seeded > 3000
