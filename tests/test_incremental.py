"""The incremental CPU checker (oracle/incremental_oracle.c, one walk per read) against the
faithful restatement of mapper2_body (oracle/mapper2_oracle.c): every per-node value of the
-p mode, the winner, the counts and the list of optimal nodes, on >= 10 000 random
(tree, sample) pairs (SURVEY.md Appendix B fuzz recipe).  Only after this does a GPU test use
the incremental checker at sizes the faithful one cannot reach."""
import numpy as np

import fuzz_trees as ft


def _cols(S):
    return list(zip(*S)) if S else ([], [], [], [])


def _check_pair(ot, inc, S):
    cols = _cols(S)
    a = ot.place_sample(*cols, want_best_vec=True)
    b = inc.place_sample(*cols, want_best_vec=True)
    for k in ("score", "num_best", "best_j", "best_node_id", "has_unique"):
        assert a[k] == b[k], (k, a, b, S)
    assert a["best_j_vec"].tolist() == b["best_j_vec"].tolist()
    pa = ot.place_sample(*cols, per_node_scores=True)
    pb = inc.place_sample(*cols, per_node_scores=True)
    assert pa["node_scores"].tolist() == pb["node_scores"].tolist(), S


def test_incremental_equals_faithful_on_10k_pairs(oracle):
    rng = np.random.default_rng(20240)
    pairs = 0
    while pairs < 10500:
        # genome 60: positions collide, back-mutations and repeated positions along a path
        tree, ref = ft.random_tree(rng)
        ot = oracle.OracleTree(tree)
        inc = ot.incremental()
        for _ in range(6):
            _check_pair(ot, inc, ft.random_sample(rng, ref))
            pairs += 1
        inc.close()
        ot.close()


def test_incremental_equals_faithful_dense_variants(oracle):
    """Shapes the default recipe under-samples: many masked nodes, many ambiguity codes, long
    samples (every position listed), deep chains."""
    rng = np.random.default_rng(20241)
    for it in range(400):
        kind = it % 4
        if kind == 0:
            tree, ref = ft.random_tree(rng, p_masked=0.3, p_root_masked=0.5)
        elif kind == 1:
            tree, ref = ft.random_tree(rng, p_ambig=0.5, genome=12)
        elif kind == 2:
            tree, ref = ft.random_tree(rng, genome=20, max_muts=8)
        else:
            tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(40, 200)), genome=30)
        ot = oracle.OracleTree(tree)
        inc = ot.incremental()
        for _ in range(4):
            genome = 12 if kind == 1 else 20 if kind == 2 else 30 if kind == 3 else 60
            _check_pair(ot, inc, ft.random_sample(rng, ref, genome=genome, max_k=genome if kind == 2 else 7))
        inc.close()
        ot.close()


def test_grouped_walk_equals_faithful_on_10k_pairs(oracle):
    """inc_place_batch streams the tree once per GROUP of 16 reads (static fast path + per-read
    evaluation of the nodes with a listed mutation): the five result fields of every read against
    the faithful restatement, partial and multiple groups, 1-3 threads."""
    rng = np.random.default_rng(20242)
    pairs = 0
    it = 0
    while pairs < 10500:
        it += 1
        if it % 3 == 0:
            tree, ref = ft.random_tree(rng, p_masked=0.1, p_ambig=0.2, genome=40)
            genome = 40
        else:
            tree, ref = ft.random_tree(rng)
            genome = 60
        k = int(rng.integers(1, 50))
        samples = [ft.random_sample(rng, ref, genome=genome, max_k=int(rng.integers(0, 12))) for _ in range(k)]
        reads = ft.reads_from_samples(samples)
        ot = oracle.OracleTree(tree)
        inc = ot.incremental()
        a = ot.place_batch(reads, 1)
        b = inc.place_batch(reads, nthreads=int(rng.integers(1, 4)))
        assert (a == b).all(), (np.nonzero(a != b), samples)
        pairs += k
        inc.close()
        ot.close()


def test_incremental_batch_driver(oracle):
    import wepp_amd as w
    g = w.generate_tree(5, 4000, genome_len=3000, p_ambiguous=0.02, p_masked_node=0.003, root_mutations=2)
    reads = g.reads(6, 300, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.004, p_n=0.01,
                    p_iupac=0.1)
    ot = oracle.OracleTree(g.tree)
    inc = ot.incremental()
    a = ot.place_batch(reads, nthreads=4)
    b = inc.place_batch(reads, nthreads=3)
    assert (a == b).all()
    long_reads = g.reads(7, 40, read_len=1200, amplicon_len=1200, amplicon_step=1100, p_substitution=0.03, p_n=0.02,
                         p_iupac=0.1)
    assert (ot.place_batch(long_reads, nthreads=4) == inc.place_batch(long_reads, nthreads=2)).all()
