"""WEPP's own read placement on the GPU (wepp_epp_map) against the oracle's restatement of
wepp_filter::cartesian_map (src/WEPP/initial_filter.cpp:140-239).  Integer outputs are bit-exact;
haplotype scores are sums of doubles the reference itself adds in a thread-dependent order
(:186-199): tolerance 1e-9 * (1 + |score|), the library accumulates them in 64-bit fixed point."""
import numpy as np
import pytest

import epp_fuzz
import fuzz_trees as ft
import wepp_amd as w

pytestmark = pytest.mark.gpu


def _check(got, want, n_reads, tag):
    assert (got["max_parsimony"] == want["max_parsimony"]).all(), tag
    assert (got["multiplicity"] == want["multiplicity"]).all(), tag
    assert (got["epp_off"] == want["epp_off"]).all(), tag
    assert (got["epp_nodes"] == want["epp_nodes"]).all(), tag
    assert np.all(np.abs(got["score"] - want["score"]) <= 1e-9 * (1 + np.abs(want["score"]))), tag
    assert ((got["score"] == 0) == (want["score"] == 0)).all(), tag
    assert (got["counts"] == want["counts"]).all(), tag
    assert np.array_equal(got["divergence"], want["divergence"], equal_nan=True), tag


@pytest.mark.parametrize("rpl", [1, 4])
def test_fuzz_small_trees(oracle, rpl, monkeypatch):
    """rpl = reads per lane of the sweep (a tile is 64 * rpl reads); the library picks 4 for big batches."""
    monkeypatch.setenv("WEPP_EPP_RPL", str(rpl))
    rng = np.random.default_rng(31337)
    for it in range(40):
        genome = 60
        tree, ref = ft.random_tree(rng, genome=genome)
        reads = epp_fuzz.random_epp_reads(rng, tree, ref, genome, n_reads=int(rng.integers(1, 600)))
        mat = w.Mat(tree)
        got = mat.epp_map(reads, genome)
        want = oracle.OracleTree(tree).epp_map(reads, genome_size=genome)
        _check(got, want, reads.n_reads, it)
        mat.close()


def test_short_list_buffer_is_fetched_not_recomputed(oracle):
    """wepp_epp_map with too small an epp_nodes buffer (or none: the capacity query) still delivers every other
    output, reports WEPP_ELIMIT with epp_off filled, and the lists come from wepp_epp_fetch_lists -- equal to a call
    whose buffer was large enough; a second fetch has nothing left."""
    import ctypes
    rng = np.random.default_rng(4242)
    genome = 60
    tree, ref = ft.random_tree(rng, genome=genome, n_nodes=60)
    reads = epp_fuzz.random_epp_reads(rng, tree, ref, genome, n_reads=300)
    mat = w.Mat(tree)
    full = mat.epp_map(reads, genome)
    assert len(full["epp_nodes"]) > 8
    for cap in (0, 1, len(full["epp_nodes"]) - 1):
        got = mat.epp_map(reads, genome, epp_capacity=cap)          # (the binding fetches on WEPP_ELIMIT)
        _check(got, full, reads.n_reads, cap)
    rc = w._lib.lib.wepp_epp_fetch_lists(mat._h, None, 0)
    assert rc == 1 and "no EPP lists are pending" in w._lib.lib.wepp_last_error().decode()
    mat.close()


def test_edge_cases(oracle):
    # single-node tree, reads with no mutations, all-N reads, a window of one base, degree 0
    tree = w.Tree.from_lists([-1], [[]])
    reads = w.EppReads.from_lists([[], [(7, w.A, w.N, 1)], [(9, w.C, w.T)], []], start=[1, 5, 9, 60], end=[60, 9, 9, 60],
                                  degree=[1, 2, 0, 5])
    mat = w.Mat(tree)
    got = mat.epp_map(reads, 60)
    want = oracle.OracleTree(tree).epp_map(reads, genome_size=60)
    _check(got, want, 4, "single")
    assert got["max_parsimony"].tolist() == [0, 0, 1, 0] and got["multiplicity"].tolist() == [1, 1, 1, 1]
    mat.close()
    # no reads at all
    tree, ref = ft.random_tree(np.random.default_rng(5), n_nodes=30)
    mat = w.Mat(tree)
    empty = w.EppReads.from_lists([], [], [])
    got = mat.epp_map(empty, 60)
    assert (got["score"] == 0).all() and (got["counts"] == 0).all() and len(got["epp_nodes"]) == 0
    # a read that lists the reference base is outside the domain (sam2pb only lists differences)
    bad = w.EppReads.from_lists([[(3, w.A, w.A)]], [1], [10])
    with pytest.raises(Exception):
        mat.epp_map(bad, 60)
    mat.close()


@pytest.mark.parametrize("rpl", [1, 4])
@pytest.mark.parametrize("n_nodes,n_reads,read_len,cap", [(3000, 700, 150, 2048), (20000, 3000, 150, 64),
                                                          (8000, 300, 1200, 2048)])
def test_generated_trees(oracle, n_nodes, n_reads, read_len, cap, rpl, monkeypatch):
    monkeypatch.setenv("WEPP_EPP_RPL", str(rpl))
    # SARS-CoV-2-sized genome, amplicon reads drawn from leaf genotypes (several windows, tiles,
    # chunks); EPP lists capped at `cap` placements
    g = w.generate_tree(5, n_nodes)
    reads = g.reads(6, n_reads, read_len=read_len, amplicon_len=max(400, read_len), amplicon_step=300 if read_len < 400 else 1000,
                    p_substitution=0.003, p_n=0.01, windows=True, max_degree=7)
    mat = w.Mat(g.tree)
    got = mat.epp_map(reads, 29903, max_cached_epp=cap)
    want = oracle.OracleTree(g.tree).epp_map(reads, genome_size=29903)
    # the oracle keeps lists up to 2048; cut them down to `cap` the way the library does
    if cap != 2048:
        keep = want["multiplicity"] <= cap
        off = np.zeros(n_reads + 1, np.uint64)
        nodes = []
        for r in range(n_reads):
            if keep[r]:
                nodes.append(want["epp_nodes"][int(want["epp_off"][r]):int(want["epp_off"][r + 1])])
            off[r + 1] = off[r] + (want["multiplicity"][r] if keep[r] else 0)
        want["epp_off"] = off
        want["epp_nodes"] = np.concatenate(nodes) if nodes else np.zeros(0, np.uint32)
    _check(got, want, n_reads, (n_nodes, n_reads))
    t = w.epp_last_timing()
    assert t["groups"] >= 1 and t["jobs"] >= t["groups"]
    mat.close()


def test_run_to_run_identical():
    g = w.generate_tree(9, 5000)
    reads = g.reads(10, 2000, windows=True, max_degree=3)
    mat = w.Mat(g.tree)
    a = mat.epp_map(reads, 29903)
    b = mat.epp_map(reads, 29903)
    for k in ("max_parsimony", "multiplicity", "epp_nodes", "score", "counts"):
        assert np.array_equal(a[k], b[k]), k
    mat.close()
