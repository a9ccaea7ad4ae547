"""WEPP-native read placement (src/WEPP/initial_filter.cpp:41-239): the oracle's literal
restatement of single_read_tree / cartesian_map against hand cases and against the reference's
second formulation of the same distance, haplotype::mutation_distance (haplotype.hpp:123-173)."""
import numpy as np

import epp_fuzz
import fuzz_trees as ft
from wepp_amd import A, C, G, T, N, EppReads, Tree, unpack_read_word


def _hand_tree():
    # pre-order == id order: 0 root; 1 (10 A>G); 2 = child of 1 (20 C>T); 3 = child of 1 (10 G>A, back to ref);
    # 4 = child of root (30 G>T)
    return Tree.from_lists([-1, 0, 1, 1, 0], [[], [(10, A, A, G)], [(20, C, C, T)], [(10, A, G, A)], [(30, G, G, T)]])


def test_hand_cases(oracle):
    ot = oracle.OracleTree(_hand_tree())
    reads = EppReads.from_lists(
        [[(10, A, G)],                 # window 5..25 carrying 10 A>G: nodes 1 (0 mismatches); 2 has 20 C>T -> 1
         [],                           # window 5..25, reference everywhere: root, 3 (back-mutated), 4 (30 outside)
         [(10, A, N, 1)],              # N at 10 matches anything: root, 1, 3, 4 (2 still has 20 C>T)
         [(30, G, C)],                 # 28..35: nobody has C at 30; everybody but 4 mismatches once (4: T != C too)
         ],
        start=[5, 5, 5, 28], end=[25, 25, 25, 35], degree=[2, 1, 3, 1])
    out = ot.epp_map(reads, genome_size=100)
    assert out["max_parsimony"].tolist() == [0, 0, 0, 1]
    assert out["multiplicity"].tolist() == [1, 3, 4, 5]
    lists = [out["epp_nodes"][int(out["epp_off"][r]):int(out["epp_off"][r + 1])].tolist() for r in range(4)]
    assert lists == [[1], [0, 3, 4], [0, 1, 3, 4], [0, 1, 2, 3, 4]]
    # node_score = degree / ((1 + parsimony) * epps), initial_filter.hpp:54-57
    want = np.zeros(5)
    want[1] += 2 / 1
    for h in (0, 3, 4):
        want[h] += 1 / 3
    for h in (0, 1, 3, 4):
        want[h] += 3 / 4
    for h in range(5):
        want[h] += 1 / (2 * 5)
    assert np.allclose(out["score"], want, rtol=1e-12)
    # bins of genome/50 = 2 positions: starts 5 -> bin 2, 28 -> bin 14
    assert out["counts"][1, 2] == 2 + 3 and out["counts"][0, 2] == 1 + 3 and out["counts"][2, 14] == 1
    assert out["counts"].sum() == 2 * 1 + 1 * 3 + 3 * 4 + 1 * 5


def test_single_read_tree_equals_mutation_distance(oracle):
    """The incremental walk over the range trees and the closed-form distance over stack_muts
    are the reference's two statements of one quantity: same minimum, same EPP set."""
    rng = np.random.default_rng(77)
    checked = 0
    for it in range(60):
        genome = 60
        tree, ref = ft.random_tree(rng, genome=genome, p_masked=0.0, p_root_masked=0.0)
        reads = epp_fuzz.random_epp_reads(rng, tree, ref, genome, n_reads=int(rng.integers(1, 40)))
        ot = oracle.OracleTree(tree)
        out = ot.epp_map(reads, genome_size=genome)
        for r in range(reads.n_reads):
            pos, rf, mu, _ = reads.entries(r)
            d = ot.epp_distance(pos, rf, mu, reads.start[r], reads.end[r])
            m = int(d.min())
            epp = np.flatnonzero(d == m)
            assert out["max_parsimony"][r] == m, (it, r)
            assert out["multiplicity"][r] == len(epp), (it, r)
            got = out["epp_nodes"][int(out["epp_off"][r]):int(out["epp_off"][r + 1])]
            assert got.tolist() == epp.tolist(), (it, r)
            checked += 1
    assert checked > 500


def test_flat_event_stream_model_equals_oracle(oracle):
    """The flattener's EPP event stream + the closed form the kernels use, run as a Python model,
    reproduce the oracle on random trees (back-mutations, ambiguous alleles, masked mutations)."""
    import epp_model
    import wepp_amd as w
    rng = np.random.default_rng(4242)
    for it in range(40):
        genome = 60
        tree, ref = ft.random_tree(rng, genome=genome)
        reads = epp_fuzz.random_epp_reads(rng, tree, ref, genome, n_reads=int(rng.integers(1, 25)))
        fv = w.FlatView(tree)
        ew, en = fv.get("epp_word"), fv.get("epp_node")
        assert len(ew) == len(en) and (np.diff(en.astype(np.int64)) >= 0).all()
        got = epp_model.epp_map(ew, en, tree.n_nodes, reads, genome)
        want = oracle.OracleTree(tree).epp_map(reads, genome_size=genome)
        assert (got["max_parsimony"] == want["max_parsimony"]).all(), it
        assert (got["multiplicity"] == want["multiplicity"]).all(), it
        for r in range(reads.n_reads):
            wl = want["epp_nodes"][int(want["epp_off"][r]):int(want["epp_off"][r + 1])]
            assert got["lists"][r].tolist() == wl.tolist(), (it, r)
        assert np.allclose(got["score"], want["score"], rtol=1e-12, atol=1e-15)
        assert (got["counts"] == want["counts"]).all()
        fv.close()
