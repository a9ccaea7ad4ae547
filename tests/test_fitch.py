"""Per-site Fitch-Sankoff (mapper_body, usher_mapper.cpp:7-162): oracle sanity on
the CPU, GPU kernels vs oracle through the C-ABI."""
import numpy as np
import pytest

import fuzz_trees as ft
import wepp_amd as w
from wepp_amd import A, C, G, T, Tree


def _random_rows(rng, tree, n_rows, p_var=0.15, p_internal=0.02, p_amb=0.1):
    n = tree.n_nodes
    has_child = np.zeros(n, bool)
    for p in tree.parent:
        if p >= 0:
            has_child[p] = True
    site_ref, var_off, var_node, var_nuc = [], [0], [], []
    for _ in range(n_rows):
        ref = 1 << int(rng.integers(0, 4))
        site_ref.append(ref)
        for i in range(n):
            pr = p_internal if has_child[i] else p_var
            if rng.random() < pr:
                nuc = 1 << int(rng.integers(0, 4))
                if rng.random() < p_amb:
                    nuc |= 1 << int(rng.integers(0, 4))
                var_node.append(i)
                var_nuc.append(nuc)
        var_off.append(len(var_node))
    return (np.array(site_ref, np.uint8), np.array(var_off, np.uint32), np.array(var_node, np.uint32),
            np.array(var_nuc, np.uint8))


def test_oracle_mapper_body_hand_cases(oracle):
    # ((A,B),(C,D)); site with ref A
    t = Tree.from_lists([-1, 0, 0, 1, 1, 2, 2], [[] for _ in range(7)])
    ot = oracle.OracleTree(t)
    assert ot.mapper_body(A, [], []) == []                                  # nobody differs: no mutation
    assert ot.mapper_body(A, [3], [G]) == [(3, A, G)]                       # one leaf: mutation on the leaf
    assert ot.mapper_body(A, [3, 4], [G, G]) == [(1, A, G)]                 # both leaves of a cherry: on their parent
    # three of four leaves G: G wins at the root, D (=6) mutates back to the reference base
    assert ot.mapper_body(A, [3, 4, 5], [G, G, G]) == [(0, A, G), (6, G, A)]
    # an ambiguous leaf (A or G) next to a G leaf resolves to G on the parent
    assert ot.mapper_body(A, [3, 4], [G, A | G]) == [(1, A, G)]


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["levels", "sets", "scores"])
@pytest.mark.parametrize("chunks", [None, 2, 5, 17])
def test_fitch_kernels_vs_oracle(oracle, chunks, form, monkeypatch):
    """chunks: how many waves share one walk of the tree (None = the library's choice, 1 for
    trees this small); nodes spanning chunk boundaries go through the stitch kernel.
    form: "levels" = level-synchronous kernels on optimal sets (the default whenever every observed
    allele set is non-empty; `chunks` does not apply), "sets" = the same arithmetic on the DFS
    stack, "scores" = the four integer scores on the DFS stack (the general form)."""
    if chunks:
        if form == "levels":
            pytest.skip("chunks only exist in the DFS stack forms")
        monkeypatch.setenv("WEPP_FITCH_CHUNKS", str(chunks))
    if form != "levels":
        monkeypatch.setenv("WEPP_FITCH_DFS", "1")
    if form == "scores":
        monkeypatch.setenv("WEPP_FITCH_SCORES", "1")
    rng = np.random.default_rng(2024)
    total = 0
    for it in range(30):
        tree, _ = ft.random_tree(rng, n_nodes=int(rng.integers(1, 300)), max_muts=0, root_muts=False, p_masked=0.0,
                                 p_root_masked=0.0)
        n_rows = int(rng.integers(1, 200))
        site_ref, var_off, var_node, var_nuc = _random_rows(rng, tree, n_rows)
        s, nd, par, mut = w.fitch_sites(tree, site_ref, var_off, var_node, var_nuc)
        ot = oracle.OracleTree(tree)
        k = 0
        for r in range(n_rows):
            a, b = int(var_off[r]), int(var_off[r + 1])
            want = ot.mapper_body(int(site_ref[r]), var_node[a:b].astype(np.int32), var_nuc[a:b])
            got = [(int(nd[i]), int(par[i]), int(mut[i])) for i in range(k, k + len(want))]
            assert (s[k:k + len(want)] == r).all() and got == want, (it, r)
            k += len(want)
        assert k == len(s)
        total += k
    assert total > 1000


@pytest.mark.gpu
def test_fitch_rebuilds_the_mutations_of_a_generated_tree(oracle):
    """Round trip: take a generated MAT, turn every leaf genotype into VCF rows, run
    Fitch-Sankoff on the bare topology -- the result must equal the oracle's on every row."""
    g = w.generate_tree(71, 4000, genome_len=400)
    tree = g.tree
    n = tree.n_nodes
    has_child = np.zeros(n, bool)
    has_child[tree.parent[tree.parent >= 0]] = True
    # genotype of every leaf: walk up to the root
    geno = {}
    for leaf in np.flatnonzero(~has_child):
        gt = {}
        i = int(leaf)
        while i >= 0:
            for k in range(int(tree.mut_off[i]), int(tree.mut_off[i + 1])):
                gt.setdefault(int(tree.mut_pos[k]), (int(tree.mut_ref[k]), int(tree.mut_mut[k])))
            i = int(tree.parent[i])
        geno[int(leaf)] = gt
    sites = sorted({p for gt in geno.values() for p in gt})
    ref_of = {}
    for gt in geno.values():
        for p, (r, m) in gt.items():
            ref_of[p] = r
    site_ref, var_off, var_node, var_nuc = [], [0], [], []
    for p in sites:
        site_ref.append(ref_of[p])
        for leaf, gt in geno.items():
            if p in gt and gt[p][1] != ref_of[p]:
                var_node.append(leaf)
                var_nuc.append(gt[p][1])
        var_off.append(len(var_node))
    bare = Tree(tree.parent, np.zeros(n + 1, np.uint32), [], [], [])
    s, nd, par, mut = w.fitch_sites(bare, np.array(site_ref, np.uint8), np.array(var_off, np.uint32),
                                    np.array(var_node, np.uint32), np.array(var_nuc, np.uint8))
    ot = oracle.OracleTree(bare)
    k = 0
    for r in range(len(sites)):
        a, b = var_off[r], var_off[r + 1]
        want = ot.mapper_body(site_ref[r], np.array(var_node[a:b], np.int32), np.array(var_nuc[a:b], np.uint8))
        got = [(int(nd[i]), int(par[i]), int(mut[i])) for i in range(k, k + len(want))]
        assert got == want and (s[k:k + len(want)] == r).all(), r
        k += len(want)
    assert k == len(s) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", [None, 3])
def test_fitch_empty_allele_sets_take_the_score_form(oracle, chunks, monkeypatch):
    """A row that observes an EMPTY allele set on some node (mask 0: every base costs num_nodes,
    usher_mapper.cpp:57-62) drives sums past num_nodes, where the clamp of :98 binds: such calls
    must fall back to the integer-score kernels and still equal the oracle."""
    if chunks:
        monkeypatch.setenv("WEPP_FITCH_CHUNKS", str(chunks))
    rng = np.random.default_rng(77)
    for it in range(12):
        tree, _ = ft.random_tree(rng, n_nodes=int(rng.integers(3, 120)), max_muts=0, root_muts=False, p_masked=0.0,
                                 p_root_masked=0.0)
        n_rows = int(rng.integers(1, 70))
        site_ref, var_off, var_node, var_nuc = _random_rows(rng, tree, n_rows, p_var=0.3, p_internal=0.1)
        if len(var_nuc) == 0:
            continue
        zero = rng.random(len(var_nuc)) < 0.3
        zero[int(rng.integers(0, len(var_nuc)))] = True
        var_nuc = np.where(zero, 0, var_nuc).astype(np.uint8)
        s, nd, par, mut = w.fitch_sites(tree, site_ref, var_off, var_node, var_nuc)
        ot = oracle.OracleTree(tree)
        k = 0
        for r in range(n_rows):
            a, b = int(var_off[r]), int(var_off[r + 1])
            want = ot.mapper_body(int(site_ref[r]), var_node[a:b].astype(np.int32), var_nuc[a:b])
            got = [(int(nd[i]), int(par[i]), int(mut[i])) for i in range(k, k + len(want))]
            assert (s[k:k + len(want)] == r).all() and got == want, (it, r)
            k += len(want)
        assert k == len(s)


@pytest.mark.gpu
def test_fitch_argument_errors():
    t = Tree.from_lists([-1, 0, 0], [[], [], []])
    with pytest.raises(w.WeppError) as ei:
        w.fitch_sites(t, [A | C], [0, 0], [], [])
    assert ei.value.code == 1
    with pytest.raises(w.WeppError):
        w.fitch_sites(t, [A], [0, 1], [7], [G])


@pytest.mark.gpu
def test_fitch_plan_reuse_vs_oracle(oracle):
    """wepp_fitch_plan_*: the tree-dependent work done once, several batches of rows on the same plan
    (what read_vcf does row after row), each equal to the oracle and to the one-shot call."""
    rng = np.random.default_rng(41)
    tree, _ = ft.random_tree(rng, n_nodes=180, genome=30)
    bare = Tree(tree.parent, np.zeros(tree.n_nodes + 1, np.uint32), [], [], [])
    ot = oracle.OracleTree(bare)
    plan = w.FitchPlan(bare)
    for n_rows in (3, 70, 1, 300):
        site_ref, var_off, var_node, var_nuc = _random_rows(rng, bare, n_rows)
        got = plan.run(site_ref, var_off, var_node, var_nuc)
        one = w.fitch_sites(bare, site_ref, var_off, var_node, var_nuc)
        assert all((a == b).all() for a, b in zip(got, one))
        k = 0
        for r in range(n_rows):
            x, y = int(var_off[r]), int(var_off[r + 1])
            want = ot.mapper_body(int(site_ref[r]), var_node[x:y].astype(np.int32), var_nuc[x:y])
            rows = [(int(got[1][i]), int(got[2][i]), int(got[3][i])) for i in range(k, k + len(want))]
            assert rows == want and (got[0][k:k + len(want)] == r).all()
            k += len(want)
        assert k == len(got[0])
        t = w.fitch_last_timing()
        assert t["kernels_ms"] > 0
    plan.close()
