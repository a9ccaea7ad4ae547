"""CPU tests of the host logic: flattener (orders, parent alleles, per-node
constants, event stream, checkpoints, pruning bound) through the Python model
of the sweep, against the oracle; generators; error behaviour."""
import ctypes
import hashlib

import numpy as np
import pytest

import fuzz_trees as ft
import sweep_model as sm
import wepp_amd as w
from wepp_amd import A, C, G, T, N, Tree


def _cols(S):
    return list(zip(*S)) if S else ([], [], [], [])


def test_orders_and_leaf_counts_match_oracle(oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        tree, _ = ft.random_tree(rng)
        ot = oracle.OracleTree(tree)
        fv = w.FlatView(tree)
        assert (fv.get("bfs2id") == ot.bfs_ids()).all()
        assert (fv.get("dfs2id") == ot.dfs_ids()).all()
        nl = ot.num_leaves()
        assert (fv.get("num_leaves") == nl[fv.get("dfs2id")]).all()


def test_model_of_sweep_matches_oracle_fuzz(oracle):
    """Flattener + closed form (incl. chunk checkpoints and the pruning bound,
    asserted inside the model) == oracle, per-node scores included."""
    rng = np.random.default_rng(2)
    n = 0
    for _ in range(250):
        tree, ref = ft.random_tree(rng)
        ot = oracle.OracleTree(tree)
        fv = w.FlatView(tree)
        fm = sm.FlatModel(fv)
        d2b = fv.get("dfs2bfs")
        for _ in range(4):
            S = ft.random_sample(rng, ref)
            o = ot.place_sample(*_cols(S))
            p = ot.place_sample(*_cols(S), per_node_scores=True)
            ns = np.zeros(tree.n_nodes, np.int64)
            fm.place(S, node_scores=ns)
            ns_bfs = np.zeros_like(ns)
            ns_bfs[d2b] = ns
            assert (ns_bfs == p["node_scores"]).all()
            got = fm.place_full(S, nchunks=int(rng.integers(1, 6)))
            assert (got["score"], got["num_best"], got["best_j"], got["has_unique"]) == \
                (o["score"], o["num_best"], o["best_j"], o["has_unique"])
            n += 1
    assert n == 1000


def test_model_matches_oracle_on_generated_tree(oracle):
    g = w.generate_tree(5, 3000, genome_len=1500, p_ambiguous=0.01, p_masked_node=0.003, root_mutations=2)
    reads = g.reads(6, 60, p_substitution=0.004, p_n=0.01, p_iupac=0.2)
    ot = oracle.OracleTree(g.tree)
    want = ot.place_batch(reads, 4)
    fm = sm.FlatModel(w.FlatView(g.tree))
    for r in range(reads.n_reads):
        p, rf, a, ms = reads.entries(r)
        S = [(int(p[i]), int(rf[i]), int(a[i]), int(ms[i])) for i in range(len(p))]
        got = fm.place_full(S, nchunks=1 + r % 4)
        assert (got["score"], got["num_best"], got["best_j"], got["has_unique"]) == \
            (want["score"][r], want["num_best"][r], want["best_j"][r], want["has_unique"][r])


def test_checkpoints_reproduce_running_state():
    rng = np.random.default_rng(3)
    checked = 0
    for _ in range(120):
        tree, ref = ft.random_tree(rng, n_nodes=64, max_muts=6)
        fm = sm.FlatModel(w.FlatView(tree))
        if fm.NB < 2:
            continue
        for _ in range(3):
            S = ft.random_sample(rng, ref)
            tr = {}
            fm.place(S, trace_c=tr)
            for b in range(0, fm.NB, fm.cp_stride):
                assert fm.chunk_start_c(S, b) == tr[b]
                checked += 1
    assert checked > 50


def test_stream_layout_invariants():
    g = w.generate_tree(9, 20000)
    fv = w.FlatView(g.tree)
    n0, eo = fv.get("blk_node0"), fv.get("blk_eoff")
    assert n0[0] == 0 and n0[-1] == 20000 and (np.diff(n0) >= 1).all() and (np.diff(n0) <= 64).all()
    assert (eo % 2 == 0).all() and (np.diff(eo.astype(np.int64)) >= 0).all()
    one_node = np.diff(n0) == 1
    assert (np.diff(eo.astype(np.int64))[~one_node] <= 128).all()
    ev = fv.get("ev_word")
    assert len(ev) == eo[-1] == fv.stats.n_events
    words = fv.get("words")
    real = ev[(ev & 0xFFFFF) != 0xFFFFF]
    # every mutation word appears once as an enter event; exits only for internal nodes
    enters = real[(real >> 30) & 1 == 0]
    assert sorted((enters & 0x3FFFFFFF).tolist()) == sorted(words.tolist())
    # ranks are a permutation and the root (most leaves) ranks first
    r2d = fv.get("rank2dfs")
    assert sorted(r2d.tolist()) == list(range(20000)) and r2d[0] == 0


def test_generators_are_deterministic():
    def digest(seed):
        g = w.generate_tree(seed, 5000, p_ambiguous=0.01, p_masked_node=0.001)
        r = g.reads(seed + 1, 2000, p_iupac=0.05)
        h = hashlib.sha256()
        for a in (g.tree.parent, g.tree.mut_off, g.tree.mut_pos, g.tree.mut_ref, g.tree.mut_mut, r.read_off,
                  r.read_word):
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()
    assert digest(3) == digest(3)
    assert digest(3) != digest(4)
    # reads are sorted by position with unique positions (precondition of usher_mapper.cpp:205-243)
    r = w.generate_tree(3, 5000).reads(4, 3000, read_len=1200, amplicon_len=1200, amplicon_step=1000,
                                       p_substitution=0.03, p_n=0.02)
    pos = (r.read_word & 0xFFFFF).astype(np.int64)
    for q in range(r.n_reads):
        p = pos[r.read_off[q]:r.read_off[q + 1]]
        assert (np.diff(p) > 0).all()


@pytest.mark.parametrize("mutate,msg", [
    (lambda t: t.parent.__setitem__(0, 1), "root"),                    # no root / cycle
    (lambda t: t.parent.__setitem__(3, -1), "more than one root"),
    (lambda t: t.parent.__setitem__(2, 99), "out of range"),
])
def test_flatten_rejects_malformed_trees(mutate, msg):
    t = Tree.from_lists([-1, 0, 0, 1, 1], [[], [(10, A, A, C)], [(20, G, G, T)], [], [(30, C, C, A)]])
    mutate(t)
    with pytest.raises(w.WeppError) as ei:
        w.FlatView(t)
    assert ei.value.code == 1 and msg in str(ei.value)


def test_flatten_rejects_unsorted_duplicate_and_bad_ref():
    with pytest.raises(w.WeppError, match="not sorted"):
        w.FlatView(Tree.from_lists([-1, 0], [[], [(20, G, G, T), (10, A, A, C)]]))
    with pytest.raises(w.WeppError, match="duplicate"):
        w.FlatView(Tree.from_lists([-1, 0], [[], [(10, A, A, C), (10, A, A, G)]]))
    with pytest.raises(w.WeppError, match="single nucleotide"):
        w.FlatView(Tree.from_lists([-1, 0], [[], [(10, A | C, A, G)]]))
    with pytest.raises(w.WeppError, match="inconsistent ref_nuc"):
        w.FlatView(Tree.from_lists([-1, 0, 0], [[], [(10, A, A, G)], [(10, C, C, G)]]))
    with pytest.raises(w.WeppError, match="exceeds"):
        w.FlatView(Tree.from_lists([-1, 0], [[], [(0xFFFFF, A, A, G)]]))


def test_true_parent_allele_is_recomputed():
    """Mutation::par_nuc is never read by the scorer (usher_mapper.cpp only
    copies it); a tree with garbage par_nuc must flatten to the same words."""
    good = Tree.from_lists([-1, 0, 1], [[(10, A, A, C)], [(10, A, C, G)], [(10, A, G, A)]])
    bad = Tree.from_lists([-1, 0, 1], [[(10, A, T, C)], [(10, A, A, G)], [(10, A, T, A)]])
    assert (w.FlatView(good).get("words") == w.FlatView(bad).get("words")).all()
    assert ((w.FlatView(good).get("words") >> 22) & 15).tolist() == [0, C, G]


def test_crown_streams_give_the_same_answer_as_the_whole_tree(oracle):
    """Work skipping: a read routed to a crown stream (k_route's theta bound)
    must be placed exactly as on the whole-tree stream and as by the oracle."""
    rng = np.random.default_rng(4)
    routed = np.zeros(16, int)
    for _ in range(120):
        tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(1, 400)), genome=int(rng.choice([60, 200, 1000])),
                                   max_muts=int(rng.choice([2, 4])))
        ot = oracle.OracleTree(tree)
        fv = w.FlatView(tree)
        tm = sm.TieredModel(fv)
        for _ in range(4):
            S = ft.random_sample(rng, ref, genome=max(ref), max_k=int(rng.choice([2, 4, 7])))
            o = ot.place_sample(*_cols(S))
            routed[tm.route(S)] += 1
            got = tm.place_full(S, nchunks=int(rng.integers(1, 4)))
            assert (got["score"], got["num_best"], got["best_j"], got["has_unique"]) == \
                (o["score"], o["num_best"], o["best_j"], o["has_unique"])
    assert (routed[:3] > 20).all(), routed


def test_crown_is_ancestor_closed_and_covers_low_scores():
    g = w.generate_tree(13, 50000)
    fv = w.FlatView(g.tree)
    assert fv.n_streams >= 3 and fv.stats.stream_tau[fv.n_streams - 1] == 2**31 - 1
    base = (fv.get("nkey") >> 32)
    for i in range(fv.n_streams - 1):
        tau = fv.stats.stream_tau[i]
        keys = fv.get("nkey", i)
        # every node with base <= tau is in the crown (matched through its unique global rank)
        want = set((fv.get("nkey")[base <= tau] & 0xFFFFFFFF).tolist())
        have = set((keys & 0xFFFFFFFF).tolist())
        assert want <= have
        assert len(have) == fv.stats.stream_nodes[i] < 50000


def test_window_streams_match_oracle(oracle):
    """The whole tree as the reads of one genome window see it (touched nodes + pseudo-nodes for the runs in
    between): the sweep model on a window stream against the oracle, for reads whose positions lie inside."""
    from wepp_amd import _lib
    rng = np.random.default_rng(21)
    checked = 0
    kinds = {True: 0, False: 0}
    for it in range(12):
        g = w.generate_tree(100 + it, int(rng.integers(300, 3000)), genome_len=5000, p_ambiguous=0.02, p_masked_node=0.01,
                            root_mutations=int(rng.integers(0, 3)))
        fv = w.FlatView(g.tree)
        ot = oracle.OracleTree(g.tree)
        reads = g.reads(200 + it, 40, read_len=900, amplicon_len=900, amplicon_step=700, p_substitution=0.02, p_n=0.01, p_iupac=0.1)
        want = ot.place_batch(reads, 4)
        models = {}
        for r in range(reads.n_reads):
            p, rf, a, ms = reads.entries(r)
            if len(p) == 0:
                continue
            wi = int(p.min()) // 1024
            if int(p.max()) >= wi * 1024 + 2560:
                continue
            if wi not in models:
                models[wi] = sm.FlatModel(fv, "w%d" % wi)
            S = [(int(p[i]), int(rf[i]), int(a[i]), int(ms[i])) for i in range(len(p))]
            got = models[wi].place_full(S, nchunks=1 + r % 3)
            assert (got["score"], got["num_best"], got["best_j"], got["has_unique"]) == \
                (want["score"][r], want["num_best"][r], want["best_j"][r], want["has_unique"][r]), (it, r, wi)
            checked += 1
        for wi, m in models.items():
            assert m.N < g.tree.n_nodes or g.tree.n_nodes < 50        # fewer elements than nodes
            kinds[len(m.ncnt) == 0] += 1
    # both kinds of window stream were swept: the window's candidate crown (real nodes only), and the whole tree with
    # pseudo-nodes where the candidates were too many to be worth a crown
    assert checked > 300 and kinds[True] >= 5 and kinds[False] >= 5, kinds


def test_flat_image_round_trips_through_a_file(tmp_path):
    """wepp_flat_save / wepp_flat_load (one flatten per node when the ranks are processes): every array of the image
    comes back byte for byte; a damaged or foreign file is refused."""
    g = w.generate_tree(17, 30000, genome_len=6000, p_ambiguous=0.02, p_masked_node=0.004, root_mutations=2)
    a = w.FlatView(g.tree)
    path = str(tmp_path / "image.bin")
    a.save(path)
    b = w.FlatView.load(path)
    assert bytes(a.stats) == bytes(b.stats) and a.cp_stride == b.cp_stride
    for name in ("node_woff", "words", "rank2dfs", "dfs2bfs", "rank2bfs", "bfs2id", "dfs2id", "parent_dfs", "dfs_end", "num_leaves",
                 "maxnest", "epp_word", "epp_node", "seed_sig", "wc_tau", "wc_nodes"):
        assert np.array_equal(a.get(name), b.get(name)), name
    fields = ("nkey", "nstat", "blk_node0", "blk_eoff", "blk_sum", "ev_word", "ev_meta", "ev_lb", "cp_off", "cp_word", "ix_head", "ix_ent",
              "ix_nest", "nrec", "rq_pre", "rq_suf", "rq_dst", "sp")
    for st in list(range(a.n_streams)) + ["w0", "w2", "c1.0"]:
        for name in fields:
            assert np.array_equal(a.get(name, stream=st), b.get(name, stream=st)), (st, name)
    raw = open(path, "rb").read()
    bad = str(tmp_path / "bad.bin")
    open(bad, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(w.WeppError):
        w.FlatView.load(bad)
    open(bad, "wb").write(b"not an image" + raw[12:])
    with pytest.raises(w.WeppError):
        w.FlatView.load(bad)
