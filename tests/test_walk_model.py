"""The per-read walk (position index + range queries, k_walk) as a Python model over the arrays the
flattener builds, against the oracle: whole-tree stream and the crown stream a read is routed to."""
import numpy as np

import fuzz_trees as ft
import sweep_model as sm
import walk_model as wm
import wepp_amd as w


def _cols(S):
    return list(zip(*S)) if S else ([], [], [], [])


def _root_score(fv, S):
    return sm.theta(fv, S) - len(S)


def test_walk_model_matches_oracle_fuzz(oracle):
    rng = np.random.default_rng(12)
    n = exact = segs = by_entry = 0
    deepest = 0
    for it in range(300):
        if it % 3 == 2:
            tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(60, 300)), genome=40, p_masked=0.05, p_ambig=0.15)
            genome = 40
        else:
            tree, ref = ft.random_tree(rng)
            genome = 60
        ot = oracle.OracleTree(tree)
        fv = w.FlatView(tree)
        models = [wm.WalkModel(fv, i) for i in range(fv.n_streams)]
        tiers = sm.TieredModel(fv)
        for _ in range(4):
            S = ft.random_sample(rng, ref, genome=genome, max_k=int(rng.integers(0, 10)))
            o = ot.place_sample(*_cols(S))
            rs = _root_score(fv, S)
            for m in (models[-1], models[tiers.route(S)]):
                got = m.result(S, rs)
                assert got == (o["score"], o["best_j"], o["num_best"], o["has_unique"]), (S, got, o, m.stream)
                deepest = max(deepest, m.max_stack)
                exact += m.n_exact
                segs += m.n_segments
                by_entry += m.n_by_entry
            n += 1
    assert n == 1200 and segs > exact > 0 and deepest >= 2
    assert by_entry > 0          # ranges decided by the entry's own pre-test byte (tests/conftest.py builds it for every stream)


def test_all_pairs_walk_matches_oracle(oracle):
    """The wave-per-read form (k_walk_wave: no walk, an all-pairs pass over the read's entries) gives the walk's
    result, on the whole-tree stream and on the stream the read is routed to; small genomes make the lists long."""
    rng = np.random.default_rng(33)
    n = most = 0
    for it in range(240):
        genome = int(rng.choice([6, 12, 40]))
        tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(30, 260)), genome=genome, p_masked=0.04, p_ambig=0.12, max_muts=3)
        ot = oracle.OracleTree(tree)
        fv = w.FlatView(tree)
        models = [wm.WalkModel(fv, i) for i in range(fv.n_streams)]
        tiers = sm.TieredModel(fv)
        for _ in range(4):
            S = ft.random_sample(rng, ref, genome=genome, max_k=int(rng.integers(0, 8)))
            o = ot.place_sample(*_cols(S))
            rs = _root_score(fv, S)
            for m in (models[-1], models[tiers.route(S)]):
                bs, br, cnt, hu = m.place_all_pairs(S, rs)
                got = (bs, int(m.rank2bfs[br]), cnt, hu)
                assert got == (o["score"], o["best_j"], o["num_best"], o["has_unique"]), (S, got, o, m.stream)
                most = max(most, m.events_of(S))
            n += 1
    assert n == 960 and most > 64


def test_chunked_walk_matches_oracle(oracle):
    """A read's walk cut into C independent jobs (start state from binary searches + the chains of
    enclosing entries), combined: same result for every C the longest list allows."""
    rng = np.random.default_rng(15)
    n = multi = 0
    for it in range(200):
        genome = int(rng.choice([8, 20, 60]))
        tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(40, 400)), genome=genome, p_masked=0.03, p_ambig=0.1,
                                   max_muts=3)
        ot = oracle.OracleTree(tree)
        fv = w.FlatView(tree)
        models = [wm.WalkModel(fv, i) for i in range(fv.n_streams)]
        tiers = sm.TieredModel(fv)
        for _ in range(4):
            S = ft.random_sample(rng, ref, genome=genome, max_k=int(rng.integers(1, 9)))
            o = ot.place_sample(*_cols(S))
            rs = _root_score(fv, S)
            for m in (models[-1], models[tiers.route(S)]):
                npos = len(m.ix_off) - 1
                longest = max([int(m.ix_off[p + 1]) - int(m.ix_off[p]) - 1 for (p, _, _, _) in S if p < npos] + [0])
                for C in sorted({1, 2, 3, min(7, longest), longest}):
                    if C < 1 or (C > 1 and longest < C):
                        continue
                    m.max_stack = 0
                    got = m.result(S, rs, C)
                    assert got == (o["score"], o["best_j"], o["num_best"], o["has_unique"]), (S, C, got, o, m.stream)
                    assert m.max_stack <= m.stack_bound(S)
                    multi += C > 1
            n += 1
    assert n == 800 and multi > 1500


def test_walk_stack_never_exceeds_the_route_bound():
    """k_route sends a read to the walk only if the sum over its positions of `maxnest` fits the stack."""
    rng = np.random.default_rng(13)
    worst = 0
    for _ in range(150):
        tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(30, 200)), genome=12, max_muts=3)
        fv = w.FlatView(tree)
        m = wm.WalkModel(fv)
        for _ in range(4):
            S = ft.random_sample(rng, ref, genome=12, max_k=8)
            m.max_stack = 0
            m.place(S, _root_score(fv, S))
            assert m.max_stack <= m.stack_bound(S), (m.max_stack, m.stack_bound(S), S)
            worst = max(worst, m.max_stack)
    assert worst >= 3


def test_index_layout_invariants():
    g = w.generate_tree(9, 20000)
    fv = w.FlatView(g.tree)
    for s in (0, fv.n_streams - 1):
        head, ent = fv.get("ix_head", s), fv.get("ix_ent", s)
        off, node, end = head[:, 0], ent[:, 0], ent[:, 1]
        n = len(fv.get("nkey", s))
        assert off[0] == 0 and off[-1] == len(node)
        assert (node[off[1:] - 1] == wm.IX_NONE).all()                 # every list ends in its sentinel
        real = node != wm.IX_NONE
        assert (end[real] > node[real]).all() and (end[real] <= n).all()
        # inside a list the node indices ascend
        brk = np.zeros(len(node), bool)
        brk[off[:-1]] = True
        d = np.diff(node.astype(np.int64))
        assert (d[~brk[1:]] > 0).all()
        # pre-test byte of an entry (top byte of its rank): the minimum static score (sp encoding) of the eligible
        # nodes between the list's previous entry and the entry's node
        assert fv.get("ix_pre", s)[0] == 1
        nk, ns_ = fv.get("nkey", s), fv.get("nstat", s)
        val = np.where((ns_ & sm.NS_ELIG0) != 0, np.clip(nk >> 32, 0, wm.SP_CLAMP), wm.SP_NONE).astype(np.int64)
        rng = np.random.default_rng(3)
        for p in rng.choice(len(off) - 1, size=min(60, len(off) - 1), replace=False):
            prev = 0
            for e in range(int(off[p]), int(off[p + 1]) - 1):
                want = int(val[prev:int(node[e])].min()) if int(node[e]) > prev else wm.SP_NONE
                assert int(ent[e, 6]) >> 24 == want, (s, p, e)
                prev = int(node[e]) + 1
        sp, pre, suf, dst = fv.get("sp", s), fv.get("rq_pre", s), fv.get("rq_suf", s), fv.get("rq_dst", s)
        assert len(sp) % n == 0 and len(pre) == len(suf) == n and len(dst) % ((n + 15) // 16) == 0
        elig = (fv.get("nstat", s) & sm.NS_ELIG0) != 0
        nb = (n + 15) // 16
        has = np.add.reduceat(elig.astype(np.int64), np.arange(0, n, 16)) > 0
        assert ((dst[:nb, 2] > 0) == has).all()          # row 0 = the blocks: a count iff the block holds an eligible node


def test_range_queries_against_brute_force():
    """sparse table and block / disjoint-sparse-table query against a scan, random ranges."""
    rng = np.random.default_rng(14)
    g = w.generate_tree(10, 5000, genome_len=800)
    fv = w.FlatView(g.tree)
    for s in range(fv.n_streams):
        m = wm.WalkModel(fv, s)
        base = (m.nkey >> 32).astype(np.int64)
        rank = (m.nkey & 0xFFFFFFFF).astype(np.int64)
        elig = (m.nstat & sm.NS_ELIG0) != 0
        for _ in range(300):
            a = int(rng.integers(0, m.n))
            b = int(rng.integers(a + 1, min(m.n, a + int(rng.choice([3, 20, 200, 5000]))) + 1))
            sel = np.nonzero(elig[a:b])[0] + a
            if len(sel) == 0:
                assert m.range_exact(a, b)[2] == 0
                continue
            mn = int(base[sel].min())
            at = sel[base[sel] == mn]
            # the pre-test reads the minimum of a superset of the range at most twice as long
            ln = b - a
            sup = np.nonzero(elig[a:min(m.n, a + (1 << (ln - 1).bit_length()))])[0] + a
            assert m.range_min(a, b) == min(int(base[sup].min()), wm.SP_CLAMP) <= min(mn, wm.SP_CLAMP)
            assert m.range_exact(a, b)[:3] == (mn, int(rank[at].min()), len(at))


def test_window_crowns_hold_every_node_that_can_win(oracle):
    """Window crowns (flatmat.hpp: wcrowns): the walk of a read confined to a genome window, run on the crown of its
    window whose bound is the read's ROOT score, gives the oracle's placement (score, node, number of optimal nodes,
    has_unique) -- on trees whose mutations crowd a few windows (long genome, 1024-position stride), with reads that
    list up to 16 positions of one window, most of them N; whole and cut into jobs."""
    rng = np.random.default_rng(77)
    n = covered = n_last = 0
    levels = set()
    for it in range(6):
        g = w.generate_tree(100 + it, 4000, genome_len=5000, p_ambiguous=0.02, p_masked_node=0.01, root_mutations=it % 2,
                            p_back_mutation=0.1)
        ot = oracle.OracleTree(g.tree)
        fv = w.FlatView(g.tree)
        tau = fv.get("wc_tau").view(np.int32).reshape(-1, w.WINDOW_CROWN_LEVELS)
        nodes = fv.get("wc_nodes").reshape(-1, w.WINDOW_CROWN_LEVELS)
        assert tau.shape[0] >= 4 and (nodes[:, 0] > 0).all()
        assert fv.stats.n_window_crowns == int((nodes > 0).sum()) and fv.stats.window_crown_nodes == int(nodes.sum())
        reads = g.reads(200 + it, 160, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.01, p_n=0.04, p_iupac=0.2)
        # ... and long reads full of substitutions: a root score beyond every bounded crown, so the window's LAST crown
        # (all the nodes any read of the window can be placed on: out_w - in_w <= base(root), flatmat.cpp)
        long_reads = g.reads(300 + it, 60, read_len=900, amplicon_len=1000, amplicon_step=700, p_substitution=0.04, p_n=0.02, p_iupac=0.1)
        models = {}
        for q in range(reads.n_reads + long_reads.n_reads):
            pos, ref, mut, miss = reads.entries(q) if q < reads.n_reads else long_reads.entries(q - reads.n_reads)
            S = list(zip(pos.tolist(), ref.tolist(), mut.tolist(), miss.tolist()))
            if not S or (len(S) > 16 and q < reads.n_reads):
                continue
            wi = S[0][0] // 1024
            if wi >= tau.shape[0] or S[-1][0] >= wi * 1024 + 2560:
                continue
            rs = _root_score(fv, S)
            ci = next((i for i in range(w.WINDOW_CROWN_LEVELS) if nodes[wi, i] and rs <= tau[wi, i]), None)
            n += 1
            if ci is None:
                continue
            m = models.get((wi, ci)) or models.setdefault((wi, ci), wm.WalkModel(fv, f"c{wi}.{ci}"))
            o = ot.place_sample(pos, ref, mut, miss)
            want = (o["score"], o["best_j"], o["num_best"], o["has_unique"])
            assert m.result(S, rs) == want, (it, q, wi, ci, S)
            npos = len(m.ix_off) - 1
            longest = max([int(m.ix_off[p + 1]) - int(m.ix_off[p]) - 1 for (p, _, _, _) in S if p < npos] + [0])
            if longest >= 2:
                assert m.result(S, rs, 2) == want, (it, q, "two jobs")
            assert m.n < len(fv.get("nkey"))          # a crown, not the tree
            covered += 1
            levels.add(ci)
            n_last += int(tau[wi, ci] == 0x7FFFFFFF and len(S) > 16)
    assert covered > 500 and covered > 0.6 * n and len(levels) >= 3 and n_last > 30
