"""Parity tests proper: the HIP path, called through the C-ABI, against the
oracle on the same inputs (bit-exact: integer work), against the committed
golden fixtures, and -- at BASELINE.json's full size -- through
size-independent properties."""
import json
import os

import numpy as np
import pytest

import fuzz_trees as ft
import wepp_amd as w
from wepp_amd import A, C, G, T, N, Reads, Tree

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def assert_same(res, want, ctx=""):
    for name, got, exp in (("score", res.score, want["score"]), ("best_bfs_j", res.best_bfs_j, want["best_j"]),
                           ("num_best", res.num_best, want["num_best"]),
                           ("has_unique", res.has_unique, want["has_unique"])):
        bad = np.flatnonzero(np.asarray(got) != np.asarray(exp))
        assert bad.size == 0, f"{ctx}: {name} differs at reads {bad[:10].tolist()} (gpu {np.asarray(got)[bad[:5]]}, oracle {np.asarray(exp)[bad[:5]]})"


def test_golden_fixtures_through_the_c_abi():
    cases = json.load(open(os.path.join(HERE, "golden", "mapper2_cases.json")))["cases"]
    for case in cases:
        t = case["tree"]
        tree = Tree(t["parent"], t["mut_off"], t["mut_pos"], t["mut_ref"], t["mut_mut"], t["mut_par"])
        reads = Reads.from_lists([r["sample"] for r in case["results"]])
        mat = w.Mat(tree)
        assert mat.bfs_order().tolist() == case["bfs_ids"]
        res = mat.place_batch(reads, per_node_scores=True)
        want = {k: np.array([r[k] for r in case["results"]]) for k in ("score", "num_best", "best_j", "has_unique")}
        assert_same(res, want, case["name"])
        # -p mode: every node's value in BFS order (incl. the +1 of non-competing nodes)
        assert res.per_node_scores.tolist() == [r["node_scores"] for r in case["results"]], case["name"]
        mat.close()


@pytest.mark.parametrize("tile", [1, 5, 64])
def test_fuzz_trees_vs_oracle(oracle, tile):
    rng = np.random.default_rng(100 + tile)
    for it in range(60):
        tree, ref = ft.random_tree(rng)
        reads = ft.reads_from_samples([ft.random_sample(rng, ref) for _ in range(int(rng.integers(1, 150)))])
        mat = w.Mat(tree)
        mat.set_tile_reads(tile)
        res = mat.place_batch(reads)
        assert_same(res, oracle.OracleTree(tree).place_batch(reads, 8), f"fuzz {it} tile {tile}")
        mat.close()


@pytest.mark.parametrize("crowns", [True, False])
def test_duplicate_reads_and_sorted_batches(oracle, crowns):
    """Batches with many word-for-word equal reads (one node-by-node evaluation serves all the equal
    reads of a tile) in shuffled and in position-sorted order, on trees small enough for ties; with
    >= 4096 reads on the whole-tree stream the library sorts them by first position itself."""
    rng = np.random.default_rng(4242 + int(crowns))
    for it in range(6):
        tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(30, 400)), genome=80)
        base = [ft.random_sample(rng, ref, genome=80, max_k=int(rng.integers(1, 7))) for _ in range(60)]
        samples = []
        for smp in base:
            samples += [smp] * int(rng.integers(70, 200))
        order = rng.permutation(len(samples))
        shuffled = [samples[i] for i in order]
        assert len(shuffled) >= 4096
        by_pos = sorted(shuffled, key=lambda e: (e[0][0] if e else -1, len(e)))
        mat = w.Mat(tree)
        mat.set_use_crowns(crowns)
        ot = oracle.OracleTree(tree)
        for name, batch in (("shuffled", shuffled), ("sorted", by_pos)):
            reads = ft.reads_from_samples(batch)
            assert_same(mat.place_batch(reads), ot.place_batch(reads, 8), f"dups {it} {name} crowns={crowns}")
        mat.close()


def test_one_handle_many_batch_sizes(oracle):
    """The handle keeps grow-only device / pinned buffers and workspaces between calls: batches of
    growing, shrinking and zero size on ONE handle must all match the oracle."""
    rng = np.random.default_rng(909)
    tree, ref = ft.random_tree(rng, n_nodes=300, genome=120)
    mat = w.Mat(tree)
    ot = oracle.OracleTree(tree)
    for n in (3, 700, 1, 0, 5000, 64, 65, 4097, 2):
        reads = ft.reads_from_samples([ft.random_sample(rng, ref, genome=120) for _ in range(n)])
        res = mat.place_batch(reads)
        assert len(res.score) == n
        if n:
            assert_same(res, ot.place_batch(reads, 8), f"batch of {n}")
    mat.close()


def test_per_node_scores_mode_vs_oracle(oracle):
    """--write-parsimony-scores-per-node: all N values per read (usher_common.cpp:403-409)."""
    rng = np.random.default_rng(321)
    for it in range(25):
        tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(1, 500)), genome=int(rng.choice([60, 300])),
                                   max_muts=int(rng.choice([2, 5])))
        samples = [ft.random_sample(rng, ref, genome=max(ref)) for _ in range(int(rng.integers(1, 12)))]
        mat = w.Mat(tree)
        res = mat.place_batch(ft.reads_from_samples(samples), per_node_scores=True)
        ot = oracle.OracleTree(tree)
        for q, S in enumerate(samples):
            cols = list(zip(*S)) if S else ([], [], [], [])
            assert (res.per_node_scores[q] == ot.place_sample(*cols, per_node_scores=True)["node_scores"]).all(), (it, q)
        mat.close()
    g = w.generate_tree(61, 30000, p_ambiguous=0.01, p_masked_node=0.002, root_mutations=2)
    reads = g.reads(62, 12, p_iupac=0.1)
    mat = w.Mat(g.tree)
    res = mat.place_batch(reads, per_node_scores=True)
    ot = oracle.OracleTree(g.tree)
    for q in range(reads.n_reads):
        p, rf, a, ms = reads.entries(q)
        assert (res.per_node_scores[q] == ot.place_sample(p, rf, a, ms, per_node_scores=True)["node_scores"]).all()
    mat.close()


def test_imputed_mutations_of_the_chosen_node(oracle):
    """Column 4 of placement_stats.tsv: node_imputed_mutations[best_j] (usher_common.cpp:764-781)."""
    rng = np.random.default_rng(654)
    total = 0
    for it in range(40):
        tree, ref = ft.random_tree(rng, n_nodes=int(rng.integers(1, 200)), genome=int(rng.choice([60, 300])))
        samples = [ft.random_sample(rng, ref, genome=max(ref)) for _ in range(int(rng.integers(1, 20)))]
        reads = ft.reads_from_samples(samples)
        mat = w.Mat(tree)
        res = mat.place_batch(reads)
        got = mat.imputed_mutations(reads, res.best_bfs_j)
        ot = oracle.OracleTree(tree)
        for q, S in enumerate(samples):
            cols = list(zip(*S)) if S else ([], [], [], [])
            want = ot.imputed_at_node(*cols, res.best_bfs_j[q])
            assert got[q] == want, (it, q, S)
            total += len(want)
        mat.close()
    assert total > 100


def test_config1_rsv_like(oracle):
    """BASELINE.json configs[0] substitute: RSV-A-like MAT, 150 bp reads."""
    g = w.generate_tree(1, 50000, genome_len=15225, p_ambiguous=0.002, p_masked_node=0.0005, root_mutations=1)
    reads = g.reads(2, 3000, p_iupac=0.05)
    mat = w.Mat(g.tree)
    assert_same(mat.place_batch(reads), oracle.OracleTree(g.tree).place_batch(reads, os.cpu_count()), "config1")
    mat.close()


def test_config2_100k_nodes(oracle):
    """BASELINE.json configs[1]: N=1e5, 1e5 ARTIC-like reads on the GPU; the
    oracle checks a 1500-read subsample, the rest through tile invariance."""
    g = w.generate_tree(11, 100000)
    reads = g.reads(12, 100000)
    mat = w.Mat(g.tree)
    res = mat.place_batch(reads)
    sub = reads.slice(0, 1500)
    want = oracle.OracleTree(g.tree).place_batch(sub, os.cpu_count())
    for k, arr in (("score", res.score), ("best_j", res.best_bfs_j), ("num_best", res.num_best),
                   ("has_unique", res.has_unique)):
        assert (arr[:1500] == want[k]).all(), k
    mat.set_tile_reads(7)
    res7 = mat.place_batch(reads)
    assert (res7.score == res.score).all() and (res7.best_bfs_j == res.best_bfs_j).all()
    assert (res7.num_best == res.num_best).all() and (res7.flags == res.flags).all()
    mat.close()


def test_small_batches_on_a_large_tree(oracle):
    """Few reads on many nodes: every tile is swept in many chunks, whose partial results k_finalize
    combines with 4, 16 or 64 lanes per read (finalize_lanes_per_read) instead of one thread."""
    g = w.generate_tree(61, 200000)
    reads = g.reads(62, 20000, p_iupac=0.01)
    ot = oracle.OracleTree(g.tree)
    want = ot.place_batch(reads.slice(0, 1200), os.cpu_count())
    mat = w.Mat(g.tree)
    for crowns in (True, False):
        mat.set_use_crowns(crowns)
        for n in (1, 9, 70, 1200, 5000, 20000):
            res = mat.place_batch(reads.slice(0, n))
            m = min(n, 1200)
            for k, arr in (("score", res.score), ("best_j", res.best_bfs_j), ("num_best", res.num_best),
                           ("has_unique", res.has_unique)):
                assert (np.asarray(arr[:m]) == np.asarray(want[k][:m])).all(), (crowns, n, k)
    mat.close()


def test_crowns_on_and_off_agree(oracle):
    """Work skipping (crown streams) never changes a result."""
    g = w.generate_tree(51, 300000, p_ambiguous=0.002, root_mutations=1)
    reads = g.reads(52, 30000, p_iupac=0.02)
    mat = w.Mat(g.tree)
    assert mat.stats.n_streams >= 3
    on = mat.place_batch(reads)
    mat.set_use_crowns(False)
    off = mat.place_batch(reads)
    for a, b in ((on.score, off.score), (on.best_bfs_j, off.best_bfs_j), (on.num_best, off.num_best),
                 (on.flags, off.flags)):
        assert (a == b).all()
    sub = reads.slice(0, 800)
    want = oracle.OracleTree(g.tree).place_batch(sub, os.cpu_count())
    assert (on.score[:800] == want["score"]).all() and (on.best_bfs_j[:800] == want["best_j"]).all()
    assert (on.num_best[:800] == want["num_best"]).all() and (on.has_unique[:800] == want["has_unique"]).all()
    mat.close()


def test_config5_long_reads(oracle):
    """BASELINE.json configs[4] shape: ~1.2 kb reads, 3 % substitutions, 2 % N (40-70 entries per read)."""
    g = w.generate_tree(11, 100000)
    reads = g.reads(24, 600, read_len=1200, amplicon_len=1200, amplicon_step=1000, p_substitution=0.03, p_n=0.02)
    assert np.diff(reads.read_off).mean() > 30
    mat = w.Mat(g.tree)
    assert_same(mat.place_batch(reads), oracle.OracleTree(g.tree).place_batch(reads, os.cpu_count()), "config5")
    mat.close()


def test_long_reads_on_both_kinds_of_window_stream(oracle):
    """The tiles of long reads sweep their genome window's CANDIDATE crown -- the nodes with out_w - in_w <= base(root),
    the only ones a read confined to the window can be placed on whatever it lists (wepp_mat_stats:
    n_window_streams_crown) -- or, where the candidates are too many to be worth a crown (small trees), the whole tree
    as the window sees it.  Trees of both kinds, substitution-rich 900 bp reads (root scores far beyond every bounded
    crown), ambiguity codes, masked nodes, root mutations: the oracle's placements on either."""
    rng = np.random.default_rng(5)
    crowns = whole = on_win = 0
    for it, n in enumerate([400, 1200, 3000, 9000, 30000, 90000]):
        # (a 5 kb genome: a window is half of it and nearly every node a candidate -- whole-tree window streams; the
        # 29.9 kb one: candidate crowns)
        g = w.generate_tree(300 + it, n, genome_len=5000 if it < 3 else 29903, p_ambiguous=0.02, p_masked_node=0.01,
                            root_mutations=it % 3, p_back_mutation=0.05)
        reads = g.reads(400 + it, 700, read_len=900, amplicon_len=1000, amplicon_step=700, p_substitution=0.05, p_n=0.02, p_iupac=0.1)
        mat = w.Mat(g.tree)
        st = mat.stats
        assert st.n_window_streams >= 4 and st.window_stream_nodes > 0
        crowns += st.n_window_streams_crown
        whole += st.n_window_streams - st.n_window_streams_crown
        for tile in (64, 16):
            mat.set_tile_reads(tile)
            assert_same(mat.place_batch(reads), oracle.OracleTree(g.tree).place_batch(reads, os.cpu_count()), f"n={n} tile={tile}")
        pcls, _ = mat.last_plans(reads.n_reads)
        on_win += int((pcls == w.PLAN_WIN).sum())
        mat.close()
    assert crowns >= 60 and whole >= 10 and on_win > 2000, (crowns, whole, on_win)


def test_ambiguous_masked_root_mutations(oracle):
    g = w.generate_tree(41, 20000, genome_len=3000, p_ambiguous=0.03, p_masked_node=0.01, root_mutations=5,
                        p_back_mutation=0.2)
    reads = g.reads(42, 2000, p_substitution=0.01, p_n=0.02, p_iupac=0.3)
    mat = w.Mat(g.tree)
    assert_same(mat.place_batch(reads), oracle.OracleTree(g.tree).place_batch(reads, os.cpu_count()), "ambig")
    mat.close()


def test_edge_batches(oracle):
    g = w.generate_tree(43, 5000, genome_len=2000)
    mat = w.Mat(g.tree)
    ot = oracle.OracleTree(g.tree)
    # all reads empty, a single read, reads with positions the tree never mutates / beyond max_position
    for reads in (Reads.from_lists([[] for _ in range(130)]),
                  Reads.from_lists([[(5, A, C, 0)]]),
                  Reads.from_lists([[(1999, A, N, 1), (2000, C, T, 0)], [(100000, G, A, 0)], []])):
        assert_same(mat.place_batch(reads), ot.place_batch(reads, 2), "edge")
    # zero reads is a no-op
    res = mat.place_batch(Reads(np.zeros(1, np.uint32), np.zeros(0, np.uint32)))
    assert res.score.size == 0
    mat.close()


def test_big_tile_falls_back_to_global_read_words(oracle):
    """Tiles whose read words exceed the LDS budget take the S-in-global variant."""
    g = w.generate_tree(45, 30000)
    reads = g.reads(46, 256, read_len=5000, amplicon_len=5000, amplicon_step=4000, p_substitution=0.04, p_n=0.03)
    assert np.diff(reads.read_off).max() * 64 > 16384
    mat = w.Mat(g.tree)
    assert_same(mat.place_batch(reads), oracle.OracleTree(g.tree).place_batch(reads, os.cpu_count()), "bigtile")
    mat.close()


def test_mixed_batch_short_long_and_huge_reads(oracle):
    """One batch that exercises every sweep variant at once: short reads (fused plain
    plans on several crown streams), long reads (dense 8-wave variant) and a few reads too
    long for LDS (words left in global memory)."""
    g = w.generate_tree(81, 60000, genome_len=20000, p_ambiguous=0.003, root_mutations=1)
    short = g.reads(82, 700, p_iupac=0.05)
    long_ = g.reads(83, 150, read_len=1200, amplicon_len=1200, amplicon_step=1000, p_substitution=0.03, p_n=0.02)
    huge = g.reads(84, 3, read_len=20000, amplicon_len=20000, amplicon_step=20000, p_substitution=0.5, p_n=0.3)
    assert np.diff(huge.read_off).max() > 8192
    lists = []
    rng = np.random.default_rng(1)
    for src in (short, long_, huge):
        for q in range(src.n_reads):
            p, rf, a, ms = src.entries(q)
            lists.append([(int(p[i]), int(rf[i]), int(a[i]), int(ms[i])) for i in range(len(p))])
    order = rng.permutation(len(lists))
    reads = Reads.from_lists([lists[i] for i in order])
    mat = w.Mat(g.tree)
    res = mat.place_batch(reads)
    assert_same(res, oracle.OracleTree(g.tree).place_batch(reads, os.cpu_count()), "mixed")
    mat.set_use_crowns(False)
    off = mat.place_batch(reads)
    assert (off.score == res.score).all() and (off.best_bfs_j == res.best_bfs_j).all()
    assert (off.num_best == res.num_best).all() and (off.flags == res.flags).all()
    mat.close()


def _window_edge_case():
    """Tree and reads of test_window_plans_at_their_edges (also built by its child process)."""
    g = w.generate_tree(91, 60000, p_ambiguous=0.002, p_back_mutation=0.05)
    tree = g.tree
    L = 29903
    ref_at = np.full(L + 1, 1, np.uint8)                      # reference allele by position (A where nothing mutates)
    ref_at[tree.mut_pos] = tree.mut_ref
    rng = np.random.default_rng(5)
    lists = []
    for start, span in [(0, 2560), (0, 2561), (1023, 1537), (1023, 1538), (1024, 2560), (2047, 1536), (2047, 1537),
                        (5000, 1536), (5000, 1200), (5119, 2561), (29696, 207), (29000, 903), (27136, 2560),
                        (27136, 2767), (28671, 1232), (12287, 1538)]:
        for k in (17, 24, 60, 140):
            inner = rng.choice(np.arange(start + 1, start + span - 1), size=k - 2, replace=False)
            pos = sorted(set([start, start + span - 1]) | set(int(x) for x in inner))
            ents = []
            for q in pos:
                rf = int(ref_at[q])
                if rng.random() < 0.3:
                    ents.append((q, rf, 15, 1))                                   # N
                else:
                    ents.append((q, rf, int(1 << rng.integers(0, 4)), 0))         # any allele, the reference's too
            lists.append(ents)
    order = rng.permutation(len(lists))
    return g, Reads.from_lists([lists[i] for i in order])


def test_window_plans_at_their_edges(oracle):
    """Reads with more entries than a walk takes are swept on the stream of the genome window (2560 positions
    every 1024) their first position falls in -- if their last position fits, on the whole tree otherwise.  Reads
    built to sit exactly on both sides of that rule, at window starts, at the genome's end and across the
    tree's last mutated position; long (table variant of the window sweep) and 17..24 entries, with walks on
    and off (plain sweeps of window streams) and without work skipping (no window plans at all).  A child
    process with WEPP_DEBUG_PLANS=1 shows that the batch really takes window plans and whole-tree plans."""
    import subprocess, sys
    g, reads = _window_edge_case()
    tree = g.tree
    want = oracle.OracleTree(tree).place_batch(reads, os.cpu_count())
    mat = w.Mat(tree)
    for walk in (True, False):
        mat.set_use_walk(walk)
        assert_same(mat.place_batch(reads), want, f"window edges, walk={walk}")
    mat.set_use_walk(True)
    mat.set_use_crowns(False)
    assert_same(mat.place_batch(reads), want, "window edges, no work skipping")
    mat.close()
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import wepp_amd as w; "
            "from test_gpu_parity import _window_edge_case; g, r = _window_edge_case(); m = w.Mat(g.tree); m.place_batch(r); m.close()"
            % (os.path.dirname(here), here))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WEPP_DEBUG_PLANS="1"), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    plans = [l for l in out.stderr.splitlines() if l.startswith("[plan]")]
    win = [l for l in plans if l.startswith("[plan] window")]
    assert len(win) >= 6 and any("dense=1" in l for l in win), plans          # several windows, the table variant among them
    assert any(l.startswith("[plan] sweep") and "dense=1" in l for l in plans), plans   # reads that fit no window


def test_rejects_unsorted_or_duplicate_read_positions():
    g = w.generate_tree(47, 1000)
    mat = w.Mat(g.tree)
    with pytest.raises(w.WeppError) as ei:
        mat.place_batch(Reads.from_lists([[(20, A, C, 0), (10, G, T, 0)]]))
    assert ei.value.code == 1
    with pytest.raises(w.WeppError):
        mat.place_batch(Reads.from_lists([[(10, A, C, 0), (10, A, G, 0)]]))
    mat.close()


def test_large_batches_are_checked_by_the_host_workers():
    """Batches of 65 536 reads and more are checked and staged by the handle's host workers with a fast pass
    (descents of the positions counted over all words and over the read starts) that falls back to the exact
    read-by-read check: a broken read is reported by its index, the lowest one when there are several; a read
    that starts below its predecessor's last position is fine."""
    g = w.generate_tree(48, 3000)
    reads = g.reads(49, 90_000, p_n=0.03)
    k = np.diff(reads.read_off)
    multi = np.flatnonzero(k >= 2)
    assert len(multi) > 200
    mat = w.Mat(g.tree)
    good = mat.place_batch(reads)
    # sorted batches make many reads start below the previous read's last position: must not be flagged
    starts = reads.read_off[1:-1][(k[1:] > 0) & (k[:-1] > 0)]
    prev_last = reads.read_word[starts - 1] & 0xFFFFF
    first = reads.read_word[starts] & 0xFFFFF
    assert (first <= prev_last).any()

    def broken(fn):
        rw = reads.read_word.copy()
        ro = reads.read_off.copy()
        fn(ro, rw)
        return Reads(ro, rw)

    r1, r2 = int(multi[len(multi) // 2]), int(multi[-3])
    def swap(ro, rw, r):
        a = int(ro[r]); rw[a], rw[a + 1] = rw[a + 1], rw[a]
    with pytest.raises(w.WeppError, match=f"read {r1}:") as ei:
        mat.place_batch(broken(lambda ro, rw: (swap(ro, rw, r1), swap(ro, rw, r2))))
    assert ei.value.code == 1
    with pytest.raises(w.WeppError, match=f"read {r2}:"):
        mat.place_batch(broken(lambda ro, rw: swap(ro, rw, r2)))
    def dup(ro, rw):
        a = int(ro[r1]); rw[a + 1] = (rw[a + 1] & ~np.uint32(0xFFFFF)) | (rw[a] & np.uint32(0xFFFFF))
    with pytest.raises(w.WeppError, match=f"read {r1}:"):
        mat.place_batch(broken(dup))
    def zero_mask(ro, rw):
        rw[int(ro[r2])] &= ~np.uint32(0xF << 24)
    with pytest.raises(w.WeppError, match=f"read {r2}: zero nucleotide mask"):
        mat.place_batch(broken(zero_mask))
    def bad_off(ro, rw):
        ro[r1 + 1] = ro[r1] - 1 if ro[r1] else ro[r1 + 2] + 1
    with pytest.raises(w.WeppError, match="read_off"):
        mat.place_batch(broken(bad_off))
    again = mat.place_batch(reads)                       # the handle is fine after the rejections
    assert (again.score == good.score).all() and (again.best_bfs_j == good.best_bfs_j).all()
    mat.close()


def test_device_pointer_entry_point(oracle):
    import torch
    g = w.generate_tree(49, 40000)
    reads = g.reads(50, 5000)
    mat = w.Mat(g.tree)
    dev = torch.device("cuda", 0)
    d_off = torch.from_numpy(reads.read_off.astype(np.int32)).to(dev)
    d_word = torch.from_numpy(reads.read_word.astype(np.int32)).to(dev)
    outs = [torch.zeros(reads.n_reads, dtype=torch.int32, device=dev) for _ in range(4)]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        mat.place_batch_device(d_off.data_ptr(), d_word.data_ptr(), reads.n_reads, int(reads.read_off[-1]),
                               outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(),
                               stream.cuda_stream)
    stream.synchronize()
    want = oracle.OracleTree(g.tree).place_batch(reads, os.cpu_count())
    assert (outs[0].cpu().numpy().view(np.uint32) == want["best_j"]).all()
    assert (outs[1].cpu().numpy() == want["score"]).all()
    assert (outs[2].cpu().numpy().view(np.uint32) == want["num_best"]).all()
    assert ((outs[3].cpu().numpy() & 1) == want["has_unique"]).all()
    ms, n, passes, nbytes = mat.last_timing()
    assert ms > 0 and n == 1 and passes >= (reads.n_reads + 63) // 64 and nbytes > 0
    # with work skipping and the per-read walks off every tile sweeps the whole-tree stream exactly once
    mat.set_use_crowns(False)
    mat.set_use_walk(False)
    mat.timing_reset()
    with torch.cuda.stream(stream):
        mat.place_batch_device(d_off.data_ptr(), d_word.data_ptr(), reads.n_reads, int(reads.read_off[-1]),
                               outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(),
                               stream.cuda_stream)
    stream.synchronize()
    ms, n, passes, nbytes = mat.last_timing()
    assert passes == (reads.n_reads + 63) // 64 and nbytes == passes * mat.stats.stream_bytes
    assert (outs[1].cpu().numpy() == want["score"]).all()
    mat.close()


def test_full_size_parity_and_properties(oracle):
    """BASELINE.json configs[2] at full size -- 16 M nodes, the whole 1 M-read batch, work skipping on --
    configs[4]'s read shape and configs[3]'s per-GPU shard (1.25 M reads) on the same MAT.
      1. the incremental CPU checker (oracle/incremental_oracle.c, proven equal to the faithful
         restatement by tests/test_incremental.py) on >= 10 000 of the 150-bp reads, drawn so that EVERY
         non-empty sweep stream (crown tier) is covered, and on >= 500 1.2-kb reads;
      2. the faithful restatement of mapper2_body itself (node range split over the host threads, as the
         reference's parallel_for does) on 2 reads of every non-empty tier and 4 of the long reads;
      3. size-independent properties over the whole batch: tile-size invariance, work skipping on/off,
         batch-permutation invariance, score bounds."""
    nthr = os.cpu_count() or 1
    g = w.generate_tree(21, 16_000_000)
    reads = g.reads(22, 1_000_000)
    mat = w.Mat(g.tree)
    res = mat.place_batch(reads)
    tiers = mat.last_tiers(reads.n_reads)
    ot = oracle.OracleTree(g.tree)
    inc = ot.incremental()

    def gather(idx, reads=reads):
        lists_off = np.zeros(len(idx) + 1, np.uint32)
        words = []
        for i, q in enumerate(idx):
            a, b = int(reads.read_off[q]), int(reads.read_off[q + 1])
            words.append(reads.read_word[a:b])
            lists_off[i + 1] = lists_off[i] + (b - a)
        return Reads(lists_off, np.concatenate(words) if words else np.zeros(0, np.uint32))

    def same(got, idx, want):
        assert (got.score[idx] == want["score"]).all()
        assert (got.best_bfs_j[idx] == want["best_j"]).all()
        assert (got.num_best[idx] == want["num_best"]).all()
        assert (got.has_unique[idx] == want["has_unique"]).all()

    # 1. up to 900 reads of every non-empty PLAN -- (class, stream): plain walks of both sizes, chunked walks, sweeps,
    # each on every stream the batch reaches --, then the head of the batch up to 10 240 reads
    pcls, pst = mat.last_plans(reads.n_reads)
    assert (np.where(pcls == w.PLAN_WIN, mat.stats.n_streams - 1, pst) == tiers).all()
    present = [int(t) for t in np.unique(tiers)]
    # a read that walks a window crown (stream slot 15) counts per crown LEVEL of its window: every level the batch
    # reaches gets its share of checked reads, like every tree-wide stream
    _, wcrown = mat.last_crowns(reads.n_reads)
    assert ((wcrown != 255) == ((pst == w.WINDOW_CROWN_SLOT) & (pcls != w.PLAN_WIN) & (pcls != w.PLAN_SWEEP))).all()
    plan_key = pcls.astype(np.int32) * 1024 + pst.astype(np.int32) * 16 + np.where(wcrown == 255, 0, wcrown)
    assert len(np.unique(plan_key)) >= 6, np.unique(plan_key)      # the batch reaches many streams and crown levels
    assert (pst == w.WINDOW_CROWN_SLOT).mean() > 0.3               # ... most of its reads with entries through window crowns
    pick = []
    for key in np.unique(plan_key):
        pick.extend(np.nonzero(plan_key == key)[0][:900].tolist())
    rest = np.setdiff1d(np.arange(20000), np.array(pick))
    pick = np.array(sorted(set(pick) | set(rest[: max(0, 10240 - len(pick))].tolist())))
    assert len(pick) >= 10000
    same(res, pick, inc.place_batch(gather(pick), nthreads=nthr))
    # every non-empty plan has at least 200 of its reads (or all of them) among the checked ones
    checked = np.zeros(reads.n_reads, bool)
    checked[pick] = True
    for key in np.unique(plan_key):
        members = plan_key == key
        assert (checked & members).sum() >= min(200, members.sum()), (w.PLAN_NAMES[key // 1024], (key // 16) % 64, key % 16)
    assert w.PLAN_WALK8 in set(np.unique(pcls).tolist())
    # 2. the faithful oracle: 2 reads per tier
    few = np.array([q for t in present for q in np.nonzero(tiers == t)[0][:2].tolist()])
    same(res, few, ot.place_batch(gather(few), nthr, node_parallel=True))

    # configs[4] shape: 1.2 kb reads, ~58 entries each (dense sweep variant), same MAT
    long_reads = g.reads(24, 600, read_len=1200, amplicon_len=1200, amplicon_step=1100, p_substitution=0.03, p_n=0.02)
    rl = mat.place_batch(long_reads)
    all_long = np.arange(long_reads.n_reads)
    same(rl, all_long, inc.place_batch(long_reads, nthreads=nthr))
    same(rl, np.arange(4), ot.place_batch(long_reads.slice(0, 4), nthr, node_parallel=True))
    ot.close()

    # configs[3] shape: the 1.25 M-read shard one of eight GPUs places (rank 3's reads of a 10 M-read run);
    # every 2441st read against the incremental checker, and two half batches = the whole
    shard = g.reads(22 + 3, 1_250_000)
    rs = mat.place_batch(shard)
    every = np.arange(0, shard.n_reads, 2441)[:512]
    got = inc.place_batch(gather(every, shard), nthreads=nthr)
    assert (rs.score[every] == got["score"]).all() and (rs.best_bfs_j[every] == got["best_j"]).all()
    assert (rs.num_best[every] == got["num_best"]).all() and (rs.has_unique[every] == got["has_unique"]).all()
    half = shard.n_reads // 2
    ra, rb = mat.place_batch(shard.slice(0, half)), mat.place_batch(shard.slice(half, shard.n_reads))
    for f in ("score", "best_bfs_j", "num_best", "flags"):
        assert (np.concatenate([getattr(ra, f), getattr(rb, f)]) == getattr(rs, f)).all(), f
    inc.close()

    # 3. properties
    mat.set_tile_reads(16)
    sub = reads.slice(0, 20000)
    r16 = mat.place_batch(sub)
    assert (r16.score == res.score[:20000]).all() and (r16.best_bfs_j == res.best_bfs_j[:20000]).all()
    assert (r16.num_best == res.num_best[:20000]).all() and (r16.flags == res.flags[:20000]).all()
    mat.set_tile_reads(64)
    mat.set_use_crowns(False)
    roff = mat.place_batch(sub)
    assert (mat.last_tiers(sub.n_reads) == mat.stats.n_streams - 1).all()
    mat.set_use_crowns(True)
    assert (roff.score == res.score[:20000]).all() and (roff.best_bfs_j == res.best_bfs_j[:20000]).all()
    assert (roff.num_best == res.num_best[:20000]).all() and (roff.flags == res.flags[:20000]).all()
    rng = np.random.default_rng(0)
    perm = rng.permutation(20000)
    rp = mat.place_batch(gather(perm))
    assert (rp.score == res.score[perm]).all() and (rp.best_bfs_j == res.best_bfs_j[perm]).all()
    assert (rp.num_best == res.num_best[perm]).all()
    # scores are bounded by the root placement: len(non-missing S) (root has no mutations here)
    cs = np.concatenate([[0], np.cumsum(((reads.read_word >> 28) & 1) == 0)])
    k_nm = cs[reads.read_off[1:].astype(np.int64)] - cs[reads.read_off[:-1].astype(np.int64)]
    assert (res.score <= k_nm).all() and (res.score >= 0).all() and (res.num_best >= 1).all()
    mat.close()


def test_full_size_nrich_and_long_shard():
    """What bench.py's sensitivity ladder times, oracle-checked at the size it is timed at: the N rate 5 % batch and
    the exactly-8-entries batch on the 16 M-node MAT (chunked walks of both classes, start states by bisection,
    per-entry pre-test bytes at the PRODUCT's threshold: the child process does not inherit this suite's
    WEPP_IX_PRE_MIN_NODES=0), >= 5 000 reads of each against the incremental checker with every (class, stream)
    plan covered; then configs[4] at its per-GPU shard size, 125 000 reads of 1.2 kb: every 25th read against the
    checker, two half batches = the whole, tile-size invariance.  tests/nrich_full_size.py does the work."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "WEPP_IX_PRE_MIN_NODES"}
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nrich_full_size.py")
    run = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=1500)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    rep = json.loads(run.stdout.strip().splitlines()[-1])
    assert rep["nodes"] == 16_000_000 and {"walk8", "walk16"} <= set(rep["classes_seen"])
    assert all(b["checked"] >= 5000 for b in rep["batches"]) and rep["long_reads"]["checked"] >= 5000
    assert rep["long_reads"]["window_plan_share"] > 0.9
    print("full-size N-rich / long-shard report:", json.dumps(rep))


def test_full_size_genome_samples_epp_fitch():
    """The three paths whose speeds are quoted at 16 M nodes, oracle-checked at 16 M nodes (tests/full_size_more.py, a
    child process with the product's defaults): whole-genome samples through the seeded path -- every 4th of 2 400
    against the incremental checker, 4 against the faithful restatement, best_j_vec of 50, seeds off, work skipping
    off --; wepp_epp_map -- 300+ of 20 000 reads against oracle_epp_map, run-to-run identity --; wepp_fitch_plan_run --
    64 rows against oracle_mapper_body."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "WEPP_IX_PRE_MIN_NODES"}
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "full_size_more.py")
    run = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=1500)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    rep = json.loads(run.stdout.strip().splitlines()[-1])
    assert rep["nodes"] == 16_000_000
    assert sum(leg["checked_incremental"] for leg in rep["whole_genome_samples"]["legs"]) >= 600
    assert all(leg["chunks_evaluated_per_sample"] < leg["chunks"] / 100 for leg in rep["whole_genome_samples"]["legs"])
    assert rep["epp"]["checked_against_oracle"] >= 300 and rep["fitch"]["rows"] >= 64
    print("full-size genome samples / EPP / Fitch report:", json.dumps(rep))


def test_window_bound_on_fuzz_trees():
    """The window-candidate bound on the GPU with the adversarial fuzz: a test-only build of the library with genome
    windows of 64 positions every 32 (tools/build_variant.sh win64; built by __graft_entry__.build(), or here when
    missing) places reads confined to such windows -- walks of window crowns, per-read sweeps of a window crown, window
    tiles -- on trees with masked nodes, multi-allelic alleles, repeated positions and back-mutations straddling the
    window edges; tests/window_fuzz.py compares every read with the faithful oracle."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    lib = os.path.join(root, "variants", "win64", "libwepp_place.so")
    csrc = os.path.join(root, "wepp_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".hpp")))
    if not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        subprocess.run(["bash", os.path.join(root, "tools", "build_variant.sh"), "win64", "-DWEPP_WIN_SIZE=64", "-DWEPP_WIN_STRIDE=32"],
                       check=True, capture_output=True, timeout=1500)
    env = dict(os.environ, WEPP_PLACE_LIB=lib)
    run = subprocess.run([sys.executable, os.path.join(HERE, "window_fuzz.py")], env=env, capture_output=True, text=True, timeout=1500)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    rep = json.loads(run.stdout.strip().splitlines()[-1])
    assert rep["trees_with_window_crowns"] >= 20 and rep["reads_on_window_crowns"] >= 1000, rep
    assert {"walk8", "sweep", "window"} <= set(rep["reads_by_plan_class"]), rep
    print("window fuzz report:", json.dumps(rep))


def test_genome_beyond_the_window_table(oracle):
    """A 100 kb genome: positions from 32 * 1024 on lie in no genome window (wepp_mat_stats::window_uncovered_positions);
    reads there take the tree-wide streams -- a documented limit, not a silent one; same results as the oracle on both
    sides of the edge and across it."""
    L = 100000
    g = w.generate_tree(41, 60000, genome_len=L)
    mat = w.Mat(g.tree)
    assert mat.stats.window_size == 2560 and mat.stats.window_stride == 1024
    assert mat.stats.window_uncovered_positions > 60000
    reads = g.reads(42, 3000, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.002, p_n=0.02)
    res = mat.place_batch(reads)
    first = np.array([reads.read_word[reads.read_off[r]] & 0xFFFFF if reads.read_off[r + 1] > reads.read_off[r] else 0 for r in range(reads.n_reads)])
    assert (first >= 32 * 1024 + 2560).sum() > 500 and ((first > 0) & (first < 32 * 1024)).sum() > 200
    assert_same(res, oracle.OracleTree(g.tree).incremental().place_batch(reads, nthreads=8), "100 kb genome")
    _, pst = mat.last_plans(reads.n_reads)
    beyond = first >= 32 * 1024 + 2560
    assert (pst[beyond] != w.WINDOW_CROWN_SLOT).all()              # no window crown out there
    mat.close()


def test_window_crowns_walks_and_sweeps_vs_oracle(oracle):
    """Window crowns (include/wepp_place.h: wepp_mat_last_crowns): reads confined to a genome window are placed on the
    crown of their window that their ROOT score admits -- by a walk when they list at most 16 positions, by a sweep of
    the crown of their own when they list 17-32 (k_sweep_arena), in tiles on the window's candidates when they list more
    (N-rich reads: 10 % and 20 % N) -- with the oracle's results; with the walks off every read of a crown sweeps it;
    with work skipping off none does.  Crowns of several levels are reached."""
    g = w.generate_tree(81, 150_000, p_ambiguous=0.01, p_masked_node=0.002, root_mutations=1)
    batches = [g.reads(82, 3000, p_substitution=0.004, p_n=0.02, p_iupac=0.1), g.reads(83, 1500, p_n=0.10),
               g.reads(84, 1500, p_substitution=0.01, p_n=0.20)]
    reads = Reads(np.concatenate([[0]] + [b.read_off[1:].astype(np.int64) + sum(int(x.read_off[-1]) for x in batches[:i])
                                          for i, b in enumerate(batches)]).astype(np.uint32),
                  np.concatenate([b.read_word for b in batches]))
    want = oracle.OracleTree(g.tree).incremental().place_batch(reads, nthreads=os.cpu_count() or 1)
    mat = w.Mat(g.tree)
    assert mat.stats.n_window_crowns >= 30 and mat.stats.window_crown_nodes > 0

    def check(res):
        assert (res.score == want["score"]).all() and (res.best_bfs_j == want["best_j"]).all()
        assert (res.num_best == want["num_best"]).all() and (res.has_unique == want["has_unique"]).all()

    check(mat.place_batch(reads))
    pcls, pst = mat.last_plans(reads.n_reads)
    win, crown = mat.last_crowns(reads.n_reads)
    on_crown = (pst == w.WINDOW_CROWN_SLOT) & (pcls != w.PLAN_WIN)      # (a window plan's stream number is its window)
    assert on_crown.mean() > 0.6 and len(np.unique(crown[on_crown])) >= 3 and len(np.unique(win[on_crown])) >= 20
    k = np.diff(reads.read_off.astype(np.int64))
    assert ((pcls == w.PLAN_SWEEP) & on_crown).sum() > 500 and (k[(pcls == w.PLAN_SWEEP) & on_crown] > 16).all()
    # reads with more than 32 entries share tile sweeps of their window's candidates (all of them: any root score)
    assert (pcls == w.PLAN_WIN).sum() > 100 and (k[pcls == w.PLAN_WIN] > 32).all()
    assert ((pcls == w.PLAN_WALK8) & on_crown).sum() > 1000 and ((pcls == w.PLAN_WALK16) & on_crown).sum() > 100
    mat.set_use_walk(False)
    check(mat.place_batch(reads))
    pcls2, pst2 = mat.last_plans(reads.n_reads)
    assert (((pst2 == w.WINDOW_CROWN_SLOT) & (pcls2 != w.PLAN_WIN)) == on_crown).all() and (pcls2[on_crown] == w.PLAN_SWEEP).all()
    mat.set_use_walk(True)
    mat.set_use_crowns(False)
    check(mat.place_batch(reads))
    assert (mat.last_plans(reads.n_reads)[1] != w.WINDOW_CROWN_SLOT).all()
    mat.close()


def test_pipeline_equals_unsplit_call(oracle):
    """wepp_place_batch cuts a large batch into sub-batches that overlap staging, H2D, kernels and D2H
    (include/wepp_place.h): whatever the split -- 1, 2, 3, 4 or 8 sub-batches --, with pageable buffers or with
    buffers the caller pinned, the four result arrays and the per-read plan ids are those of the unsplit call, and
    the unsplit call equals the oracle on a sample."""
    import torch
    g = w.generate_tree(61, 120_000, p_ambiguous=0.01, p_masked_node=0.002, root_mutations=1)
    reads = g.reads(62, 300_001, p_substitution=0.003, p_n=0.02, p_iupac=0.1)
    mat = w.Mat(g.tree)
    mat.set_pipeline(1)
    whole = mat.place_batch(reads)
    plans = mat.last_plans(reads.n_reads)
    every = np.arange(0, reads.n_reads, 601)
    sub = Reads(np.concatenate([[0], np.cumsum(np.diff(reads.read_off.astype(np.int64))[every])]).astype(np.uint32),
                np.concatenate([reads.read_word[reads.read_off[q]:reads.read_off[q + 1]] for q in every]))
    want = oracle.OracleTree(g.tree).incremental().place_batch(sub, nthreads=os.cpu_count() or 1)
    assert (whole.score[every] == want["score"]).all() and (whole.best_bfs_j[every] == want["best_j"]).all()
    assert (whole.num_best[every] == want["num_best"]).all() and (whole.has_unique[every] == want["has_unique"]).all()
    for S in (2, 3, 4, 8, 0):
        mat.set_pipeline(S)
        got = mat.place_batch(reads)
        for f in ("score", "best_bfs_j", "num_best", "flags"):
            assert (getattr(got, f) == getattr(whole, f)).all(), (S, f)
        # (how a read is placed may differ between calls -- the handle sizes the walks' stacks from its last call --,
        # the per-read diagnostics cover the whole call all the same)
        pc, ps = mat.last_plans(reads.n_reads)
        assert ((pc == plans[0]) & (ps == plans[1])).mean() > 0.99, S
    # buffers the caller pinned: DMA straight from / into them
    pin = lambda a: torch.from_numpy(a.copy()).pin_memory().numpy()
    preads = Reads.__new__(Reads)
    preads.read_off, preads.read_word = pin(reads.read_off), pin(reads.read_word)
    out = w.PlacementResult(pin(np.zeros(reads.n_reads, np.uint32)), pin(np.zeros(reads.n_reads, np.int32)),
                            pin(np.zeros(reads.n_reads, np.uint32)), pin(np.zeros(reads.n_reads, np.uint32)))
    mat.set_pipeline(4)
    got = mat.place_batch(preads, out=out)
    assert got.score is out.score
    for f in ("score", "best_bfs_j", "num_best", "flags"):
        assert (getattr(got, f) == getattr(whole, f)).all(), ("pinned", f)
    # a rejected batch leaves the handle usable, whichever sub-batch the bad read is in
    broken = Reads(reads.read_off.copy(), reads.read_word.copy())
    r_bad = int(np.nonzero(np.diff(reads.read_off.astype(np.int64)) >= 2)[0][-1])
    a = int(broken.read_off[r_bad])
    broken.read_word[a], broken.read_word[a + 1] = broken.read_word[a + 1], broken.read_word[a]
    with pytest.raises(w.WeppError, match=f"read {r_bad}:"):
        mat.place_batch(broken)
    again = mat.place_batch(reads)
    assert (again.score == whole.score).all() and (again.best_bfs_j == whole.best_bfs_j).all()
    mat.close()


def test_reads_with_many_events_vs_oracle(oracle, monkeypatch):
    """Reads whose positions are mutated many times in their stream: 17 - 256 events go to the wave-per-read walk
    (wave_kernels.hip: lane = list entry, all-pairs instead of a walk), more to the walks cut into jobs.  A short genome
    makes every list long (60 K nodes over 1 500 positions: ~40 mutations per position); reads of 1 - 12 entries with
    concrete alleles, ambiguity codes and Ns, inside one genome window and across windows; masked and multi-allelic
    nodes in the tree.  Against the incremental checker, and walks off = walks on."""
    g = w.generate_tree(61, 60_000, genome_len=1500, p_ambiguous=0.02, p_masked_node=0.003, root_mutations=1)
    rng = np.random.default_rng(9)
    ref = {}
    for p_, r_ in zip(g.tree.mut_pos, g.tree.mut_ref):
        if p_ >= 0:
            ref[int(p_)] = int(r_)
    positions = np.array(sorted(ref))
    samples = []
    for i in range(6000):
        k = int(rng.integers(1, 13))
        if i % 2:
            lo = int(rng.integers(0, len(positions) - 200))
            pos = np.sort(rng.choice(positions[lo:lo + 200], size=min(k, 200), replace=False))
        else:
            pos = np.sort(rng.choice(positions, size=k, replace=False))
        ents = []
        for p_ in pos:
            u = rng.random()
            if u < 0.5:
                ents.append((int(p_), ref[int(p_)], 1 << int(rng.integers(0, 4)), 0))
            elif u < 0.7:
                ents.append((int(p_), ref[int(p_)], int(rng.integers(1, 15)), 0))
            else:
                ents.append((int(p_), ref[int(p_)], 15, 1))
        samples.append(ents)
    reads = Reads.from_lists(samples)
    mat = w.Mat(g.tree)
    res = mat.place_batch(reads)
    cls, _ = mat.last_plans(reads.n_reads)
    assert (cls == w.PLAN_WALKC8).sum() + (cls == w.PLAN_WALKC16).sum() > 1000, np.bincount(cls).tolist()
    assert_same(res, oracle.OracleTree(g.tree).incremental().place_batch(reads, nthreads=os.cpu_count() or 1), "reads with many events")
    mat.set_use_walk(False)
    r0 = mat.place_batch(reads)
    for f in ("score", "best_bfs_j", "num_best", "flags"):
        assert (getattr(r0, f) == getattr(res, f)).all(), f
    mat.close()
    # the same reads with their walks cut into jobs (what a handle does by itself after a call FULL of such reads), and
    # with a few per routing block by waves and the rest by jobs
    for small, big in (("0", "0"), ("3", "1")):
        monkeypatch.setenv("WEPP_WW_BLOCK_MAX_SMALL", small)
        monkeypatch.setenv("WEPP_WW_BLOCK_MAX_BIG", big)
        m2 = w.Mat(g.tree)
        r2 = m2.place_batch(reads)
        for f in ("score", "best_bfs_j", "num_best", "flags"):
            assert (getattr(r2, f) == getattr(res, f)).all(), (f, small, big)
        m2.close()


def test_host_pipeline_with_few_workers(oracle, monkeypatch):
    """wepp_place_batch's staging with ONE host worker (WEPP_HOST_THREADS=2: what a handle gets when eight of them
    share sixteen cores): the chunks' offsets go up as soon as a chunk's own staging tasks are done, and with one
    worker the tasks run strictly one after the other -- round 3 staged a chunk's END offset in the next chunk's first
    task, so the copy could leave before it (found by tools/host_proxy.py as a fault in k_route).  Batches of different
    sizes in turn (a stale offset of the call before is wrong for this one), each against the incremental checker."""
    monkeypatch.setenv("WEPP_HOST_THREADS", "2")
    g = w.generate_tree(35, 60_000, p_ambiguous=0.01, root_mutations=1)
    inc = oracle.OracleTree(g.tree).incremental()
    batches = [g.reads(36 + i, n, p_substitution=0.004, p_n=0.03) for i, n in enumerate((70_001, 131_075, 66_000, 99_999))]
    wants = [inc.place_batch(b, nthreads=os.cpu_count() or 1) for b in batches]
    mat = w.Mat(g.tree)
    for rnd in range(3):
        for b, want in zip(batches, wants):
            assert_same(mat.place_batch(b), want, f"one host worker, round {rnd}, {b.n_reads} reads")
    mat.close()


def test_two_handles_two_host_threads(oracle):
    """include/wepp_place.h: different handles may be used concurrently from different host threads.
    Two handles on device 0, each placing its own contiguous shard of the batch from its own thread
    (what the C++ multi-GPU driver does per device), several rounds; results against the oracle."""
    import threading
    g = w.generate_tree(31, 60_000, genome_len=5000, p_ambiguous=0.01, p_masked_node=0.002, root_mutations=1)
    reads = g.reads(32, 6000, p_substitution=0.004, p_n=0.01, p_iupac=0.1)
    want = oracle.OracleTree(g.tree).incremental().place_batch(reads, nthreads=os.cpu_count() or 1)
    mats = [w.Mat(g.tree), w.Mat(g.tree)]
    shards = [reads.slice(0, 2500), reads.slice(2500, 6000)]
    los = [0, 2500]
    errs = []

    def worker(k):
        try:
            for _ in range(6):
                r = mats[k].place_batch(shards[k])
                sl = slice(los[k], los[k] + shards[k].n_reads)
                assert (r.score == want["score"][sl]).all() and (r.best_bfs_j == want["best_j"][sl]).all()
                assert (r.num_best == want["num_best"][sl]).all() and (r.has_unique == want["has_unique"][sl]).all()
        except BaseException as e:   # noqa: BLE001 (reported by the main thread)
            errs.append(e)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for m in mats:
        m.close()
    assert not errs, errs


def test_best_nodes_vs_oracle(oracle):
    """best_j_vec (usher_common.cpp:376-381, filled at usher_mapper.cpp:475-476,497): the BFS indices of all optimal
    nodes of a sample, against the oracle's vector -- every (tree, sample) pair of the fuzz trees, then 1 200 reads
    on a 100 K-node tree (reads of every routed stream, short, N-rich and 1.2 kb: the last two list their nodes on their
    genome window's candidates) with work skipping on and off."""
    rng = np.random.default_rng(808)
    n_nodes = 0
    for it in range(30):
        tree, ref = ft.random_tree(rng)
        samples = [ft.random_sample(rng, ref) for _ in range(8)]
        reads = ft.reads_from_samples(samples)
        mat = w.Mat(tree)
        ot = oracle.OracleTree(tree)
        res = mat.place_batch(reads)
        got = mat.best_nodes(reads, res)
        for q, S in enumerate(samples):
            cols = list(zip(*S)) if S else ([], [], [], [])
            want = ot.place_sample(*cols, want_best_vec=True)
            assert got[q].tolist() == want["best_j_vec"].tolist(), (it, q)
            assert res.best_bfs_j[q] in got[q]
            n_nodes += len(got[q])
        mat.close()
    assert n_nodes > 300
    g = w.generate_tree(71, 100_000, p_ambiguous=0.01, p_masked_node=0.002, root_mutations=1)
    short = g.reads(72, 900, p_substitution=0.004, p_n=0.02, p_iupac=0.1)
    long_ = g.reads(73, 100, read_len=1200, amplicon_len=1200, amplicon_step=1020, p_substitution=0.03, p_n=0.02)
    nrich = g.reads(74, 200, p_substitution=0.01, p_n=0.15)       # (a window's candidates are the smaller stream for these)
    reads = Reads(np.concatenate([short.read_off, long_.read_off[1:] + short.read_off[-1],
                                  nrich.read_off[1:] + short.read_off[-1] + long_.read_off[-1]]),
                  np.concatenate([short.read_word, long_.read_word, nrich.read_word]))
    mat = w.Mat(g.tree)
    inc = oracle.OracleTree(g.tree).incremental()
    res = mat.place_batch(reads)
    assert len(np.unique(mat.last_tiers(reads.n_reads))) >= 2
    for crowns in (True, False):
        mat.set_use_crowns(crowns)
        got = mat.best_nodes(reads, res)
        for q in range(reads.n_reads):
            want = inc.place_sample(*reads.entries(q), want_best_vec=True)
            assert got[q].tolist() == want["best_j_vec"].tolist(), (crowns, q)
    # a score that is not the read's: reported, not listed
    wrong = w.PlacementResult(res.best_bfs_j, res.score + 1, res.num_best, res.flags)
    with pytest.raises(w.WeppError, match="not this read's placement"):
        mat.best_nodes(reads, wrong)
    # capacity: WEPP_ELIMIT with the needed size
    off = np.zeros(reads.n_reads + 1, np.uint64)
    small = np.zeros(1, np.uint32)
    rc = w._lib.lib.wepp_best_nodes(mat._h, reads.read_off.ctypes.data, reads.read_word.ctypes.data, reads.n_reads,
                                    res.score.ctypes.data, res.num_best.ctypes.data, off.ctypes.data, small.ctypes.data, 1)
    assert rc == 4 and int(off[-1]) == int(res.num_best.sum())
    mat.close()


@pytest.mark.gpu
def test_excess_mutations_vs_oracle(oracle):
    """node_excess_mutations (usher_mapper.cpp:223-228, :253-258, :357-388, :394-446 with
    compute_vecs) for every (sample, node) pair of small random trees: same mutations, same order."""
    rng = np.random.default_rng(606)
    n_lists = n_muts = 0
    for it in range(25):
        # masked root mutations with non-zero nucleotides are the documented exception
        tree, ref = ft.random_tree(rng, p_root_masked=0.0)
        samples = [ft.random_sample(rng, ref) for _ in range(6)]
        reads = ft.reads_from_samples(samples)
        mat = w.Mat(tree)
        ot = oracle.OracleTree(tree)
        n = tree.n_nodes
        pr = np.repeat(np.arange(len(samples), dtype=np.uint32), n)
        pj = np.tile(np.arange(n, dtype=np.uint32), len(samples))
        got = mat.excess_mutations(reads, pr, pj)
        for q, S in enumerate(samples):
            cols = list(zip(*S)) if S else ([], [], [], [])
            full = ot.place_sample(*cols, per_node_scores=True)
            for j in range(n):
                want = ot.excess_at_node(*cols, j)
                assert got[q * n + j] == want, (it, q, j)
                # shared mutations of the node first, then what the score counts (nodes that do not
                # compete report score + 1, :500-505)
                assert len(want) >= full["node_scores"][j] - 1
                n_lists += 1
                n_muts += len(want)
        mat.close()
    assert n_lists > 3000 and n_muts > 5000
