"""Host-side Python mirror of the C-ABI (thin: numpy arrays in, numpy arrays out).

Names follow the reference's domain: a *tree* is a mutation-annotated tree
(MAT::Tree, src/mutation_annotated_tree.hpp:104-152), a *read* / *sample* is a
Missing_Sample (src/usher_graph.hpp:34-54), *placing* is the per-sample loop
of usher_common (src/usher_common.cpp:307-470).
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import lib, check

A, C, G, T, N = 1, 2, 4, 8, 15  # nucleotide masks, src/mutation_annotated_tree.cpp:19-74
# how a read was placed (include/wepp_place.h WEPP_PLAN_*, Mat.last_plans)
PLAN_WALK8, PLAN_WALK16, PLAN_SWEEP, PLAN_WALKC8, PLAN_WALKC16, PLAN_WIN, PLAN_SEED = range(7)
PLAN_NAMES = ("walk8", "walk16", "sweep", "walkc8", "walkc16", "window", "seed")
WINDOW_CROWN_LEVELS = 7  # crowns per genome window at most (flatmat.hpp: WC_MAX; FlatView 'wc_tau' / 'wc_nodes' rows)
WINDOW_CROWN_SLOT = 15   # stream slot of the window crowns in Mat.last_plans / last_tiers (Mat.last_crowns tells which crown)


def pack_read_word(position, ref_nuc, mut_nuc, is_missing=0):
    """numpy-friendly twin of wepp_pack_read_word (include/wepp_place.h)."""
    position = np.asarray(position, dtype=np.uint32)
    return (
        (position & np.uint32(0xFFFFF))
        | ((np.asarray(ref_nuc, dtype=np.uint32) & np.uint32(15)) << np.uint32(20))
        | ((np.asarray(mut_nuc, dtype=np.uint32) & np.uint32(15)) << np.uint32(24))
        | ((np.asarray(is_missing, dtype=np.uint32) & np.uint32(1)) << np.uint32(28))
    ).astype(np.uint32)


def unpack_read_word(w):
    w = np.asarray(w, dtype=np.uint32)
    return (
        (w & 0xFFFFF).astype(np.int32),
        ((w >> 20) & 15).astype(np.uint8),
        ((w >> 24) & 15).astype(np.uint8),
        ((w >> 28) & 1).astype(np.uint8),
    )


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Tree:
    """Pointer-free description of a MAT (see wepp_tree_desc in include/wepp_place.h)."""

    def __init__(self, parent, mut_off, mut_pos, mut_ref, mut_mut, mut_par=None):
        self.parent = np.ascontiguousarray(parent, dtype=np.int32)
        self.mut_off = np.ascontiguousarray(mut_off, dtype=np.uint32)
        self.mut_pos = np.ascontiguousarray(mut_pos, dtype=np.int32)
        self.mut_ref = np.ascontiguousarray(mut_ref, dtype=np.uint8)
        self.mut_mut = np.ascontiguousarray(mut_mut, dtype=np.uint8)
        self.mut_par = None if mut_par is None else np.ascontiguousarray(mut_par, dtype=np.uint8)
        if self.mut_off.shape[0] != self.parent.shape[0] + 1:
            raise ValueError("mut_off must have n_nodes + 1 entries")

    @property
    def n_nodes(self):
        return int(self.parent.shape[0])

    @classmethod
    def from_lists(cls, parent, muts):
        """muts[i] = list of (position, ref_nuc, mut_nuc) or (position, ref, par, mut)."""
        off = np.zeros(len(parent) + 1, dtype=np.uint32)
        pos, ref, par, mut = [], [], [], []
        for i, ml in enumerate(muts):
            off[i + 1] = off[i] + len(ml)
            for m in ml:
                if len(m) == 3:
                    p, r, mu = m
                    pa = r
                else:
                    p, r, pa, mu = m
                pos.append(p); ref.append(r); par.append(pa); mut.append(mu)
        return cls(parent, off, np.array(pos, np.int32), np.array(ref, np.uint8), np.array(mut, np.uint8),
                   np.array(par, np.uint8))

    def desc(self):
        d = _lib.TreeDescC()
        d.n_nodes = self.n_nodes
        d.parent = self.parent.ctypes.data_as(_lib.c_i32p)
        d.mut_off = self.mut_off.ctypes.data_as(_lib.c_u32p)
        d.mut_pos = self.mut_pos.ctypes.data_as(_lib.c_i32p)
        d.mut_ref = self.mut_ref.ctypes.data_as(_lib.c_u8p)
        d.mut_par = self.mut_par.ctypes.data_as(_lib.c_u8p) if self.mut_par is not None else None
        d.mut_mut = self.mut_mut.ctypes.data_as(_lib.c_u8p)
        return d


class Reads:
    """A batch of samples/reads in CSR form: read_off[R+1], read_word[...]."""

    def __init__(self, read_off, read_word):
        self.read_off = np.ascontiguousarray(read_off, dtype=np.uint32)
        self.read_word = np.ascontiguousarray(read_word, dtype=np.uint32)

    @property
    def n_reads(self):
        return int(self.read_off.shape[0] - 1)

    @classmethod
    def from_lists(cls, reads):
        """reads[r] = list of (position, ref_nuc, mut_nuc[, is_missing])."""
        off = np.zeros(len(reads) + 1, dtype=np.uint32)
        words = []
        for r, ents in enumerate(reads):
            off[r + 1] = off[r] + len(ents)
            for e in ents:
                miss = e[3] if len(e) > 3 else 0
                words.append(int(pack_read_word(e[0], e[1], e[2], miss)))
        return cls(off, np.array(words, dtype=np.uint32))

    def slice(self, lo, hi):
        a, b = int(self.read_off[lo]), int(self.read_off[hi])
        return Reads(self.read_off[lo:hi + 1] - self.read_off[lo], self.read_word[a:b])

    def entries(self, r):
        a, b = int(self.read_off[r]), int(self.read_off[r + 1])
        return unpack_read_word(self.read_word[a:b])


class EppReads(Reads):
    """WEPP's raw_read batch (src/WEPP/read.hpp:8-14): the CSR of mutations plus, per read,
    the 1-based inclusive genome window [start, end] and the multiplicity `degree`."""

    def __init__(self, read_off, read_word, start, end, degree):
        super().__init__(read_off, read_word)
        self.start = np.ascontiguousarray(start, dtype=np.int32)
        self.end = np.ascontiguousarray(end, dtype=np.int32)
        self.degree = np.ascontiguousarray(degree, dtype=np.int32)
        if not (self.start.shape[0] == self.end.shape[0] == self.degree.shape[0] == self.n_reads):
            raise ValueError("start / end / degree must have one entry per read")

    @classmethod
    def from_lists(cls, reads, start, end, degree=None):
        base = Reads.from_lists(reads)
        return cls(base.read_off, base.read_word, start, end, np.ones(len(reads), np.int32) if degree is None else degree)


def generate_tree(seed, n_nodes, genome_len=29903, p_recent_parent=0.25, zipf_s=0.6, p_back_mutation=0.02,
                  p_ambiguous=0.0, p_masked_node=0.0, root_mutations=0, depth_choices=0, p_hub=0.0, n_hubs=0):
    """Deterministic synthetic MAT (wepp_gen_tree_create).  Shape knobs: depth_choices (parent = the deepest of that
    many random earlier nodes: longer root paths), p_hub / n_hubs (polytomies: star-like trees)."""
    p = _lib.GenTreeParams(seed, n_nodes, genome_len, p_recent_parent, zipf_s, p_back_mutation, p_ambiguous,
                           p_masked_node, root_mutations, depth_choices, p_hub, n_hubs)
    h = ctypes.c_void_p()
    check(lib.wepp_gen_tree_create(ctypes.byref(p), ctypes.byref(h)))
    return GenTree(h)


class GenTree:
    def __init__(self, handle):
        self._h = handle
        d = _lib.TreeDescC()
        check(lib.wepp_gen_tree_desc(self._h, ctypes.byref(d)))
        n = d.n_nodes
        m = int(np.ctypeslib.as_array(d.mut_off, shape=(n + 1,))[n])
        mk = lambda p, dt: (np.ctypeslib.as_array(p, shape=(m,)).astype(dt, copy=True) if m else np.zeros(0, dt))
        self.tree = Tree(
            np.ctypeslib.as_array(d.parent, shape=(n,)).copy(),
            np.ctypeslib.as_array(d.mut_off, shape=(n + 1,)).copy(),
            mk(d.mut_pos, np.int32), mk(d.mut_ref, np.uint8), mk(d.mut_mut, np.uint8), mk(d.mut_par, np.uint8),
        )

    def shape(self):
        """Root-path mutations of the leaves (median, mean, 95 %, max), depth, largest polytomy."""
        sh = _lib.GenTreeShape()
        check(lib.wepp_gen_tree_get_shape(self._h, ctypes.byref(sh)))
        return {f: getattr(sh, f) for f, _ in sh._fields_}

    def reads(self, seed, n_reads, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001,
              p_n=0.005, p_iupac=0.0, windows=False, max_degree=1):
        """Synthetic reads; windows=True returns EppReads (with the genome window of every read and a
        multiplicity drawn from 1..max_degree) for Mat.epp_map."""
        p = _lib.GenReadsParams(seed, n_reads, read_len, amplicon_len, amplicon_step, p_substitution, p_n, p_iupac)
        h = ctypes.c_void_p()
        check(lib.wepp_gen_reads_create(self._h, ctypes.byref(p), ctypes.byref(h)))
        try:
            n = ctypes.c_uint32()
            po = _lib.c_u32p()
            pw = _lib.c_u32p()
            check(lib.wepp_gen_reads_get(h, ctypes.byref(n), ctypes.byref(po), ctypes.byref(pw)))
            off = np.ctypeslib.as_array(po, shape=(n.value + 1,)).copy()
            nw = int(off[-1])
            words = np.ctypeslib.as_array(pw, shape=(nw,)).copy() if nw else np.zeros(0, np.uint32)
            if windows:
                ps, pe = _lib.c_i32p(), _lib.c_i32p()
                check(lib.wepp_gen_reads_windows(h, ctypes.byref(ps), ctypes.byref(pe)))
                st = np.ctypeslib.as_array(ps, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int32)
                en = np.ctypeslib.as_array(pe, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int32)
        finally:
            lib.wepp_gen_reads_destroy(h)
        if windows:
            deg = (np.random.default_rng(seed).integers(1, max_degree + 1, n.value).astype(np.int32)
                   if max_degree > 1 else np.ones(n.value, np.int32))
            return EppReads(off, words, st, en, deg)
        return Reads(off, words)

    def close(self):
        if self._h:
            lib.wepp_gen_tree_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FlatView:
    """Host-only view of the flattened MAT (wepp_flat_*), for the CPU tests."""

    _DT = {"nkey": np.int64, "ev_meta": np.uint8, "ev_lb": np.uint8, "sp": np.uint8, "maxnest": np.uint8, "ix_nest": np.uint8}

    def __init__(self, tree, path=None):
        """Flattens `tree`, or -- path given, tree None -- reads an image written by save()."""
        self._tree = tree
        self._h = ctypes.c_void_p()
        if tree is None:
            check(lib.wepp_flat_load(os.fsencode(path), ctypes.byref(self._h)))
        else:
            d = tree.desc()
            check(lib.wepp_flat_create(ctypes.byref(d), ctypes.byref(self._h)))
        st = _lib.MatStats()
        cs = ctypes.c_uint32()
        check(lib.wepp_flat_scalars(self._h, ctypes.byref(st), ctypes.byref(cs)))
        self.stats = st
        self.cp_stride = cs.value

    @property
    def n_streams(self):
        return int(self.stats.n_streams)

    def save(self, path):
        """wepp_flat_save: the image as a file (one flatten per node: the other ranks FlatView(None, path))."""
        check(lib.wepp_flat_save(self._h, os.fsencode(path)))

    @classmethod
    def load(cls, path):
        return cls(None, path)

    def get(self, name, stream=None):
        """Host array `name`; stream fields take stream=i (default: whole-tree stream)."""
        base = name
        if stream is not None:
            name = f"{stream}:{name}" if isinstance(stream, str) else f"{int(stream)}:{name}"   # "w3" = window stream 3
        data = ctypes.c_void_p()
        cnt = ctypes.c_uint64()
        eb = ctypes.c_uint32()
        check(lib.wepp_flat_get(self._h, name.encode(), ctypes.byref(data), ctypes.byref(cnt), ctypes.byref(eb)))
        if cnt.value == 0:
            return np.zeros(0, np.uint32)
        if base in ("rq_pre", "rq_suf", "rq_dst", "ix_ent", "nrec", "ix_head"):
            width = 2 if base == "ix_head" else 8 if base == "ix_ent" else 4
            raw = np.ctypeslib.as_array(ctypes.cast(data, _lib.c_u32p), shape=(cnt.value * width,)).copy()
            return raw.reshape(-1, width)   # SegNode: base, rank, cnt, hu; IxEnt: node, end, word, up; NodeRec: base, rank, nstat, -; IxHead: off, first node
        if base == "blk_sum":
            raw = np.ctypeslib.as_array(ctypes.cast(data, _lib.c_u32p), shape=(cnt.value * 8,)).copy()
            return raw.reshape(-1, 8)   # base, rank, cnt, min_all, node0, nn, pad, pad
        dt = self._DT.get(base, np.uint32)
        cptr = ctypes.cast(data, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dt)))
        return np.ctypeslib.as_array(cptr, shape=(cnt.value,)).copy()

    def close(self):
        if self._h:
            lib.wepp_flat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def flatten_count():
    """Full flattens (Mat(tree), FlatView(tree)) this process has run."""
    return int(lib.wepp_debug_flatten_count())


def fitch_last_timing():
    """Wall time by phase (ms) of this thread's last Fitch-Sankoff run: rows prepared on the host, uploads,
    kernels, sort + decode + copy-out."""
    d = [ctypes.c_double() for _ in range(4)]
    check(lib.wepp_fitch_last_timing(*[ctypes.byref(x) for x in d]))
    return dict(prep_ms=d[0].value, upload_ms=d[1].value, kernels_ms=d[2].value, output_ms=d[3].value)


class FitchPlan:
    """wepp_fitch_plan_*: the tree-dependent part of the Fitch-Sankoff pass, done once."""

    def __init__(self, tree, device=0):
        self._h = ctypes.c_void_p()
        self._keep = tree
        d = tree.desc()
        check(lib.wepp_fitch_plan_create(ctypes.byref(d), int(device), ctypes.byref(self._h)))

    def run(self, site_ref, var_off, var_node, var_nuc, capacity=None):
        return fitch_sites(None, site_ref, var_off, var_node, var_nuc, capacity=capacity, plan=self)

    def close(self):
        if self._h:
            lib.wepp_fitch_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fitch_sites(tree, site_ref, var_off, var_node, var_nuc, device=0, capacity=None, plan=None):
    """Per-site Fitch-Sankoff (wepp_fitch_sites, or wepp_fitch_plan_run with plan=): returns arrays
    (site, node id, par_nuc, mut_nuc) of the mutations mapper_body would add, rows in order, BFS order
    inside a row."""
    site_ref = np.ascontiguousarray(site_ref, np.uint8)
    var_off = np.ascontiguousarray(var_off, np.uint32)
    var_node = np.ascontiguousarray(var_node, np.uint32)
    var_nuc = np.ascontiguousarray(var_nuc, np.uint8)
    n_sites = int(site_ref.shape[0])
    cap = int(capacity if capacity is not None else max(1024, 4 * int(var_off[-1]) + n_sites))
    d = tree.desc() if plan is None else None
    while True:
        o_site = np.zeros(cap, np.uint32)
        o_node = np.zeros(cap, np.uint32)
        o_par = np.zeros(cap, np.uint8)
        o_mut = np.zeros(cap, np.uint8)
        n_out = ctypes.c_uint64()
        vn = var_node if var_node.size else np.zeros(1, np.uint32)
        vc = var_nuc if var_nuc.size else np.zeros(1, np.uint8)
        if plan is None:
            rc = lib.wepp_fitch_sites(ctypes.byref(d), int(device), n_sites, _ptr(site_ref), _ptr(var_off), _ptr(vn),
                                      _ptr(vc), cap, ctypes.byref(n_out), _ptr(o_site), _ptr(o_node), _ptr(o_par),
                                      _ptr(o_mut))
        else:
            rc = lib.wepp_fitch_plan_run(plan._h, n_sites, _ptr(site_ref), _ptr(var_off), _ptr(vn), _ptr(vc), cap,
                                         ctypes.byref(n_out), _ptr(o_site), _ptr(o_node), _ptr(o_par), _ptr(o_mut))
        if rc == 4 and n_out.value > cap and capacity is None:
            cap = int(n_out.value)
            continue
        check(rc)
        n = int(n_out.value)
        return o_site[:n], o_node[:n], o_par[:n], o_mut[:n]


class PlacementResult:
    def __init__(self, best_bfs_j, score, num_best, flags):
        self.best_bfs_j = best_bfs_j
        self.score = score
        self.num_best = num_best
        self.flags = flags

    @property
    def has_unique(self):
        return (self.flags & 1).astype(np.uint8)


class Mat:
    """Flattened MAT resident in one GPU's HBM (wepp_mat_t)."""

    def __init__(self, tree, device=0, flat=None):
        """flat = a FlatView of the same tree: upload that image (wepp_mat_upload) instead of flattening again
        (one flatten per host, one upload per device)."""
        self._h = ctypes.c_void_p()
        self._keep = tree
        if flat is not None:
            check(lib.wepp_mat_upload(flat._h, int(device), ctypes.byref(self._h)))
        else:
            d = tree.desc()
            check(lib.wepp_mat_create(ctypes.byref(d), int(device), ctypes.byref(self._h)))
        self.device = int(device)
        st = _lib.MatStats()
        check(lib.wepp_mat_get_stats(self._h, ctypes.byref(st)))
        self.stats = st
        self.n_nodes = int(st.n_nodes)

    def bfs_order(self):
        out = np.zeros(self.n_nodes, dtype=np.uint32)
        check(lib.wepp_mat_bfs_order(self._h, _ptr(out)))
        return out

    def set_tile_reads(self, t):
        check(lib.wepp_mat_set_tile_reads(self._h, int(t)))

    def set_use_crowns(self, enable):
        """Work skipping on (default) / off; speed only, never results."""
        check(lib.wepp_mat_set_use_crowns(self._h, 1 if enable else 0))

    def set_use_walk(self, enable):
        """Per-read walks on (default) / off (every read placed by a sweep); speed only, never results."""
        check(lib.wepp_mat_set_use_walk(self._h, 1 if enable else 0))

    def set_use_seeds(self, enable):
        """Whole-genome samples by chunk signatures (default) or by tile sweeps; speed only, never results."""
        check(lib.wepp_mat_set_use_seeds(self._h, 1 if enable else 0))

    def last_seeds(self, detail=False):
        """(samples seeded, chunks evaluated, chunks in all) since the last timing reset; detail=True adds the most
        chunks one sample evaluated and the samples by chunks evaluated (<= 1, 4, 16, 64, 256, 1024, 4096, more)."""
        a, b, c, d = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        hist = np.zeros(8, np.uint64)
        check(lib.wepp_mat_last_seeds(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d), _ptr(hist)))
        if detail:
            return a.value, b.value, c.value, d.value, hist.tolist()
        return a.value, b.value, c.value

    def set_pipeline(self, sub_batches):
        """Sub-batches place_batch cuts a large batch into (0 = default); speed only, never results."""
        check(lib.wepp_mat_set_pipeline(self._h, int(sub_batches)))

    def place_batch(self, reads, per_node_scores=False, out=None):
        """Host buffers in/out: wepp_place_batch.  out = a PlacementResult of a previous call with the
        same number of reads, to reuse its arrays."""
        n = reads.n_reads
        if out is not None and out.score.shape[0] == n:
            bj, sc, nb, fl = out.best_bfs_j, out.score, out.num_best, out.flags
        else:
            bj = np.zeros(n, np.uint32)
            sc = np.zeros(n, np.int32)
            nb = np.zeros(n, np.uint32)
            fl = np.zeros(n, np.uint32)
        pns = np.zeros((n, self.n_nodes), np.int32) if per_node_scores else None
        rw = reads.read_word if reads.read_word.size else np.zeros(1, np.uint32)
        check(lib.wepp_place_batch(self._h, _ptr(reads.read_off), _ptr(rw), n, _ptr(bj), _ptr(sc), _ptr(nb),
                                   _ptr(fl), _ptr(pns) if per_node_scores else None))
        res = PlacementResult(bj, sc, nb, fl)
        if per_node_scores:
            res.per_node_scores = pns
        return res

    def best_nodes(self, reads, res):
        """wepp_best_nodes: best_j_vec of every read (BFS indices of all optimal nodes, ascending) as a list of
        arrays; res = the PlacementResult of place_batch for the same reads."""
        n = reads.n_reads
        off = np.zeros(n + 1, np.uint64)
        cap = int(res.num_best.astype(np.uint64).sum())
        nodes = np.zeros(max(cap, 1), np.uint32)
        rw = reads.read_word if reads.read_word.size else np.zeros(1, np.uint32)
        check(lib.wepp_best_nodes(self._h, _ptr(reads.read_off), _ptr(rw), n, _ptr(np.ascontiguousarray(res.score, np.int32)),
                                  _ptr(np.ascontiguousarray(res.num_best, np.uint32)), _ptr(off), _ptr(nodes), cap))
        return [nodes[int(off[r]):int(off[r + 1])] for r in range(n)]

    def excess_mutations(self, reads, pair_read, pair_bfs_j):
        """wepp_excess_mutations: per (read, node) pair the list of (position, ref, par, mut)."""
        pr = np.ascontiguousarray(pair_read, np.uint32); pj = np.ascontiguousarray(pair_bfs_j, np.uint32)
        n = int(pr.shape[0])
        off = np.zeros(n + 1, np.uint64)
        rw = reads.read_word if reads.read_word.size else np.zeros(1, np.uint32)
        cap = 0
        while True:
            pos = np.zeros(max(cap, 1), np.int32); ref = np.zeros(max(cap, 1), np.uint8)
            par = np.zeros(max(cap, 1), np.uint8); mut = np.zeros(max(cap, 1), np.uint8)
            rc = lib.wepp_excess_mutations(self._h, _ptr(reads.read_off), _ptr(rw), reads.n_reads, n, _ptr(pr), _ptr(pj),
                                           _ptr(off), _ptr(pos), _ptr(ref), _ptr(par), _ptr(mut), cap)
            if rc == 4 and int(off[n]) > cap:
                cap = int(off[n])
                continue
            check(rc)
            break
        return [list(zip(pos[int(off[i]):int(off[i + 1])].tolist(), ref[int(off[i]):int(off[i + 1])].tolist(),
                         par[int(off[i]):int(off[i + 1])].tolist(), mut[int(off[i]):int(off[i + 1])].tolist()))
                for i in range(n)]

    def dfs_order(self):
        """Caller node id of the haplotype with arena (pre-order) index k."""
        out = np.zeros(self.n_nodes, np.uint32)
        check(lib.wepp_mat_dfs_order(self._h, _ptr(out)))
        return out

    def epp_map(self, reads, genome_size, max_cached_epp=2048, want_counts=True, want_divergence=True,
                want_lists=True, epp_capacity=None):
        """wepp_epp_map: WEPP's cartesian_map for an EppReads batch; haplotypes by arena index."""
        R, n = reads.n_reads, self.n_nodes
        mp = np.zeros(max(R, 1), np.int32); mult = np.zeros(max(R, 1), np.uint32)
        score = np.zeros(n, np.float64)
        counts = np.zeros((n, 50), np.int32) if want_counts else None
        div = np.zeros(n, np.float64) if want_divergence else None
        cap = int(epp_capacity if epp_capacity is not None else max_cached_epp * max(R, 1))
        eoff = np.zeros(R + 1, np.uint64) if want_lists else None
        enodes = np.zeros(max(cap, 1), np.uint32) if want_lists else None
        rw = reads.read_word if reads.read_word.size else np.zeros(1, np.uint32)
        rd = _lib.EppReadsC(R, _ptr(reads.read_off).value, _ptr(rw).value, _ptr(reads.start).value,
                            _ptr(reads.end).value, _ptr(reads.degree).value)
        o = _lib.EppOutC(_ptr(mp).value, _ptr(mult).value, _ptr(eoff).value if want_lists else None,
                         _ptr(enodes).value if want_lists else None, cap, _ptr(score).value,
                         _ptr(counts).value if want_counts else None, _ptr(div).value if want_divergence else None)
        rc = lib.wepp_epp_map(self._h, ctypes.byref(rd), int(genome_size), int(max_cached_epp), ctypes.byref(o))
        if rc == 4 and want_lists and int(eoff[R]) > cap:
            # the guess was short: everything else is complete, the lists wait on the handle (wepp_epp_fetch_lists)
            enodes = np.zeros(int(eoff[R]), np.uint32)
            check(lib.wepp_epp_fetch_lists(self._h, _ptr(enodes), int(eoff[R])))
        else:
            check(rc)
        out = dict(max_parsimony=mp[:R], multiplicity=mult[:R], score=score, counts=counts, divergence=div)
        if want_lists:
            out["epp_off"] = eoff
            out["epp_nodes"] = enodes[: int(eoff[R])]
        return out

    def imputed_mutations(self, reads, best_bfs_j):
        """Per read: list of (position, nucleotide mask) imputed for its ambiguous entries at
        the chosen node (column 4 of placement_stats.tsv): wepp_imputed_mutations."""
        n = reads.n_reads
        a = (reads.read_word >> 24) & 15
        cap = int((((reads.read_word >> 28) & 1) == 0).__and__((a & (a - 1)) != 0).sum()) if reads.read_word.size else 0
        off = np.zeros(n + 1, np.uint32)
        pos = np.zeros(max(cap, 1), np.int32)
        nuc = np.zeros(max(cap, 1), np.uint8)
        bj = np.ascontiguousarray(best_bfs_j, dtype=np.uint32)
        rw = reads.read_word if reads.read_word.size else np.zeros(1, np.uint32)
        check(lib.wepp_imputed_mutations(self._h, _ptr(reads.read_off), _ptr(rw), n, _ptr(bj), _ptr(off), _ptr(pos),
                                         _ptr(nuc), cap))
        return [list(zip(pos[off[r]:off[r + 1]].tolist(), nuc[off[r]:off[r + 1]].tolist())) for r in range(n)]

    def place_batch_device(self, d_read_off, d_read_word, n_reads, n_read_words, d_best, d_score, d_num_best,
                           d_flags, stream=0):
        """Device pointers (ints) in/out: wepp_place_batch_device; not synchronised."""
        check(lib.wepp_place_batch_device(self._h, d_read_off, d_read_word, int(n_reads), int(n_read_words),
                                          d_best, d_score, d_num_best, d_flags, stream or None))

    def last_tiers(self, n_reads):
        """Sweep stream every read of the last placement call was routed to (diagnostic)."""
        out = np.zeros(int(n_reads), np.uint8)
        check(lib.wepp_mat_last_tiers(self._h, _ptr(out), int(n_reads)))
        return out

    def last_plans(self, n_reads):
        """(class, stream) of every read of the last placement call: class = PLAN_* (how it was placed), stream =
        index of the sweep stream (or of the genome window for PLAN_WIN)."""
        cls = np.zeros(int(n_reads), np.uint8)
        st = np.zeros(int(n_reads), np.uint8)
        check(lib.wepp_mat_last_plans(self._h, _ptr(cls), _ptr(st), int(n_reads)))
        return cls, st

    def last_crowns(self, n_reads):
        """(window, crown of the window) of every read of the last call that walked a window crown; 255 otherwise."""
        win = np.zeros(int(n_reads), np.uint8)
        cr = np.zeros(int(n_reads), np.uint8)
        check(lib.wepp_mat_last_crowns(self._h, _ptr(win), _ptr(cr), int(n_reads)))
        return win, cr

    def last_walk(self):
        """(reads of the last call placed by the per-read walk, walk loop iterations since timing_reset)."""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib.wepp_mat_last_walk(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def timing_reset(self):
        check(lib.wepp_mat_timing_reset(self._h))

    def last_timing(self):
        """(mean k_sweep ms per placement call, calls averaged, stream sweeps and
        algorithmic bytes of the last call)."""
        ms = ctypes.c_float()
        nl = ctypes.c_uint32()
        passes = ctypes.c_uint64()
        bpp = ctypes.c_uint64()
        check(lib.wepp_mat_last_timing(self._h, ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(passes),
                                       ctypes.byref(bpp)))
        return ms.value, nl.value, passes.value, bpp.value

    def close(self):
        if self._h:
            lib.wepp_mat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def epp_last_timing():
    """Device time by phase (ms) and work counters of this thread's last Mat.epp_map call."""
    d = [ctypes.c_double() for _ in range(4)]
    ev, se = ctypes.c_uint64(), ctypes.c_uint64()
    g, j = ctypes.c_uint32(), ctypes.c_uint32()
    check(lib.wepp_epp_last_timing(*[ctypes.byref(x) for x in d], ctypes.byref(ev), ctypes.byref(se), ctypes.byref(g),
                                   ctypes.byref(j)))
    return dict(select_ms=d[0].value, sweep1_ms=d[1].value, sweep2_ms=d[2].value, finish_ms=d[3].value,
                events_swept=ev.value, stream_events=se.value, groups=g.value, jobs=j.value)
