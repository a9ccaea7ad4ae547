"""ctypes bridge to oracle/mapper2_oracle.c (the CPU restatement of the
reference).  Test infrastructure: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this module.  The shared object is built
on demand with gcc into oracle/_build/ (git-ignored)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "oracle", f) for f in ("mapper2_oracle.c", "epp_oracle.c", "incremental_oracle.c")]
DEPS = SRCS + [os.path.join(ROOT, "oracle", "oracle_tree.h")]
OUT_DIR = os.path.join(ROOT, "oracle", "_build")
OUT = os.path.join(OUT_DIR, "liboracle.so")


def build(force=False):
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(f) for f in DEPS):
        os.makedirs(OUT_DIR, exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-DNDEBUG", "-shared", "-fPIC", "-o", OUT] + SRCS + ["-lpthread"])
    return OUT


class OracleResult(ctypes.Structure):
    _fields_ = [
        ("score", ctypes.c_int32),
        ("num_best", ctypes.c_uint32),
        ("best_j", ctypes.c_uint32),
        ("best_node_id", ctypes.c_int32),
        ("has_unique", ctypes.c_uint32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.oracle_tree_build.restype = ctypes.c_void_p
        L.oracle_tree_build.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6
        L.oracle_tree_free.argtypes = [ctypes.c_void_p]
        L.oracle_place_sample.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [
            ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(OracleResult), ctypes.c_void_p]
        L.oracle_place_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 6 + [ctypes.c_int]
        L.oracle_imputed_at_node.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [
            ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_excess_at_node.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [
            ctypes.c_uint32, ctypes.c_int] + [ctypes.c_void_p] * 4
        L.oracle_mapper_body.argtypes = [ctypes.c_void_p, ctypes.c_uint8, ctypes.c_int] + [ctypes.c_void_p] * 5
        L.oracle_place_batch_nodepar.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 6 + [ctypes.c_int]
        L.oracle_epp_map.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32] + [ctypes.c_void_p] * 12 + [
            ctypes.c_uint64] + [ctypes.c_void_p] * 3
        L.oracle_epp_map_mt.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32] + [ctypes.c_void_p] * 12 + [
            ctypes.c_uint64] + [ctypes.c_void_p] * 3 + [ctypes.c_int]
        L.oracle_epp_distance.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.inc_tree_build.restype = ctypes.c_void_p
        L.inc_tree_build.argtypes = [ctypes.c_void_p]
        L.inc_tree_free.argtypes = [ctypes.c_void_p]
        L.inc_place_sample.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5 + [
            ctypes.POINTER(OracleResult), ctypes.c_void_p]
        L.inc_place_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 6 + [ctypes.c_int]
        for f in ("oracle_tree_bfs_ids", "oracle_tree_dfs_ids", "oracle_tree_num_leaves"):
            getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleTree:
    """The reference's pointer tree rebuilt from a wepp_amd.Tree description."""

    def __init__(self, tree):
        L = lib()
        self.n = tree.n_nodes
        par = tree.mut_par if tree.mut_par is not None else tree.mut_ref
        self._keep = (tree.parent, tree.mut_off, tree.mut_pos, tree.mut_ref, par, tree.mut_mut)
        dummy = np.zeros(1, np.uint8)
        m = int(tree.mut_off[-1])
        self._h = L.oracle_tree_build(
            self.n, _p(tree.parent), _p(tree.mut_off),
            _p(tree.mut_pos if m else np.zeros(1, np.int32)), _p(tree.mut_ref if m else dummy),
            _p(par if m else dummy), _p(tree.mut_mut if m else dummy))
        if not self._h:
            raise ValueError("oracle_tree_build rejected the tree")

    def bfs_ids(self):
        out = np.zeros(self.n, np.int32)
        lib().oracle_tree_bfs_ids(self._h, _p(out))
        return out

    def dfs_ids(self):
        out = np.zeros(self.n, np.int32)
        lib().oracle_tree_dfs_ids(self._h, _p(out))
        return out

    def num_leaves(self):
        out = np.zeros(self.n, np.int64)
        lib().oracle_tree_num_leaves(self._h, _p(out))
        return out

    def place_sample(self, pos, ref, mut, missing, per_node_scores=False, want_best_vec=False):
        """One Missing_Sample against every node (usher_common.cpp:339-446)."""
        pos = np.ascontiguousarray(pos, np.int32)
        ref = np.ascontiguousarray(ref, np.uint8)
        mut = np.ascontiguousarray(mut, np.uint8)
        missing = np.ascontiguousarray(missing, np.uint8)
        n = len(pos)
        z32 = np.zeros(1, np.int32)
        z8 = np.zeros(1, np.uint8)
        res = OracleResult()
        nsd = np.zeros(self.n, np.int32) if per_node_scores else None
        bv = np.zeros(self.n, np.uint32) if want_best_vec else None
        lib().oracle_place_sample(
            self._h, n, _p(pos if n else z32), _p(ref if n else z8), _p(mut if n else z8), _p(missing if n else z8),
            1 if per_node_scores else 0, _p(nsd) if per_node_scores else None, ctypes.byref(res),
            _p(bv) if want_best_vec else None)
        out = dict(score=res.score, num_best=res.num_best, best_j=res.best_j, best_node_id=res.best_node_id,
                   has_unique=res.has_unique)
        if per_node_scores:
            out["node_scores"] = nsd
        if want_best_vec:
            out["best_j_vec"] = np.sort(bv[: res.num_best])
        return out

    def imputed_at_node(self, pos, ref, mut, missing, bfs_j):
        """node_imputed_mutations[bfs_j] of pass 2: list of (position, nucleotide mask)."""
        pos = np.ascontiguousarray(pos, np.int32); ref = np.ascontiguousarray(ref, np.uint8)
        mut = np.ascontiguousarray(mut, np.uint8); missing = np.ascontiguousarray(missing, np.uint8)
        n = len(pos)
        op = np.zeros(max(n, 1), np.int32); on = np.zeros(max(n, 1), np.uint8)
        z32 = np.zeros(1, np.int32); z8 = np.zeros(1, np.uint8)
        c = lib().oracle_imputed_at_node(self._h, n, _p(pos if n else z32), _p(ref if n else z8), _p(mut if n else z8),
                                         _p(missing if n else z8), int(bfs_j), _p(op), _p(on))
        return list(zip(op[:c].tolist(), on[:c].tolist()))

    def excess_at_node(self, pos, ref, mut, missing, bfs_j, capacity=4096):
        """node_excess_mutations[bfs_j] with compute_vecs: list of (position, ref, par, mut)."""
        pos = np.ascontiguousarray(pos, np.int32); ref = np.ascontiguousarray(ref, np.uint8)
        mut = np.ascontiguousarray(mut, np.uint8); missing = np.ascontiguousarray(missing, np.uint8)
        n = len(pos)
        op = np.zeros(capacity, np.int32); orf = np.zeros(capacity, np.uint8)
        opa = np.zeros(capacity, np.uint8); om = np.zeros(capacity, np.uint8)
        z32 = np.zeros(1, np.int32); z8 = np.zeros(1, np.uint8)
        c = lib().oracle_excess_at_node(self._h, n, _p(pos if n else z32), _p(ref if n else z8), _p(mut if n else z8),
                                        _p(missing if n else z8), int(bfs_j), capacity, _p(op), _p(orf), _p(opa), _p(om))
        return list(zip(op[:c].tolist(), orf[:c].tolist(), opa[:c].tolist(), om[:c].tolist()))

    def mapper_body(self, ref_nuc, var_node, var_nuc):
        """One VCF row through the Fitch-Sankoff mapper_body: list of (node id, par_nuc, mut_nuc)."""
        vn = np.ascontiguousarray(var_node, np.int32)
        vc = np.ascontiguousarray(var_nuc, np.uint8)
        on = np.zeros(self.n, np.int32); op = np.zeros(self.n, np.uint8); om = np.zeros(self.n, np.uint8)
        z32 = np.zeros(1, np.int32); z8 = np.zeros(1, np.uint8)
        c = lib().oracle_mapper_body(self._h, int(ref_nuc), len(vn), _p(vn if len(vn) else z32),
                                     _p(vc if len(vc) else z8), _p(on), _p(op), _p(om))
        assert c >= 0
        return list(zip(on[:c].tolist(), op[:c].tolist(), om[:c].tolist()))

    def place_batch(self, reads, nthreads=1, node_parallel=False):
        """reads: wepp_amd.Reads.  Returns a structured array.  node_parallel=True
        splits the NODE range over the threads like the reference's
        tbb::parallel_for (usher_common.cpp:386); otherwise reads are split."""
        from wepp_amd import unpack_read_word
        R = reads.n_reads
        pos, ref, mut, miss = unpack_read_word(reads.read_word)
        if pos.size == 0:
            pos = np.zeros(1, np.int32); ref = np.zeros(1, np.uint8); mut = np.zeros(1, np.uint8); miss = np.zeros(1, np.uint8)
        res = (OracleResult * max(R, 1))()
        fn = lib().oracle_place_batch_nodepar if node_parallel else lib().oracle_place_batch
        fn(self._h, R, _p(reads.read_off), _p(np.ascontiguousarray(pos)),
                                 _p(np.ascontiguousarray(ref)), _p(np.ascontiguousarray(mut)),
                                 _p(np.ascontiguousarray(miss)), ctypes.cast(res, ctypes.c_void_p), int(nthreads))
        arr = np.frombuffer(res, dtype=np.dtype([("score", "<i4"), ("num_best", "<u4"), ("best_j", "<u4"),
                                                 ("best_node_id", "<i4"), ("has_unique", "<u4")]), count=R).copy()
        return arr

    def epp_map(self, reads, genome_size, node_mapped=None, nthreads=1, want_counts=True):
        """wepp_filter::cartesian_map (src/WEPP/initial_filter.cpp:140-239) for a wepp_amd.EppReads
        batch.  Haplotype indices are pre-order (arena) indices.  nthreads > 1: the reads of a block are walked in
        parallel and folded in in read order (same sums in the same order); want_counts=False leaves out the
        [n_nodes, 50] read-count array (3.2 GB at 16 M nodes)."""
        from wepp_amd import unpack_read_word
        R = reads.n_reads
        pos, ref, mut, _ = unpack_read_word(reads.read_word)
        if pos.size == 0:
            pos = np.zeros(1, np.int32); ref = np.zeros(1, np.uint8); mut = np.zeros(1, np.uint8)
        pos = np.ascontiguousarray(pos); ref = np.ascontiguousarray(ref); mut = np.ascontiguousarray(mut)
        mp = np.zeros(max(R, 1), np.int32); mult = np.zeros(max(R, 1), np.uint32)
        eoff = np.zeros(R + 1, np.uint64)
        cap = 2048 * max(R, 1)
        enodes = np.zeros(cap, np.uint32)
        score = np.zeros(self.n, np.float64)
        counts = np.zeros((self.n, 50), np.int32) if want_counts else None
        div = np.zeros(self.n, np.float64)
        nm = None if node_mapped is None else np.ascontiguousarray(node_mapped, np.uint8)
        rc = lib().oracle_epp_map_mt(self._h, int(genome_size), R, _p(reads.read_off), _p(pos), _p(ref), _p(mut),
                                     _p(reads.start), _p(reads.end), _p(reads.degree), _p(nm) if nm is not None else None,
                                     _p(mp), _p(mult), _p(eoff), _p(enodes), cap, _p(score),
                                     _p(counts) if want_counts else None, _p(div), int(nthreads))
        if rc != 0:
            raise ValueError("oracle_epp_map failed: %d" % rc)
        return dict(max_parsimony=mp[:R], multiplicity=mult[:R], epp_off=eoff, epp_nodes=enodes[: int(eoff[R])],
                    score=score, counts=counts, divergence=div)

    def epp_distance(self, pos, ref, mut, start, end):
        """haplotype::mutation_distance(read) (src/WEPP/haplotype.hpp:123-173) for every haplotype."""
        pos = np.ascontiguousarray(pos, np.int32); ref = np.ascontiguousarray(ref, np.uint8)
        mut = np.ascontiguousarray(mut, np.uint8)
        n = len(pos)
        z32 = np.zeros(1, np.int32); z8 = np.zeros(1, np.uint8)
        out = np.zeros(self.n, np.int32)
        lib().oracle_epp_distance(self._h, n, _p(pos if n else z32), _p(ref if n else z8), _p(mut if n else z8),
                                  int(start), int(end), _p(out))
        return out

    def incremental(self):
        """The one-walk-per-read checker (oracle/incremental_oracle.c) over the same pointer tree."""
        return IncrementalTree(self)

    def close(self):
        if self._h:
            lib().oracle_tree_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


RESULT_DTYPE = np.dtype([("score", "<i4"), ("num_best", "<u4"), ("best_j", "<u4"), ("best_node_id", "<i4"),
                         ("has_unique", "<u4")])


class IncrementalTree:
    """oracle/incremental_oracle.c: one pre-order walk per read (SURVEY.md Appendix A closed form).
    Trusted only through tests/test_incremental.py (equality with the faithful restatement)."""

    def __init__(self, oracle_tree):
        self._ot = oracle_tree          # keeps the pointer tree (and its arrays) alive
        self.n = oracle_tree.n
        self._h = lib().inc_tree_build(oracle_tree._h)
        if not self._h:
            raise MemoryError("inc_tree_build failed")

    def place_sample(self, pos, ref, mut, missing, per_node_scores=False, want_best_vec=False):
        pos = np.ascontiguousarray(pos, np.int32); ref = np.ascontiguousarray(ref, np.uint8)
        mut = np.ascontiguousarray(mut, np.uint8); missing = np.ascontiguousarray(missing, np.uint8)
        n = len(pos)
        z32 = np.zeros(1, np.int32); z8 = np.zeros(1, np.uint8)
        res = OracleResult()
        nsd = np.zeros(self.n, np.int32) if per_node_scores else None
        bv = np.zeros(self.n, np.uint32) if want_best_vec else None
        rc = lib().inc_place_sample(self._h, n, _p(pos if n else z32), _p(ref if n else z8), _p(mut if n else z8),
                                    _p(missing if n else z8), _p(nsd) if per_node_scores else None, ctypes.byref(res),
                                    _p(bv) if want_best_vec else None)
        if rc:
            raise ValueError("inc_place_sample failed: %d" % rc)
        out = dict(score=res.score, num_best=res.num_best, best_j=res.best_j, best_node_id=res.best_node_id,
                   has_unique=res.has_unique)
        if per_node_scores:
            out["node_scores"] = nsd
        if want_best_vec:
            out["best_j_vec"] = np.sort(bv[: res.num_best])
        return out

    def place_batch(self, reads, nthreads=1):
        from wepp_amd import unpack_read_word
        R = reads.n_reads
        pos, ref, mut, miss = unpack_read_word(reads.read_word)
        if pos.size == 0:
            pos = np.zeros(1, np.int32); ref = np.zeros(1, np.uint8); mut = np.zeros(1, np.uint8); miss = np.zeros(1, np.uint8)
        res = (OracleResult * max(R, 1))()
        rc = lib().inc_place_batch(self._h, R, _p(reads.read_off), _p(np.ascontiguousarray(pos)),
                                   _p(np.ascontiguousarray(ref)), _p(np.ascontiguousarray(mut)),
                                   _p(np.ascontiguousarray(miss)), ctypes.cast(res, ctypes.c_void_p), int(nthreads))
        if rc:
            raise ValueError("inc_place_batch failed: %d" % rc)
        return np.frombuffer(res, dtype=RESULT_DTYPE, count=R).copy()

    def close(self):
        if self._h:
            lib().inc_tree_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
