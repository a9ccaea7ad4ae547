"""Python model of the seed bound (wepp_amd/csrc/seed_kernels.hip) over the arrays the flattener
builds: the chunks of the whole-tree stream, their signatures, and the lower bound a sample's hard entries give for
every node of a chunk.  Checked against the oracle's per-node scores in tests/test_seed_model.py."""
import numpy as np


class SeedModel:
    def __init__(self, fv):
        st = fv.stats
        self.nch = int(st.seed_chunks)
        self.stride = int(st.seed_chunk_blocks)
        self.max_pos = int(st.max_position)
        self.built = self.nch > 0
        if not self.built:
            return
        node0 = fv.get("blk_node0")                      # [NB + 1] of the whole-tree stream: local = global DFS index
        nb = len(node0) - 1
        self.chunk_lo = np.array([node0[min(nb, c * self.stride)] for c in range(self.nch + 1)], np.int64)
        sig = fv.get("seed_sig")
        self.row_words = len(sig) // (self.max_pos + 2)
        self.sig = sig.reshape(self.max_pos + 2, self.row_words)
        self.dfs2bfs = fv.get("dfs2bfs")

    def chunk_of_dfs(self, d):
        return int(np.searchsorted(self.chunk_lo, d, side="right") - 1)

    def nibble(self, pos, c):
        return (int(self.sig[pos, c >> 3]) >> ((c & 7) * 4)) & 15

    def hard_entries(self, S):
        """(entries with a signature row, entries beyond the tree's last mutated position): not missing, alleles
        excluding the entry's own reference base"""
        inside, beyond = [], 0
        for (p, ref, a, missing) in S:
            if missing or (a & ref):
                continue
            if p <= self.max_pos:
                inside.append((p, a))
            else:
                beyond += 1
        return inside, beyond

    def lower_bounds(self, S, max_hard=255):
        """|T| - H_T(C) for every chunk C (T = the first max_hard hard entries inside the tree + all beyond it)"""
        inside, beyond = self.hard_entries(S)
        inside = inside[:max_hard]
        sT = len(inside) + beyond
        H = np.zeros(self.nch, np.int64)
        for (p, a) in inside:
            for c in range(self.nch):
                if self.nibble(p, c) & a:
                    H[c] += 1
        return sT - H
