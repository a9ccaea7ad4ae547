"""Pure-Python model of the per-read walk (k_walk in wepp_amd/csrc/walk_kernels.hip): a read visits only
the events of the positions it lists -- taken from the stream's position index in stream order -- and
answers everything between two events with a range query over the statically eligible nodes (sparse table
for "can anything in there matter", segment tree for the exact (score, rank, count)).  Used by the CPU
tests to check the index structures the flattener builds and the walk's arithmetic against the oracle.
NOT a product path."""
import numpy as np

from sweep_model import NS_CNT, NS_ELIG0, NS_LEAF, NS_MASKED, NS_ROOT, enter_delta, own_adjust

RQ_BLK = 16

IX_NONE = 0xFFFFFFFF
SP_NONE, SP_CLAMP = 255, 254
SCORE_INF = 0x3FFFFFFF


class WalkModel:
    def __init__(self, flat, stream=None):
        self.stream = flat.n_streams - 1 if stream is None else stream
        for name in ("nkey", "nstat", "rq_pre", "rq_suf", "rq_dst", "sp"):
            setattr(self, name, flat.get(name, self.stream))
        head, ent, nrec = flat.get("ix_head", self.stream), flat.get("ix_ent", self.stream), flat.get("nrec", self.stream)
        self.ix_off = head[:, 0]
        self.ix_node, self.ix_end, self.ix_word, self.ix_up = ent[:, 0], ent[:, 1], ent[:, 2], ent[:, 3]
        assert (head[:-1, 1] == ent[head[:-1, 0], 0]).all()            # IxHead::first_node
        real = ent[:, 0] != IX_NONE
        assert (ent[:-1, 4][real[:-1]] == ent[1:, 0][real[:-1]]).all()  # IxEnt::next_node
        assert (ent[real, 5].astype(np.int32).astype(np.int64) == (self.nkey[ent[real, 0]] >> 32)).all()
        pre = flat.get("ix_pre", self.stream)
        self.has_pre = bool(len(pre) and pre[0])          # top byte of IxEnt::rank = pre-test of the range the entry ends
        rank_of = ent[:, 6] & (0xFFFFFF if self.has_pre else 0xFFFFFFFF)
        self.ix_pre_byte = ent[:, 6] >> 24
        self.n_by_entry = 0
        assert (rank_of[real].astype(np.int64) == (self.nkey[ent[real, 0]] & 0xFFFFFFFF)).all()
        assert (ent[real, 7] == self.nstat[ent[real, 0]]).all()
        assert (nrec[:, 0].astype(np.int64) == (self.nkey >> 32)).all() and (nrec[:, 2] == self.nstat).all()
        assert (nrec[:, 1].astype(np.int64) == (self.nkey & 0xFFFFFFFF)).all()
        self.rank2bfs = flat.get("rank2bfs")
        self.n = len(self.nkey)
        self.nblk = (self.n + RQ_BLK - 1) // RQ_BLK
        self.levels = len(self.sp) // self.n
        self.maxnest = flat.get("ix_nest", self.stream)
        self.rank2dfs = flat.get("rank2dfs")
        self.dfs2bfs = flat.get("dfs2bfs")
        self.max_stack = 0
        self.n_exact = 0
        self.n_segments = 0

    # ---- range queries -------------------------------------------------------------------
    def range_min(self, a, b):
        """The walk's pre-test: ONE entry of the sparse table, of the first level whose span from `a` reaches
        b -- the minimum of a superset of [a, b) (never larger than the true minimum)."""
        ln = b - a
        l = (ln - 1).bit_length()            # smallest l with 2^l >= ln
        assert l < self.levels
        return int(self.sp[l * self.n + a])

    def range_exact(self, a, b):
        """(score, rank, count) of the best statically eligible node of [a, b): suffix of a's block, the
        disjoint sparse table over the whole blocks in between, prefix of the last node's block."""
        none = (SCORE_INF, 0xFFFFFFFF, 0, 0)

        def comb(x, y):
            if y[0] < x[0]:
                return y
            if y[0] == x[0]:
                return (x[0], min(x[1], y[1]), x[2] + y[2], y[3] if y[1] < x[1] else x[3])
            return x

        def row(arr, i):
            return tuple(int(v) for v in arr[i])
        last = b - 1
        ba, bl = a // RQ_BLK, last // RQ_BLK
        if ba == bl:
            best = none
            for i in range(a, b):                     # inside one block: node by node
                st = int(self.nstat[i])
                if st & NS_ELIG0:
                    k = int(self.nkey[i])
                    hu = 0 if st & NS_ROOT else 1 if st & NS_MASKED else int(((st >> 14) & NS_CNT) < (st & NS_CNT))
                    best = comb(best, (k >> 32, k & 0xFFFFFFFF, 1, hu))
            return best
        best = comb(row(self.rq_suf, a), row(self.rq_pre, last))
        lo, hi = ba + 1, bl - 1
        if lo == hi:
            best = comb(best, row(self.rq_dst, lo))
        elif lo < hi:
            L = (lo ^ hi).bit_length() - 1
            best = comb(best, comb(row(self.rq_dst, L * self.nblk + lo), row(self.rq_dst, L * self.nblk + hi)))
        return best

    def stack_bound(self, S):
        return sum(int(self.maxnest[p]) if p < len(self.maxnest) else 0 for (p, _, _, _) in S)

    # ---- the walk ------------------------------------------------------------------------
    def chunk_bounds(self, S, C):
        """Node ranges of the C jobs a read's walk is cut into: quantiles of its longest list."""
        npos = len(self.ix_off) - 1
        best_len, best_off = 0, 0
        for (p, _, _, _) in S:
            if p < npos:
                ln = int(self.ix_off[p + 1]) - int(self.ix_off[p]) - 1
                if ln > best_len:
                    best_len, best_off = ln, int(self.ix_off[p])
        assert C == 1 or best_len >= C
        return [0] + [int(self.ix_node[best_off + c * best_len // C]) for c in range(1, C)] + [self.n]

    def events_of(self, S):
        npos = len(self.ix_off) - 1
        return sum(int(self.ix_off[p + 1]) - int(self.ix_off[p]) - 1 for (p, _, _, _) in S if p < npos)

    def place_chunked(self, S, root_score, C):
        """The walk cut into C independent jobs, combined like k_finalize_jobs."""
        nb = self.chunk_bounds(S, C)
        bs, br, cnt, bhu = root_score + 1, 0xFFFFFFFF, 0, 0
        for c in range(C):
            s, r, k, hu = self.place(S, root_score, nb[c], nb[c + 1])
            if k == 0:
                continue
            if s < bs:
                bs, br, cnt, bhu = s, r, k, hu
            elif s == bs:
                cnt += k
                if r < br:
                    br, bhu = r, hu
        return bs, br, cnt, bhu

    def place(self, S, root_score, start=0, stop_at=None):
        """S = [(pos, ref, mut, missing)] sorted by position.  Returns (score, rank, count) over the
        nodes [start, stop_at) (default: the whole stream)."""
        n_end = self.n if stop_at is None else stop_at
        c = sum(1 for (_, sref, a, missing) in S if not missing and (a & sref) == 0)
        bs, br, cnt, bhu = root_score + 1, 0xFFFFFFFF, 0, 0
        npos = len(self.ix_off) - 1
        cur = [int(self.ix_off[p]) if p < npos else None for (p, _, _, _) in S]
        stack = []                                # (end, delta) of the open intervals
        if start > 0:
            # state of a sequential walk when it reaches node `start`: every list's cursor at its first
            # entry >= start (binary search) and the intervals open there -- the predecessor entry if it
            # is still open, else the first open one up its chain of enclosing entries, and every entry
            # enclosing that one
            opened = []
            for j, (p, _, _, _) in enumerate(S):
                if cur[j] is None:
                    continue
                lo, hi = int(self.ix_off[p]), int(self.ix_off[p + 1]) - 1      # (the sentinel stays out)
                while lo < hi:
                    mid = (lo + hi) // 2
                    if int(self.ix_node[mid]) < start:
                        lo = mid + 1
                    else:
                        hi = mid
                cur[j] = lo
                e = lo - 1 if lo > int(self.ix_off[p]) else IX_NONE
                while e != IX_NONE and int(self.ix_end[e]) <= start:
                    e = int(self.ix_up[e])
                while e != IX_NONE:
                    node, end, wd = int(self.ix_node[e]), int(self.ix_end[e]), int(self.ix_word[e])
                    assert node < start < end
                    d = enter_delta(wd, S[j])
                    c += d
                    if d != 0:
                        opened.append((end, d))
                    e = int(self.ix_up[e])
            opened.sort(key=lambda t: -t[0])          # outermost first: the innermost interval ends first
            stack = opened
            self.max_stack = max(self.max_stack, len(stack))
        head = [int(self.ix_node[q]) if q is not None else IX_NONE for q in cur]
        pos = start

        def take(score, rank, k, hu):
            nonlocal bs, br, cnt, bhu
            if score < bs:
                bs, br, cnt, bhu = score, rank, k, hu
            elif score == bs:
                cnt += k
                if rank < br:
                    br, bhu = rank, hu

        def segment(a, b, entry=None):
            self.n_segments += 1
            if self.has_pre and entry is not None:
                # the range ends at the node of list entry `entry`: its byte stands for a superset of [a, b)
                pb = int(self.ix_pre_byte[entry])
                if pb == SP_NONE or (pb < SP_CLAMP and pb + c > bs):
                    self.n_by_entry += 1
                    return
            m = self.range_min(a, b)
            if m != SP_NONE and (m >= SP_CLAMP or m + c <= bs):
                self.n_exact += 1
                base, rank, k, hu = self.range_exact(a, b)
                if k and base + c <= bs:
                    take(base + c, rank, k, hu)

        while True:
            i_next = min(head) if head else IX_NONE
            e_next = stack[-1][0] if stack else IX_NONE
            stop = min(i_next, e_next, n_end)
            if stop > pos:
                at_node = i_next < e_next and i_next < n_end
                segment(pos, stop, cur[head.index(i_next)] if at_node else None)
                pos = stop
            if pos >= n_end:
                break
            if e_next <= i_next:
                c -= stack.pop()[1]
                continue
            # the node at i_next: every listed mutation it carries
            node = i_next
            key, st = int(self.nkey[node]), int(self.nstat[node])
            base, rank = key >> 32, key & 0xFFFFFFFF
            adj = dcom = dsum = 0
            end = None
            for j in range(len(S)):
                if head[j] != node:
                    continue
                q = cur[j]
                w, end = int(self.ix_word[q]), int(self.ix_end[q])
                a1, a2 = own_adjust(w, S[j])
                adj += a1
                dcom += a2
                if end > node + 1 or node == 0:            # descendants (the root also applies them to itself)
                    dsum += enter_delta(w, S[j])
                cur[j] = q + 1
                head[j] = int(self.ix_node[q + 1])
            root, masked, leaf = bool(st & NS_ROOT), bool(st & NS_MASKED), bool(st & NS_LEAF)
            nmut, ncom0 = st & NS_CNT, (st >> 14) & NS_CNT
            hu = 0
            if root:
                elig, score = True, base + c + dsum
            elif masked:
                elig, score = False, 0
            else:
                score = base + c + adj
                ncom = ncom0 + dcom
                elig = (ncom > 0) if leaf else (ncom > 0 or ncom == nmut)
                hu = int(ncom < nmut)
            if elig and score <= bs:
                take(score, rank, 1, hu)
            if dsum != 0 and end > node + 1:
                stack.append((end, dsum))
                self.max_stack = max(self.max_stack, len(stack))
            c += dsum
            pos = node + 1
        return bs, br, cnt, bhu

    def place_all_pairs(self, S, root_score):
        """Model of k_walk_wave (wepp_amd/csrc/wave_kernels.hip): no walk -- every entry of the read's lists learns from
        an all-pairs pass what the sequential walk would know on arrival, with the kernel's wrapping differences."""
        M = 0xFFFFFFFF
        npos = len(self.ix_off) - 1
        c0 = sum(1 for (_, sref, a, missing) in S if not missing and (a & sref) == 0)
        ents = []                       # (node, end, d, adj, dcom) in the order of the concatenated lists
        for j, (p, _, _, _) in enumerate(S):
            if p >= npos:
                continue
            for q in range(int(self.ix_off[p]), int(self.ix_off[p + 1]) - 1):
                node, end, w = int(self.ix_node[q]), int(self.ix_end[q]), int(self.ix_word[q])
                d = enter_delta(w, S[j]) if (end > node + 1 or node == 0) else 0
                a1, a2 = own_adjust(w, S[j])
                assert -2 <= d <= 2 and -1 <= a1 <= 1 and -1 <= a2 <= 1       # (the kernel packs them into bytes, biased by 2)
                ents.append((node, end, d, a1, a2))
        bs, br, cnt, bhu = root_score + 1, 0xFFFFFFFF, 0, 0

        def take(score, rank, k, hu):
            nonlocal bs, br, cnt, bhu
            if score < bs:
                bs, br, cnt, bhu = score, rank, k, hu
            elif score == bs:
                cnt += k
                if rank < br:
                    br, bhu = rank, hu

        def stretch(a, b, c):
            m = self.range_min(a, b)
            if m != SP_NONE and (m >= SP_CLAMP or m + c <= bs):
                base, rank, k, hu = self.range_exact(a, b)
                if k and base + c <= bs:
                    take(base + c, rank, k, hu)

        first_node = min((e[0] for e in ents), default=M)
        for mg, (n, e, _, _, _) in enumerate(ents):
            cb = cB = dsum = adj = dcom = 0
            stopA = stopB = M
            lower_same = lower_end = False
            sA = n + 1
            for g, (nl, el, dl, al, ml) in enumerate(ents):
                nl1, span = nl + 1, (el - nl - 1) & M
                if dl != 0:
                    if ((n - nl1) & M) < span:
                        cb += dl
                    if ((e - nl1) & M) < span:
                        cB += dl
                if nl == n:
                    dsum, adj, dcom = dsum + dl, adj + al, dcom + ml
                    lower_same |= g < mg
                lower_end |= el == e and g < mg
                stopA = min(stopA, (nl - sA) & M, (el - sA) & M)
                stopB = min(stopB, (2 * nl - 2 * e) & M, (2 * el - 1 - 2 * e) & M)
            if not lower_same:
                key, st = int(self.nkey[n]), int(self.nstat[n])
                base, rank = key >> 32, key & 0xFFFFFFFF
                root, masked, leaf = bool(st & NS_ROOT), bool(st & NS_MASKED), bool(st & NS_LEAF)
                nmut, ncom0 = st & NS_CNT, (st >> 14) & NS_CNT
                c = c0 + cb
                if root:
                    if base + c + dsum <= bs:
                        take(base + c + dsum, rank, 1, 0)
                elif not masked:
                    score, ncom = base + c + adj, ncom0 + dcom
                    elig = (ncom > 0) if leaf else (ncom > 0 or ncom == nmut)
                    if elig and score <= bs:
                        take(score, rank, 1, int(ncom < nmut))
                eA = min(sA + stopA, self.n)
                if sA < eA:
                    stretch(sA, eA, c0 + cb + dsum)
            if not lower_end:
                eB = self.n if stopB >= 0x80000000 else min(e + ((stopB + 1) >> 1), self.n)
                if e < eB:
                    stretch(e, eB, c0 + cB)
        if first_node != 0 and self.n > 0:
            stretch(0, min(first_node, self.n), c0)
        return bs, br, cnt, bhu

    def whole_stream(self):
        """The stream-wide aggregate (Stream::whole): what a read without any event in the stream takes."""
        return self.range_exact(0, self.n)

    def result(self, S, root_score, C=1):
        bs, br, cnt, hu = self.place(S, root_score) if C == 1 else self.place_chunked(S, root_score, C)
        return bs, int(self.rank2bfs[br]), cnt, hu
