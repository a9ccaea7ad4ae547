"""Pure-Python model of the GPU sweep (wepp_amd/csrc/sweep_kernels.hip) over
the flattened MAT, used by the CPU test-suite to check the flattener and the
closed form against the oracle when no GPU is present.  It is NOT a product
path (nothing in wepp_amd imports it) and it is only usable on small trees.
"""
import numpy as np

INF = 0x7FFFFFFF
SCORE_INF = 0x3FFFFFFF
NS_CNT = 0x3FFF
NS_LEAF, NS_MASKED, NS_ELIG0, NS_ROOT = 1 << 28, 1 << 29, 1 << 30, 1 << 31
W_EXIT, W_LEAF = 1 << 30, 1 << 31


def _tw(w):
    w = int(w)
    return w & 0xFFFFF, 1 << ((w >> 20) & 3), (w >> 22) & 15, (w >> 26) & 15  # pos, ref mask, par, mut


def f_state(x, tref, s):
    _, sref, a, missing = s
    c0 = 1 if (x != 0 and x != tref) else 0
    cs = 0 if missing else (1 if (a & (x if x else sref)) == 0 else 0)
    return cs - c0


def enter_delta(w, s):
    _, ref, par, mut = _tw(w)
    return f_state(mut, ref, s) - f_state(par, ref, s)


def own_adjust(w, s):
    _, ref, par, mut = _tw(w)
    _, sref, a, missing = s
    static_common = 1 if mut == ref else 0
    static_sub = (1 if (par != 0 and par != ref) else 0) if static_common else 0
    if missing:
        actual_common, actual_sub = 1, 0
    else:
        actual_common = 1 if (a & mut) != 0 else 0
        actual_sub = (1 if (a & (par if par else sref)) == 0 else 0) if actual_common else 0
    return static_sub - actual_sub, actual_common - static_common


class FlatModel:
    """Model of one sweep stream (default: the whole tree) of a flattened MAT."""

    def __init__(self, flat, stream=None):
        self.f = flat
        self.stream = flat.n_streams - 1 if stream is None else stream
        for name in ("node_woff", "words", "rank2dfs", "dfs2bfs", "dfs2id", "bfs2id"):
            setattr(self, name, flat.get(name))
        self.gnstat = flat.get("nstat")            # whole-tree stream == global DFS order
        for name in ("nkey", "nstat", "blk_node0", "blk_eoff", "blk_sum", "ev_word", "ev_meta", "ev_lb", "cp_off",
                     "cp_word"):
            setattr(self, name, flat.get(name, self.stream))
        self.NB = len(self.blk_node0) - 1
        self.N = len(self.nkey)
        # a window stream ("w<i>"): the whole tree as the reads of one genome window see it; elements stand for
        # ncnt nodes each (pseudo-nodes for the runs of nodes the window's reads cannot tell apart)
        self.window = isinstance(self.stream, str)
        self.ncnt = flat.get("ncnt", self.stream) if self.window else None
        self.tau = 0x7FFFFFFF if self.window else int(flat.stats.stream_tau[self.stream])
        self.cp_stride = max(1, (self.NB + 1023) // 1024)

    def _c_none(self, S):
        return sum(1 for (_, sref, a, missing) in S if not missing and (a & sref) == 0)

    def chunk_start_c(self, S, b0):
        """c at the first node of block b0 from the checkpoint words (b0 % cp_stride == 0)."""
        Sd = {s[0]: s for s in S}
        c = self._c_none(S)
        cpi = b0 // self.cp_stride
        for e in range(int(self.cp_off[cpi]), int(self.cp_off[cpi + 1])):
            w = self.cp_word[e]
            s = Sd.get(int(w) & 0xFFFFF)
            if s is not None:
                c += enter_delta(w, s)
        return c

    n_heavy = 0
    n_light = 0
    verify_prune = True   # evaluate pruned blocks anyway and assert that nothing could have won

    def place(self, S, b0=0, b1=None, c0=None, node_scores=None, trace_c=None, prune=True):
        """Sweep blocks [b0, b1) for one read S = [(pos, ref, mut, missing)].
        Returns (best score, best rank, count, c at the end).  node_scores
        (optional int array indexed by DFS idx) receives the -p mode value."""
        Sd = {s[0]: s for s in S}
        b1 = self.NB if b1 is None else b1
        c = self.chunk_start_c(S, b0) if c0 is None else c0
        bs, br, cnt = INF, 0xFFFFFFFF, 0
        for b in range(b0, b1):
            if trace_c is not None:
                trace_c[b] = c
            e0, e1 = int(self.blk_eoff[b]), int(self.blk_eoff[b + 1])
            n0 = int(self.blk_node0[b])
            nn = int(self.blk_node0[b + 1]) - n0
            hits = [e for e in range(e0, e1) if (int(self.ev_word[e]) & 0xFFFFF) in Sd]
            base, rank, sc, min_all = (int(x) for x in self.blk_sum[b][:4])
            if min_all >= 0x80000000:
                min_all -= 1 << 32
            if not hits and node_scores is None:
                if base != SCORE_INF:
                    s = base + c
                    if s < bs:
                        bs, br, cnt = s, rank, sc
                    elif s == bs:
                        cnt += sc
                        br = min(br, rank)
                continue
            cadd = [0] * nn
            adj = [0] * nn
            dcom = [0] * nn
            touched = [False] * nn
            net = 0
            H = 0
            lbmin = 0x3FFFFFFF
            for e in hits:
                # crown streams: per-event bound; whole-tree stream: the block minimum
                eager = (not self.window) and self.stream != self.f.n_streams - 1
                lbmin = min(lbmin, int(self.ev_lb[e]) if (eager and e < e0 + 128) else min(min_all, 0 if e >= e0 + 128 else min_all))
                w = int(self.ev_word[e])
                o = int(self.ev_meta[e]) & 63
                s = Sd[w & 0xFFFFF]
                d = enter_delta(w, s)
                if w & W_EXIT:
                    for i in range(o, nn):
                        cadd[i] -= d
                    net -= d
                    H += max(d, 0)          # nodes after the exit lose d: only d > 0 can lower a score
                else:
                    if not (w & W_LEAF):
                        is_root = (n0 + o) == 0
                        for i in range(o if is_root else o + 1, nn):
                            cadd[i] += d
                        net += d
                        # descendants gain d (only d < 0 lowers a score), the node itself takes no d but an
                        # adjustment of at least -1: no node is lowered by more than max(-d, 1)
                        H += max(-d, 1)
                    else:
                        # a leaf: only its own adjustment can lower it, by one, and only if the read shares
                        # the new allele and not the parent's
                        _, ref_, par_, mut_ = _tw(w)
                        _, sref_, a_, missing_ = s
                        lowers = (not missing_) and (a_ & mut_) != 0 and (a_ & (par_ if par_ else sref_)) == 0
                        H += 1 if lowers else 0     # (= actual_sub of own_adjust)
                    touched[o] = True
                    a, dc = own_adjust(w, s)
                    adj[o] += a
                    dcom[o] += dc
            pruned = (bool(hits) and prune and node_scores is None and (lbmin + c - H > bs)
                      and (base == SCORE_INF or base + c > bs))
            self.n_heavy += 0 if (pruned or not hits) else 1
            self.n_light += 1 if pruned else 0
            if pruned and not self.verify_prune:
                c += net
                continue
            for i in range(nn):
                key = int(self.nkey[n0 + i])
                st = int(self.nstat[n0 + i])
                base, rank = key >> 32, key & 0xFFFFFFFF
                nmut, ncom0 = st & NS_CNT, (st >> 14) & NS_CNT
                leaf, masked, root = bool(st & NS_LEAF), bool(st & NS_MASKED), bool(st & NS_ROOT)
                score = base + c + cadd[i]
                if root:
                    elig = True
                elif masked:
                    elig = False
                elif touched[i]:
                    score += adj[i]
                    ncom = ncom0 + dcom[i]
                    elig = (ncom > 0) if leaf else (ncom > 0 or ncom == nmut)
                else:
                    elig = bool(st & NS_ELIG0)
                if node_scores is not None:
                    node_scores[n0 + i] = score if elig else score + 1   # usher_mapper.cpp:500-505
                if elig and pruned:
                    assert score > bs, "pruning bound violated"
                if elig and not pruned:
                    # (a window stream that is the window's candidate crown has real nodes only: no counts)
                    k = int(self.ncnt[n0 + i]) if self.window and len(self.ncnt) else 1
                    if score < bs:
                        bs, br, cnt = score, rank, k
                    elif score == bs:
                        cnt += k
                        br = min(br, rank)
            c += net
        return bs, br, cnt, c

    def has_unique(self, S, rank):
        d = int(self.rank2dfs[rank])
        st = int(self.gnstat[d])
        if st & NS_ROOT:
            return 0
        if st & NS_MASKED:
            return 1
        Sd = {s[0]: s for s in S}
        ncom = (st >> 14) & NS_CNT
        for w in range(int(self.node_woff[d]), int(self.node_woff[d + 1])):
            s = Sd.get(int(self.words[w]) & 0xFFFFF)
            if s is not None:
                ncom += own_adjust(self.words[w], s)[1]
        return 1 if ncom < (st & NS_CNT) else 0

    def place_full(self, S, nchunks=1):
        """Whole placement the way the kernels do it: chunks + finalize."""
        ncp = len(self.cp_off) - 1
        nchunks = max(1, min(nchunks, ncp))
        cps = (ncp + nchunks - 1) // nchunks
        bpc = cps * self.cp_stride
        bs, br, cnt = INF, 0xFFFFFFFF, 0
        b = 0
        while b < self.NB:
            s, r, c_, _ = self.place(S, b, min(self.NB, b + bpc))
            if s < bs:
                bs, br, cnt = s, r, c_
            elif s == bs:
                cnt += c_
                br = min(br, r)
            b += bpc
        d = int(self.rank2dfs[br])
        return dict(score=bs, num_best=cnt, best_j=int(self.dfs2bfs[d]), has_unique=self.has_unique(S, br))


def theta(flat, S):
    """score(root) + |S|: no node whose static score exceeds it can win or tie
    (k_route in route_kernels.hip)."""
    words, woff = flat.get("words"), flat.get("node_woff")
    root_base = int(flat.get("nkey")[0]) >> 32
    Sd = {s[0]: s for s in S}
    c = sum(1 for (_, sref, a, missing) in S if not missing and (a & sref) == 0)
    for w in range(int(woff[0]), int(woff[1])):
        s = Sd.get(int(words[w]) & 0xFFFFF)
        if s is not None:
            c += enter_delta(words[w], s)
    return root_base + c + len(S)


class TieredModel:
    """Routes a read to the smallest crown stream covering theta, like k_route."""

    def __init__(self, flat):
        self.flat = flat
        self.models = [FlatModel(flat, i) for i in range(flat.n_streams)]

    def route(self, S):
        th = theta(self.flat, S)
        for i, m in enumerate(self.models[:-1]):
            if th <= m.tau:
                return i
        return len(self.models) - 1

    def place_full(self, S, nchunks=1):
        return self.models[self.route(S)].place_full(S, nchunks)
