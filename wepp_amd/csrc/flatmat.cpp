// flatmat.cpp -- Tree -> FlatMAT flattener (host, one-time).
//
// Reference behaviour folded into constants here (paths relative to
// /root/reference/src):
//   * BFS / DFS enumeration      mutation_annotated_tree.cpp:1115-1163
//   * get_num_leaves             mutation_annotated_tree.cpp:839-852
//   * ancestor walk "most recent mutation per position, masked skipped"
//                                usher_mapper.cpp:276-287
//   * root: every root mutation enters the ancestral set
//                                usher_mapper.cpp:266-271
//   * masked node mutation => has_unique, break
//                                usher_mapper.cpp:198-201
//   * a node mutation whose position is not in the sample is "common" iff it
//     reverts to the reference allele
//                                usher_mapper.cpp:244-260
//   * eligibility predicate      usher_mapper.cpp:455-456
//   * tie-break (num_leaves, then larger BFS index j)
//                                usher_mapper.cpp:484-487
#include "flatmat.hpp"
#include "flatten_options.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <numeric>
#include <thread>

namespace wepp {

namespace {
inline uint32_t cost0(uint32_t x, uint32_t ref) { return (x != 0 && x != ref) ? 1u : 0u; }
}  // namespace
namespace {

// The elements of a stream, in stream order.  A crown: the selected nodes themselves.  A window stream: the
// nodes that mutate a position of the window plus PSEUDO-NODES, one per maximal run of skipped nodes between
// two places where a read of the window can change its running c_S (a selected node, the end of a selected
// node's subtree) -- a pseudo-node carries the run's best statically eligible node and how many tie it.
struct StreamElems {
    std::vector<uint32_t> g;         // global DFS index of a real element, IX_NONE for a pseudo-node
    std::vector<int64_t> nkey;       // (static score << 32) | rank of the element / of the run's best node
    std::vector<uint32_t> nstat;     // a pseudo-node: NS_ELIG0 iff its run holds an eligible node
    std::vector<uint32_t> cnt;       // nodes the element stands for at its (score, eligible): 1 for a real one
    std::vector<int32_t> min_all;    // min static score over ALL nodes of the element (pruning bound)
    std::vector<uint32_t> lend;      // last element of the element's subtree
    std::vector<uint32_t> lpar;      // innermost REAL element whose subtree holds the element (0 for the root)
    uint32_t pos_lo = 0, pos_hi = 0xFFFFFFFFu;   // only mutations with pos_lo <= position < pos_hi become events
    bool weighted = false;           // has pseudo-nodes: the stream carries ncnt
    bool walk_index = true;          // build the position index / range-query tables of the walk
    uint32_t cp_max_stride = 0;      // != 0: checkpoints at least every cp_max_stride blocks (the whole-tree stream: seed chunks start at checkpoints)
    uint32_t ix_pre_min_nodes = IX_PRE_MIN_NODES;
};

int build_stream_core(const FlatMAT& f, const StreamElems& el, Stream& st, std::string& err) {
    const uint32_t n = (uint32_t)el.g.size();
    st.n = n;
    st.nkey = el.nkey;
    st.nstat = el.nstat;
    if (el.weighted) st.ncnt = el.cnt;
    const std::vector<uint32_t>&lend = el.lend, &lpar = el.lpar;
    // the mutation words of an element that take part in this stream
    auto words_of = [&](uint32_t i, auto&& fn) {
        const uint32_t g = el.g[i];
        if (g == IX_NONE) return;
        for (uint32_t w = f.node_woff[g]; w < f.node_woff[g + 1]; w++) {
            const uint32_t p = f.words[w] & W_POS_MASK;
            if (p >= el.pos_lo && p < el.pos_hi) fn(w);
        }
    };
    auto n_words = [&](uint32_t i) { uint32_t c = 0; words_of(i, [&](uint32_t) { c++; }); return c; };
    // events at local position x: enter words of node x, and exit words of every
    // node a with selected descendants whose subtree ends at x-1.  A node without
    // selected descendants emits no exit; its enter is flagged W_LEAF (it changes
    // no other node's state) -- except the root, which scores itself with its own
    // mutations applied.
    std::vector<uint32_t> evcnt((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t nm = n_words(i);
        evcnt[i] += nm;
        if (lend[i] > i && lend[i] + 1 < n) evcnt[lend[i] + 1] += nm;
    }
    st.blk_node0.clear();
    st.blk_eoff.clear();
    {
        uint32_t d = 0;
        uint64_t eoff = 0;
        while (d < n) {
            st.blk_node0.push_back(d);
            st.blk_eoff.push_back((uint32_t)eoff);
            uint32_t nn = 0;
            uint64_t ne = 0;
            while (d < n && nn < BLK_MAX_NODES && (nn == 0 || ne + evcnt[d] <= BLK_MAX_EVENTS)) {
                ne += evcnt[d];
                nn++;
                d++;
            }
            eoff += ne + (ne & 1);   // even: every lane fetches two words with one 8-byte load
            if (eoff >= 0xFFFFFFF0ull) { err = "too many sweep events"; return WEPP_ELIMIT; }
        }
        st.NB = (uint32_t)st.blk_node0.size();
        st.blk_node0.push_back(n);
        st.blk_eoff.push_back((uint32_t)eoff);
        st.E = eoff;
    }
    {
        st.ev_word.assign(st.E, W_PAD);
        st.ev_meta.assign(st.E, 0);
        st.ev_lb.assign(st.E, 255);
        // min static score over the (selected) subtree of every node, and over the rest of its block
        std::vector<int32_t> submin(n), sufmin(n);
        for (uint32_t i = 0; i < n; i++) submin[i] = el.min_all[i];
        for (uint32_t i = n; i-- > 1;) submin[lpar[i]] = std::min(submin[lpar[i]], submin[i]);
        for (uint32_t b = 0; b < st.NB; b++) {
            int32_t run = SCORE_INF;
            for (uint32_t d = st.blk_node0[b + 1]; d-- > st.blk_node0[b];) {
                run = std::min(run, el.min_all[d]);
                sufmin[d] = run;
            }
        }
        auto clamp8 = [](int32_t v) { return (uint8_t)std::max(0, std::min(255, v)); };
        std::vector<uint32_t> blk_of(n), fill(n, 0);
        {
            uint32_t b = 0, cur = 0;
            for (uint32_t d = 0; d < n; d++) {
                while (st.blk_node0[b + 1] <= d) b++;
                if (d == st.blk_node0[b]) cur = st.blk_eoff[b];
                blk_of[d] = b;
                fill[d] = cur;
                cur += evcnt[d];
            }
        }
        for (uint32_t i = 0; i < n; i++) {           // exits take the first slots of a position
            if (lend[i] > i && lend[i] + 1 < n) {
                const uint32_t x = lend[i] + 1;
                const uint32_t xoff = x - st.blk_node0[blk_of[x]];
                words_of(i, [&](uint32_t w) {
                    const uint32_t e = fill[x]++;
                    st.ev_word[e] = f.words[w] | W_EXIT;
                    st.ev_meta[e] = (uint8_t)xoff;
                    st.ev_lb[e] = clamp8(sufmin[x]);
                });
            }
        }
        for (uint32_t i = 0; i < n; i++) {
            const bool leaf = (lend[i] == i) && i != 0;
            const uint32_t off = i - st.blk_node0[blk_of[i]];
            words_of(i, [&](uint32_t w) {
                const uint32_t e = fill[i]++;
                st.ev_word[e] = f.words[w] | (leaf ? W_LEAF : 0);
                st.ev_meta[e] = (uint8_t)off;
                st.ev_lb[e] = clamp8(submin[i]);
            });
        }
    }
    st.blk_sum.assign(st.NB, BlkSum{SCORE_INF, 0xFFFFFFFFu, 0, SCORE_INF, 0, 0, 0, 0});
    for (uint32_t b = 0; b < st.NB; b++) {
        BlkSum s{SCORE_INF, 0xFFFFFFFFu, 0, SCORE_INF, st.blk_node0[b], st.blk_node0[b + 1] - st.blk_node0[b], 0, 0};
        for (uint32_t d = st.blk_node0[b]; d < st.blk_node0[b + 1]; d++) {
            const int32_t bs = (int32_t)(st.nkey[d] >> 32);
            const uint32_t rk = (uint32_t)(st.nkey[d] & 0xFFFFFFFFll);
            s.min_all = std::min(s.min_all, el.min_all[d]);
            if (!(st.nstat[d] & NS_ELIG0)) continue;
            if (bs < s.base) { s.base = bs; s.rank = rk; s.cnt = el.cnt[d]; }
            else if (bs == s.base) { s.cnt += el.cnt[d]; s.rank = std::min(s.rank, rk); }
        }
        st.blk_sum[b] = s;
    }
    // ---- position index and range-query tables of the walk ------------------------------
    if (el.walk_index) {
        const uint32_t np = f.max_pos + 1;
        std::vector<uint32_t> off((size_t)np + 1, 0);
        for (uint32_t i = 0; i < n; i++) words_of(i, [&](uint32_t w) { off[(f.words[w] & W_POS_MASK) + 1]++; });
        for (uint32_t p = 0; p < np; p++) off[p + 1] += off[p] + 1;   // + the sentinel of list p
        const size_t total = off[np];
        st.ix_ent.assign(total, IxEnt{IX_NONE, 0, W_PAD, IX_NONE, IX_NONE, SCORE_INF, 0xFFFFFFFFu, 0});
        std::vector<uint32_t> fill(off.begin(), off.end() - 1);
        for (uint32_t i = 0; i < n; i++)            // ascending node index: every list ends up in stream order
            words_of(i, [&](uint32_t w) {
                IxEnt& e = st.ix_ent[fill[f.words[w] & W_POS_MASK]++];
                e.node = i;
                e.end = lend[i] + 1;
                e.word = f.words[w];
                e.base = (int32_t)(st.nkey[i] >> 32);
                e.rank = (uint32_t)(st.nkey[i] & 0xFFFFFFFFll);
                e.nstat = st.nstat[i];
            });
        for (size_t e = 0; e + 1 < total; e++) st.ix_ent[e].next_node = st.ix_ent[e + 1].node;   // (a sentinel's is never read)
        st.ix_head.resize((size_t)np + 1);
        for (uint32_t p = 0; p <= np; p++) st.ix_head[p] = IxHead{off[p], p < np ? st.ix_ent[off[p]].node : IX_NONE};
        // innermost enclosing entry of every entry of a list: the subtrees are nested, so a stack of the
        // entries still open does it
        st.ix_nest.assign(np, 0);
        {
            std::vector<uint32_t> open;
            for (uint32_t p = 0; p < np; p++) {
                open.clear();
                size_t deepest = 0;
                for (uint32_t e = off[p]; e + 1 < off[p + 1]; e++) {
                    while (!open.empty() && st.ix_ent[open.back()].end <= st.ix_ent[e].node) open.pop_back();
                    if (!open.empty()) st.ix_ent[e].up = open.back();
                    open.push_back(e);
                    deepest = std::max(deepest, open.size());
                }
                st.ix_nest[p] = (uint8_t)std::min<size_t>(deepest, 255);
            }
        }
        st.nrec.resize(n);
        for (uint32_t i = 0; i < n; i++)
            st.nrec[i] = NodeRec{(int32_t)(st.nkey[i] >> 32), (uint32_t)(st.nkey[i] & 0xFFFFFFFFll), st.nstat[i], 0};
        const SegNode none{SCORE_INF, 0xFFFFFFFFu, 0, 0};
        auto join = [](const SegNode& a, const SegNode& b) {
            if (b.base < a.base) return b;
            if (b.base > a.base) return a;
            return SegNode{a.base, std::min(a.rank, b.rank), a.cnt + b.cnt, b.rank < a.rank ? b.hu : a.hu};
        };
        st.sp_levels = 1;
        while ((1u << (st.sp_levels - 1)) < n) st.sp_levels++;       // levels 0 .. ceil(log2 n): the last one spans the stream from any start
        st.sp.assign((size_t)st.sp_levels * n, SP_NONE);
        st.rq_pre.assign(n, none);
        st.rq_suf.assign(n, none);
        st.rq_blocks = (n + RQ_BLK - 1) / RQ_BLK;
        st.rq_levels = 1;
        while ((1u << st.rq_levels) < st.rq_blocks) st.rq_levels++;   // rows 0 .. highest bit two block indices can differ in
        st.rq_dst.assign((size_t)st.rq_levels * st.rq_blocks, none);
        auto leaf = [&](uint32_t i) {
            const uint32_t ns = st.nstat[i];
            if (!(ns & NS_ELIG0)) return none;
            // has_unique of a node no listed position touches (usher_mapper.cpp:184,199,262): some mutation the
            // sample does not share -- never for the root, always behind a masked mutation
            const uint32_t hu = (ns & NS_ROOT) ? 0u : (ns & NS_MASKED) ? 1u : (((ns >> 14) & NS_CNT_MASK) < (ns & NS_CNT_MASK) ? 1u : 0u);
            return SegNode{(int32_t)(st.nkey[i] >> 32), (uint32_t)(st.nkey[i] & 0xFFFFFFFFll), 1, hu};
        };
        for (uint32_t i = 0; i < n; i++)
            if (st.nstat[i] & NS_ELIG0) st.sp[i] = (uint8_t)std::max(0, std::min<int32_t>((int32_t)(st.nkey[i] >> 32), SP_CLAMP));
        st.whole = none;
        for (uint32_t b = 0; b < st.rq_blocks; b++) {
            const uint32_t lo = b * RQ_BLK, hi = std::min(n, lo + RQ_BLK);
            SegNode run = none;
            for (uint32_t i = lo; i < hi; i++) { run = join(run, leaf(i)); st.rq_pre[i] = run; }
            st.rq_dst[b] = run;
            st.whole = join(st.whole, run);
            run = none;
            for (uint32_t i = hi; i-- > lo;) { run = join(run, leaf(i)); st.rq_suf[i] = run; }
        }
        for (uint32_t l = 1; l < st.rq_levels; l++) {
            // row l serves the pairs of blocks whose indices differ in bit l at the highest: they sit in the two
            // halves of one 2^(l+1)-aligned group, and row l holds every block's aggregate towards the middle of
            // its group (row 0, the blocks themselves, is that for adjacent pairs)
            SegNode* row = st.rq_dst.data() + (size_t)l * st.rq_blocks;
            const uint32_t half = 1u << l, size = half << 1;
            for (uint32_t c = 0; c < st.rq_blocks; c += size) {
                const uint32_t mid = c + half;
                if (mid >= st.rq_blocks) break;
                SegNode run = none;
                for (uint32_t i = mid; i-- > c;) { run = join(run, st.rq_dst[i]); row[i] = run; }
                run = none;
                for (uint32_t i = mid; i < std::min(c + size, st.rq_blocks); i++) { run = join(run, st.rq_dst[i]); row[i] = run; }
            }
        }
        for (uint32_t l = 1; l < st.sp_levels; l++) {
            const uint32_t half = 1u << (l - 1);
            const uint8_t* lo = st.sp.data() + (size_t)(l - 1) * n;
            uint8_t* hi = st.sp.data() + (size_t)l * n;
            for (uint32_t i = 0; i < n; i++) hi[i] = i + half < n ? std::min(lo[i], lo[i + half]) : lo[i];
        }
        // pre-test byte of the nodes between consecutive entries of a list, in the entry that ends the range
        st.ix_pre.assign(1, 0);
        if (f.N <= (1u << IX_RANK_BITS) && n >= el.ix_pre_min_nodes) {
            st.ix_pre[0] = 1;
            auto range_min = [&](uint32_t a, uint32_t b) -> uint32_t {     // [a, b), a < b: exact minimum (two overlapping spans)
                uint32_t l = 0;
                while ((2u << l) <= b - a) l++;
                const uint8_t* row = st.sp.data() + (size_t)l * n;
                return std::min<uint32_t>(row[a], row[b - (1u << l)]);
            };
            for (uint32_t p = 0; p < np; p++) {
                uint32_t prev = 0;
                for (uint32_t e = off[p]; e + 1 < off[p + 1]; e++) {
                    IxEnt& x = st.ix_ent[e];
                    const uint32_t v = x.node > prev ? range_min(prev, x.node) : (uint32_t)SP_NONE;
                    x.rank = (x.rank & IX_RANK_MASK) | (v << IX_RANK_BITS);
                    prev = x.node + 1;
                }
            }
        }
    }
    // checkpoints
    st.cp_stride = std::max<uint32_t>(1, (st.NB + 1023) / 1024);   // <= 1024 places where a sweep may start ...
    if (el.cp_max_stride) st.cp_stride = std::min(st.cp_stride, el.cp_max_stride);   // ... more on the whole-tree stream: one per seed chunk
    {
        const uint32_t ncp = (st.NB + st.cp_stride - 1) / st.cp_stride;
        st.cp_off.assign(ncp + 1, 0);
        st.cp_word.clear();
        std::vector<uint32_t> path;
        for (uint32_t i = 0; i < ncp; i++) {
            st.cp_off[i] = (uint32_t)st.cp_word.size();
            // Running state of a sequential sweep when it reaches block b: the
            // enter words of every node still open after local node x-1, i.e. x-1
            // itself (if it has selected descendants) and all its ancestors.
            // Subtrees ending exactly at x-1 are closed by exit events INSIDE block b.
            const uint32_t x = st.blk_node0[i * st.cp_stride];
            path.clear();
            if (x > 0) {
                uint32_t a = x - 1;
                if (lend[a] > a) path.push_back(a);
                while (a != 0) { a = lpar[a]; path.push_back(a); }
            }
            for (size_t k = path.size(); k-- > 0;) words_of(path[k], [&](uint32_t w) { st.cp_word.push_back(f.words[w]); });
        }
        st.cp_off[ncp] = (uint32_t)st.cp_word.size();
    }
    return WEPP_OK;
}

// a crown (or the whole tree): the ancestor-closed node subset `sel` (sorted global DFS indices, sel[0] == 0)
int build_stream(const FlatMAT& f, const std::vector<uint32_t>& sel, Stream& st, std::string& err, const FlattenOptions& opt,
                 uint32_t pos_lo = 0, uint32_t pos_hi = 0xFFFFFFFFu, bool walk_index = true, uint32_t cp_max_stride = 0) {
    const uint32_t n = (uint32_t)sel.size();
    const uint32_t N = f.N;
    // number of selected nodes with global index <= g
    std::vector<uint32_t> upto(N);
    {
        uint32_t c = 0, k = 0;
        for (uint32_t g = 0; g < N; g++) {
            if (k < n && sel[k] == g) { c++; k++; }
            upto[g] = c;
        }
    }
    StreamElems el;
    el.g = sel;
    el.pos_lo = pos_lo;          // (a window crown indexes the mutations of its window only)
    el.pos_hi = pos_hi;
    el.walk_index = walk_index;
    el.cp_max_stride = cp_max_stride;
    el.ix_pre_min_nodes = opt.ix_pre_min_nodes;
    el.nkey.resize(n); el.nstat.resize(n); el.cnt.assign(n, 1); el.min_all.resize(n); el.lend.resize(n); el.lpar.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t g = sel[i];
        el.nkey[i] = f.nkey[g];
        el.nstat[i] = f.nstat[g];
        el.min_all[i] = (int32_t)(f.nkey[g] >> 32);
        el.lend[i] = upto[f.dfs_end[g]] - 1;          // last selected node of the subtree
        el.lpar[i] = i ? upto[f.parent_dfs[g]] - 1 : 0;
    }
    return build_stream_core(f, el, st, err);
}

// the whole tree as the reads of one genome window [lo, hi) see it: only the nodes that mutate a position of
// the window are elements of their own; every maximal run of other nodes between two places where such a
// read's c_S can change becomes one pseudo-node.  20x fewer events than the whole-tree stream for a 2 K window.
int build_window_stream(const FlatMAT& f, uint32_t lo, uint32_t hi, Stream& st, std::string& err) {
    const uint32_t N = f.N;
    // boundaries: a touched node starts an element of its own; behind the subtree of a touched node c_S changes back
    std::vector<uint8_t> touched(N, 0), cut((size_t)N + 1, 0);
    touched[0] = 1;                                          // the root is element 0 of every stream
    for (uint32_t g = 0; g < N; g++) {
        if (!touched[g])
            for (uint32_t w = f.node_woff[g]; w < f.node_woff[g + 1]; w++) {
                const uint32_t p = f.words[w] & W_POS_MASK;
                if (p >= lo && p < hi) { touched[g] = 1; break; }
            }
        if (touched[g]) { cut[g] = 1; cut[g + 1] = 1; cut[f.dfs_end[g] + 1] = 1; }
    }
    StreamElems el;
    el.pos_lo = lo;
    el.pos_hi = hi;
    el.weighted = true;
    el.walk_index = false;
    std::vector<uint32_t> first_of;            // global index an element starts at (for lend)
    std::vector<uint32_t> open;                // real elements whose subtree holds the current position (local indices)
    std::vector<uint32_t> open_end;            // their global subtree ends
    uint32_t g = 0;
    while (g < N) {
        while (!open.empty() && open_end.back() < g) { open.pop_back(); open_end.pop_back(); }
        const uint32_t local = (uint32_t)el.g.size();
        const uint32_t par = open.empty() ? 0u : open.back();
        if (touched[g]) {
            el.g.push_back(g);
            el.nkey.push_back(f.nkey[g]);
            el.nstat.push_back(f.nstat[g]);
            el.cnt.push_back(1);
            el.min_all.push_back((int32_t)(f.nkey[g] >> 32));
            el.lpar.push_back(par);
            first_of.push_back(g);
            if (f.dfs_end[g] > g) { open.push_back(local); open_end.push_back(f.dfs_end[g]); }
            g++;
            continue;
        }
        // a run of untouched nodes up to the next cut
        uint32_t y = g + 1;
        while (y < N && !cut[y]) y++;
        int32_t best = SCORE_INF, mall = SCORE_INF;
        uint32_t rank = 0xFFFFFFFFu, cnt = 0;
        for (uint32_t x = g; x < y; x++) {
            const int32_t bs = (int32_t)(f.nkey[x] >> 32);
            mall = std::min(mall, bs);
            if (!(f.nstat[x] & NS_ELIG0)) continue;
            const uint32_t rk = (uint32_t)(f.nkey[x] & 0xFFFFFFFFll);
            if (bs < best) { best = bs; rank = rk; cnt = 1; }
            else if (bs == best) { cnt++; rank = std::min(rank, rk); }
        }
        el.g.push_back(IX_NONE);
        el.nkey.push_back(((int64_t)best << 32) | (int64_t)rank);
        el.nstat.push_back(cnt ? NS_ELIG0 : 0u);
        el.cnt.push_back(cnt);
        el.min_all.push_back(mall);
        el.lpar.push_back(par);
        first_of.push_back(g);
        g = y;
    }
    // last element of every real element's subtree: the element before the first one starting behind its end
    const uint32_t n = (uint32_t)el.g.size();
    el.lend.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        if (el.g[i] == IX_NONE || f.dfs_end[el.g[i]] == el.g[i]) { el.lend[i] = i; continue; }
        const uint32_t endg = f.dfs_end[el.g[i]];
        el.lend[i] = (uint32_t)(std::upper_bound(first_of.begin() + i, first_of.end(), endg) - first_of.begin()) - 1;
    }
    st.tau = 0x7FFFFFFF;
    return build_stream_core(f, el, st, err);
}

}  // namespace

static std::atomic<uint64_t> g_flatten_count{0};
uint64_t flatten_count() { return g_flatten_count.load(std::memory_order_relaxed); }

int flatten_tree(const wepp_tree_desc& t, FlatMAT& f, std::string& err, bool topology_only) {
    const FlattenOptions opt = FlattenOptions::from_env();
    const uint32_t N = t.n_nodes;
    if (N == 0 || !t.parent || !t.mut_off) { err = "empty tree or null arrays"; return WEPP_EINVAL; }
    if (N >= 0xFFFFFFF0u) { err = "too many nodes"; return WEPP_ELIMIT; }
    const uint64_t Mall = t.mut_off[N];
    if (Mall && (!t.mut_pos || !t.mut_ref || !t.mut_mut)) { err = "null mutation arrays"; return WEPP_EINVAL; }
    if (Mall >= 0xFFFFFFF0ull) { err = "too many mutations"; return WEPP_ELIMIT; }

    // ---- children CSR (ascending id order), root ---------------------------
    std::vector<uint32_t> coff(N + 1, 0), child(N ? N - 1 : 0);
    int64_t root = -1;
    for (uint32_t i = 0; i < N; i++) {
        int32_t p = t.parent[i];
        if (p < 0) {
            if (root >= 0) { err = "more than one root"; return WEPP_EINVAL; }
            root = i;
        } else {
            if ((uint32_t)p >= N || (uint32_t)p == i) { err = "parent id out of range"; return WEPP_EINVAL; }
            coff[p + 1]++;
        }
        if (t.mut_off[i + 1] < t.mut_off[i]) { err = "mut_off not monotone"; return WEPP_EINVAL; }
    }
    if (root < 0) { err = "no root"; return WEPP_EINVAL; }
    for (uint32_t i = 0; i < N; i++) coff[i + 1] += coff[i];
    {
        std::vector<uint32_t> fill(coff.begin(), coff.end() - 1);
        for (uint32_t i = 0; i < N; i++)
            if (t.parent[i] >= 0) child[fill[t.parent[i]]++] = i;
    }

    // ---- BFS order (mutation_annotated_tree.cpp:1115-1141) ------------------
    std::vector<uint32_t> bfs_of_id(N, 0xFFFFFFFFu);
    f.bfs2id.assign(N, 0);
    {
        uint32_t head = 0, tail = 0;
        f.bfs2id[tail++] = (uint32_t)root;
        while (head < tail) {
            uint32_t c = f.bfs2id[head];
            bfs_of_id[c] = head++;
            for (uint32_t k = coff[c]; k < coff[c + 1]; k++) {
                if (tail >= N) { err = "tree has a cycle"; return WEPP_EINVAL; }
                f.bfs2id[tail++] = child[k];
            }
        }
        if (tail != N) { err = "tree is disconnected or has a cycle"; return WEPP_EINVAL; }
    }

    // ---- DFS pre-order (mutation_annotated_tree.cpp:1143-1163) --------------
    f.dfs2id.assign(N, 0);
    std::vector<uint32_t> dfs_of_id(N), depth(N, 0);
    {
        std::vector<uint32_t> stack;
        stack.reserve(1024);
        stack.push_back((uint32_t)root);
        uint32_t cnt = 0;
        while (!stack.empty()) {
            uint32_t c = stack.back();
            stack.pop_back();
            dfs_of_id[c] = cnt;
            f.dfs2id[cnt++] = c;
            for (uint32_t k = coff[c + 1]; k > coff[c]; k--) stack.push_back(child[k - 1]);
        }
    }
    f.N = N;
    f.parent_dfs.assign(N, 0);
    f.dfs_end.assign(N, 0);
    f.num_leaves.assign(N, 0);
    f.dfs2bfs.assign(N, 0);
    f.max_depth = 0;
    for (uint32_t d = 0; d < N; d++) {
        uint32_t id = f.dfs2id[d];
        f.dfs2bfs[d] = bfs_of_id[id];
        if (t.parent[id] >= 0) {
            uint32_t pd = dfs_of_id[t.parent[id]];
            f.parent_dfs[d] = pd;
            depth[d] = depth[pd] + 1;
            f.max_depth = std::max(f.max_depth, depth[d]);
        }
    }
    // subtree end + leaf counts: reverse DFS order visits children before parents
    f.n_leaves = 0;
    for (uint32_t d = 0; d < N; d++) f.dfs_end[d] = d;
    for (uint32_t d = N; d-- > 0;) {
        uint32_t id = f.dfs2id[d];
        if (coff[id] == coff[id + 1]) { f.num_leaves[d] = 1; f.n_leaves++; }
        if (d != 0) {
            uint32_t pd = f.parent_dfs[d];
            f.num_leaves[pd] += f.num_leaves[d];
            f.dfs_end[pd] = std::max(f.dfs_end[pd], f.dfs_end[d]);
        }
    }

    // ---- mutation words in DFS order, validation ----------------------------
    f.node_woff.assign(N + 1, 0);
    f.nstat.assign(N, 0);
    f.n_masked = 0;
    uint32_t max_pos = 0;
    std::vector<uint32_t> root_masked_cost(1, 0);
    for (uint32_t d = 0; d < N; d++) {
        uint32_t id = f.dfs2id[d];
        uint32_t nm = 0;
        int32_t prev = INT32_MIN;
        bool masked = false;
        for (uint32_t k = t.mut_off[id]; k < t.mut_off[id + 1]; k++) {
            int32_t p = t.mut_pos[k];
            if (p < prev) { err = "node mutations not sorted by position (node id " + std::to_string(id) + ")"; return WEPP_EINVAL; }
            if (p >= 0 && p == prev) { err = "duplicate mutation position in node id " + std::to_string(id); return WEPP_EINVAL; }
            prev = p;
            if (p < 0) {
                masked = true;
                f.n_masked++;
                // root: masked mutations enter the ancestral set (usher_mapper.cpp:267-270)
                // and cost 1 in the back-mutation loop iff ref_nuc != mut_nuc (:429-437)
                if (d == 0 && (t.mut_ref[k] & 15) != (t.mut_mut[k] & 15)) root_masked_cost[0]++;
                continue;
            }
            if ((uint32_t)p > WEPP_MAX_POSITION) { err = "mutation position exceeds 2^20-2"; return WEPP_ELIMIT; }
            if ((t.mut_mut[k] & 15) == 0 || (t.mut_ref[k] & 15) == 0) { err = "zero nucleotide mask on a non-masked mutation"; return WEPP_EINVAL; }
            if ((t.mut_ref[k] & 15) & ((t.mut_ref[k] & 15) - 1)) { err = "ref_nuc must be a single nucleotide (position " + std::to_string(p) + ")"; return WEPP_EINVAL; }
            max_pos = std::max(max_pos, (uint32_t)p);
            nm++;
        }
        if (nm > NS_CNT_MASK) { err = "more than 16383 mutations on one node"; return WEPP_ELIMIT; }
        f.node_woff[d + 1] = f.node_woff[d] + nm;
        if (masked && d != 0) f.nstat[d] |= NS_MASKED;
        if (coff[id] == coff[id + 1]) f.nstat[d] |= NS_LEAF;
    }
    f.nstat[0] |= NS_ROOT;
    f.M = f.node_woff[N];
    f.max_pos = max_pos;
    f.words.assign(f.M, 0);

    // ---- true parent alleles, D0, per-node constants ------------------------
    // state[p] = allele mask of the most recent mutation at p on the current
    // root path (0 = none); refm[p] = the (single) ref mask seen at p.
    std::vector<uint8_t> state((size_t)max_pos + 1, 0), refm((size_t)max_pos + 1, 0);
    std::vector<uint8_t> nest((size_t)max_pos + 1, 0);     // mutations at p on the current root path (saturating)
    f.maxnest.assign((size_t)max_pos + 1, 0);
    std::vector<int32_t> D0(N, 0);
    std::vector<int32_t> base(N, 0);
    // window crowns (flatmat.hpp): inw[w] = positions of window w at which the current path's genotype differs from
    // the reference; cand[w] = the nodes a read of window w can be placed on at all (below), in DFS order, with their out_w
    const uint32_t n_win = (!topology_only && max_pos + 1 > WIN_STRIDE) ? std::min<uint32_t>(MAX_WINDOWS, (max_pos + WIN_STRIDE) / WIN_STRIDE) : 0u;
    // inw_p[w] = upper bound of the window positions at which the genotype a node is SCORED against can differ from the
    // reference, whatever the read: the parent genotype's, plus the node's own mutations away from the reference
    std::vector<int32_t> inw(n_win, 0), inw_e(n_win, 0), inw_par(n_win, 0), inw_p(n_win, 0);
    std::vector<std::vector<std::pair<uint32_t, int32_t>>> cand(n_win);
    // the windows [w * WIN_STRIDE, w * WIN_STRIDE + WIN_SIZE) that hold position p
    auto windows_of = [&](uint32_t p, auto&& fn) {
        const uint32_t hi = std::min(n_win, p / WIN_STRIDE + 1);
        const uint32_t lo = p + 1 > WIN_SIZE ? (p + 1 - WIN_SIZE + WIN_STRIDE - 1) / WIN_STRIDE : 0u;
        for (uint32_t w = lo; w < hi; w++) fn(w);
    };
    {
        std::vector<uint32_t> open;           // stack of open nodes (DFS idx)
        open.reserve(f.max_depth + 2);
        std::vector<uint8_t> undo(f.M, 0);     // previous state per word
        for (uint32_t d = 0; d < N; d++) {
            while (!open.empty() && f.dfs_end[open.back()] < d) {
                uint32_t x = open.back();
                open.pop_back();
                for (uint32_t w = f.node_woff[x + 1]; w > f.node_woff[x]; w--) {
                    const uint32_t p = f.words[w - 1] & W_POS_MASK;
                    if (n_win) {
                        const int32_t dl = (int32_t)cost0(state[p], refm[p]) - (int32_t)cost0(undo[w - 1], refm[p]);
                        if (dl) windows_of(p, [&](uint32_t wi) { inw[wi] -= dl; });
                    }
                    state[p] = undo[w - 1];
                    if (nest[p] != 255) nest[p]--;
                }
            }
            uint32_t id = f.dfs2id[d];
            int32_t dpar = (d == 0) ? 0 : D0[f.parent_dfs[d]];
            int32_t dcur = dpar;
            uint32_t nback_cost = 0, ncommon0 = 0;
            uint32_t w = f.node_woff[d];
            if (n_win) { inw_par = inw; inw_e = inw; inw_p = inw; }     // the parent genotype's counts; own back-mutations are taken off inw_e below
            for (uint32_t k = t.mut_off[id]; k < t.mut_off[id + 1]; k++) {
                int32_t p = t.mut_pos[k];
                if (p < 0) continue;
                uint32_t ref = t.mut_ref[k] & 15, mut = t.mut_mut[k] & 15;
                if (refm[p] == 0) refm[p] = (uint8_t)ref;
                else if (refm[p] != ref) { err = "inconsistent ref_nuc at position " + std::to_string(p); return WEPP_EINVAL; }
                uint32_t par = state[p];
                f.words[w] = w_pack((uint32_t)p, (uint32_t)__builtin_ctz(ref), par, mut);
                undo[w] = (uint8_t)par;
                state[p] = (uint8_t)mut;
                if (nest[p] != 255) nest[p]++;          // (a position that ever saturates stays at 255)
                f.maxnest[p] = std::max(f.maxnest[p], nest[p]);
                dcur += (int32_t)cost0(mut, ref) - (int32_t)cost0(par, ref);
                if (mut == ref) { ncommon0++; nback_cost += cost0(par, ref); }
                if (n_win) {
                    const int32_t dl = (int32_t)cost0(mut, ref) - (int32_t)cost0(par, ref);
                    const bool back = mut == ref && cost0(par, ref);
                    if (dl || back) windows_of((uint32_t)p, [&](uint32_t wi) { inw[wi] += dl; if (back) inw_e[wi]--; if (dl > 0) inw_p[wi]++; });
                }
                w++;
            }
            D0[d] = dcur;
            uint32_t nmut = f.node_woff[d + 1] - f.node_woff[d];
            uint32_t st = f.nstat[d];
            bool leaf = st & NS_LEAF, masked = st & NS_MASKED;
            bool elig0;
            if (d == 0) {
                // root score = D(root) (usher_mapper.cpp:266-271); its masked mutations are
                // seen by the root only (descendants skip masked ancestors, :281)
                base[0] = dcur + (int32_t)root_masked_cost[0];
                elig0 = true;             // root always competes (usher_mapper.cpp:455)
                ncommon0 = 0;             // root: node_num_mut = num_common_mut = 0
            } else {
                base[d] = masked ? dpar : dpar - (int32_t)nback_cost;   // masked: loop breaks before any common
                if (masked) elig0 = false;                    // has_unique, num_common 0
                else if (leaf) elig0 = ncommon0 > 0;
                else elig0 = (ncommon0 > 0) || (ncommon0 == nmut);
            }
            f.nstat[d] = st | (nmut & NS_CNT_MASK) | ((ncommon0 & NS_CNT_MASK) << 14) | (elig0 ? NS_ELIG0 : 0);
            if (n_win && d) {
                // out_w(d) = base(d) minus the window's share of it (a masked node is scored on its parent's genotype
                // as it stands: none of its back-mutations counts, inside or outside the window)
                // A read of the window scores d at least s0 + out_w(d) - in_w(d) -- s0 = what its entries cost against the
                // reference, in_w(d) <= inw_p the window positions at which the scored genotype is not the reference
                // (each can save the read at most one) -- and the root, which always competes, at most s0 + base(root):
                // a node with out_w - inw_p > base(root) can neither win nor tie, whatever the read lists (flatmat.hpp)
                for (uint32_t wi = 0; wi < n_win; wi++) {
                    const int32_t out = base[d] - (masked ? inw_par[wi] : inw_e[wi]);
                    if (out - (masked ? inw_par[wi] : inw_p[wi]) <= base[0]) cand[wi].emplace_back(d, out);
                }
            }
            open.push_back(d);
        }
    }

    if (topology_only) return WEPP_OK;     // orders, parents, leaf flags: all the Fitch-Sankoff pass needs
    g_flatten_count.fetch_add(1, std::memory_order_relaxed);

    // ---- tie-break rank: larger num_leaves first, then larger BFS index -----
    f.rank2dfs.resize(N);
    std::iota(f.rank2dfs.begin(), f.rank2dfs.end(), 0u);
    std::sort(f.rank2dfs.begin(), f.rank2dfs.end(), [&](uint32_t a, uint32_t b) {
        if (f.num_leaves[a] != f.num_leaves[b]) return f.num_leaves[a] > f.num_leaves[b];
        return f.dfs2bfs[a] > f.dfs2bfs[b];
    });
    f.nkey.assign(N, 0);
    for (uint32_t r = 0; r < N; r++) {
        uint32_t d = f.rank2dfs[r];
        f.nkey[d] = ((int64_t)base[d] << 32) | (int64_t)r;
    }

    f.rank2bfs.resize(N);
    for (uint32_t r = 0; r < N; r++) f.rank2bfs[r] = f.dfs2bfs[f.rank2dfs[r]];
    f.root_base = base[0];

    // ---- sweep streams: crowns of increasing tau, then the whole tree ------------
    // A node with static score base(n) can never reach a read's best score if
    // base(n) > theta(read), so a read only needs the
    // crown {base <= theta} closed under ancestors.
    {
        // theta(read) = score(root) + |S| >= base(root): the ladder is relative to the root's static score
        static const int32_t dtaus[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 16, 19, 23, 27};
        std::vector<uint8_t> keep(N);
        std::vector<uint32_t> sel;
        size_t prev = 0;
        for (int32_t dtau : dtaus) {
            const int32_t tau = f.root_base + dtau;
            if (f.streams.size() + 2 >= MAX_STREAMS) break;     // (the last slot belongs to the window crowns)
            for (uint32_t d = 0; d < N; d++) keep[d] = base[d] <= tau;
            keep[0] = 1;
            for (uint32_t d = N; d-- > 1;)
                if (keep[d]) keep[f.parent_dfs[d]] = 1;
            sel.clear();
            for (uint32_t d = 0; d < N; d++)
                if (keep[d]) sel.push_back(d);
            if (sel.size() * 2 > N) break;                       // not worth a separate stream
            if (prev && sel.size() < prev + prev / 4) continue;  // too close to the previous crown
            f.streams.emplace_back();
            f.streams.back().tau = tau;
            int rc = build_stream(f, sel, f.streams.back(), err, opt);
            if (rc != WEPP_OK) return rc;
            prev = sel.size();
        }
        sel.resize(N);
        std::iota(sel.begin(), sel.end(), 0u);
        f.streams.emplace_back();
        int rc = build_stream(f, sel, f.streams.back(), err, opt, 0, 0xFFFFFFFFu, true, opt.seed_chunk_blocks);
        if (rc != WEPP_OK) return rc;
    }

    // ---- seed signatures (flatmat.hpp): per (position, chunk of the whole-tree stream) the alleles a genotype scored in
    // the chunk can hold -- the mutations of the chunk's nodes and of the ancestors of its first node ---------------
    {
        const Stream& full = f.streams.back();
        uint32_t stride = full.cp_stride * ((opt.seed_chunk_blocks + full.cp_stride - 1) / full.cp_stride);
        if ((full.NB + stride - 1) / stride > SEED_MAX_CHUNKS) {
            const uint32_t need = (full.NB + SEED_MAX_CHUNKS - 1) / SEED_MAX_CHUNKS;
            stride = full.cp_stride * ((need + full.cp_stride - 1) / full.cp_stride);
        }
        const uint32_t nch = (full.NB + stride - 1) / stride;
        const uint32_t row_words = (((nch + 7) / 8) + 3) & ~3u;          // 16-byte rows
        const uint64_t bytes = (uint64_t)(f.max_pos + 2) * row_words * 4;
        if (f.max_pos <= SEED_MAX_POS && bytes <= SEED_MAX_SIG_BYTES) {
            f.seed_stride = stride;
            f.seed_chunks = nch;
            f.seed_row_words = row_words;
            f.seed_sig.assign((size_t)(f.max_pos + 2) * row_words, 0u);
            auto mark = [&](uint32_t word, uint32_t c) {
                f.seed_sig[(size_t)(word & W_POS_MASK) * row_words + (c >> 3)] |= w_mut(word) << ((c & 7u) * 4u);
            };
            for (uint32_t c = 0; c < nch; c++) {
                // (the whole-tree stream's local node indices are the global DFS indices)
                const uint32_t a = full.blk_node0[c * stride], b = full.blk_node0[std::min(full.NB, (c + 1) * stride)];
                for (uint32_t w = f.node_woff[a]; w < f.node_woff[b]; w++) mark(f.words[w], c);
                for (uint32_t v = a; v != 0;) {
                    v = f.parent_dfs[v];
                    for (uint32_t w = f.node_woff[v]; w < f.node_woff[v + 1]; w++) mark(f.words[w], c);
                }
            }
        }
    }

    // ---- window crowns (flatmat.hpp): per genome window, the candidates with out_w <= tau and their ancestors; and the
    // window STREAM the tiles of long reads sweep: all the window's candidates (any root score), or the whole tree as
    // the window sees it when they are too many to be worth a crown --------
    if (n_win) {
        f.wcrowns.assign(n_win, {});
        f.wstreams.resize(n_win);
        std::vector<std::string> errs(n_win);
        std::vector<int> rcs(n_win, WEPP_OK);
        const bool no_balanced = opt.win_whole_tree;   // (A/B and test aid)
        auto build_window = [&](uint32_t wi, std::vector<uint8_t>& mark) {
            // the crowns of a window nest: the marks stay from one tau to the next
            std::vector<uint32_t> sel(1, 0u);
            mark[0] = 1;
            size_t prev = 0, at = 0;
            const auto& cw = cand[wi];
            std::vector<std::pair<int32_t, uint32_t>> by_out(cw.size());
            for (size_t i = 0; i < cw.size(); i++) by_out[i] = {cw[i].second, cw[i].first};
            std::sort(by_out.begin(), by_out.end());
            const uint32_t lo = wi * WIN_STRIDE, hi = wi * WIN_STRIDE + WIN_SIZE;
            bool all = false;                                   // the last crown built holds every candidate
            for (uint32_t dt = 0; dt <= WC_MAX_DTAU + 1 && f.wcrowns[wi].size() < WC_MAX; dt++) {
                // (the last level takes all candidates: tau = "any root score")
                const bool last = dt == WC_MAX_DTAU + 1;
                const int32_t tau = last ? 0x7FFFFFFF : f.root_base + (int32_t)dt;
                for (; at < by_out.size() && by_out[at].first <= tau; at++)
                    for (uint32_t d = by_out[at].second; !mark[d]; d = f.parent_dfs[d]) { mark[d] = 1; sel.push_back(d); }
                if (sel.size() > WC_MAX_NODES || sel.size() * 4 > N) break;         // (the tree-wide crowns take over)
                if (at == by_out.size() && !last) continue;                          // (nothing more to come: the last level is this one)
                if (prev && sel.size() < prev + prev / 4 && !last) continue;        // too close to the previous one
                if (last && sel.size() == prev) {                                    // the previous crown already holds them all
                    f.wcrowns[wi].back().tau = tau;
                    all = true;
                    break;
                }
                std::sort(sel.begin(), sel.end());
                f.wcrowns[wi].emplace_back();
                Stream& st = f.wcrowns[wi].back();
                st.tau = tau;
                rcs[wi] = build_stream(f, sel, st, errs[wi], opt, lo, hi);
                if (rcs[wi] != WEPP_OK) break;
                prev = sel.size();
                all = last;
            }
            if (rcs[wi] == WEPP_OK) {
                if (all && !no_balanced) {
                    rcs[wi] = build_stream(f, sel, f.wstreams[wi], errs[wi], opt, lo, hi, /*walk_index=*/false);
                    f.wstreams[wi].tau = 0x7FFFFFFF;
                } else rcs[wi] = build_window_stream(f, lo, hi, f.wstreams[wi], errs[wi]);
            }
            for (uint32_t d : sel) mark[d] = 0;
        };
        const uint32_t nthr = std::max(1u, std::min<uint32_t>({n_win, 8u, std::thread::hardware_concurrency()}));
        std::atomic<uint32_t> next{0};
        auto worker = [&]() {
            uint32_t wi = 0;
            try {
                std::vector<uint8_t> mark(N, 0);
                for (; (wi = next.fetch_add(1)) < n_win;) build_window(wi, mark);
            } catch (const std::bad_alloc&) {          // (an exception must not leave a worker thread)
                if (wi < n_win) { rcs[wi] = WEPP_ENOMEM; errs[wi] = "out of host memory while building the window crowns"; }
                else { rcs[0] = WEPP_ENOMEM; errs[0] = "out of host memory while building the window crowns"; }
            }
        };
        if (nthr == 1) worker();
        else {
            std::vector<std::thread> th;
            for (uint32_t i = 0; i < nthr; i++) th.emplace_back(worker);
            for (auto& x : th) x.join();
        }
        for (uint32_t wi = 0; wi < n_win; wi++)
            if (rcs[wi] != WEPP_OK) { err = errs[wi]; return rcs[wi]; }
    }

    // ---- EPP event stream (see flatmat.hpp) -----------------------------------------
    {
        f.epp_word.reserve(2 * f.M);
        f.epp_node.reserve(2 * f.M);
        std::vector<uint32_t> open;                          // nodes with mutations on the current path
        auto close = [&](uint32_t a) {
            for (uint32_t k = f.node_woff[a]; k < f.node_woff[a + 1]; k++) {
                f.epp_word.push_back(f.words[k] | W_EXIT);
                f.epp_node.push_back(f.dfs_end[a] + 1);
            }
        };
        for (uint32_t d = 0; d < N; d++) {
            while (!open.empty() && f.dfs_end[open.back()] < d) { close(open.back()); open.pop_back(); }
            if (f.node_woff[d + 1] == f.node_woff[d]) continue;
            for (uint32_t k = f.node_woff[d]; k < f.node_woff[d + 1]; k++) {
                f.epp_word.push_back(f.words[k]);
                f.epp_node.push_back(d);
            }
            open.push_back(d);
        }
        while (!open.empty()) { close(open.back()); open.pop_back(); }
    }
    return WEPP_OK;
}

}  // namespace wepp
