// flatmat.hpp -- host-side flattened mutation-annotated tree (MAT).
//
// The reference keeps the MAT as a pointer graph (MAT::Node / MAT::Tree,
// src/mutation_annotated_tree.hpp:80-152) and re-derives, for every
// (sample, node) pair, the root-to-node genotype by walking parents
// (src/usher_mapper.cpp:276-287).  Here the tree is flattened ONCE into
// DFS-ordered arrays that live in HBM; everything that does not depend on the
// read is folded into per-node constants.  See DESIGN.md section 3 for the
// layout and section 2 for the closed form the constants implement.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/wepp_place.h"

namespace wepp {

// ---- packed tree mutation word: pos:20 | ref:2 | par:4 | mut:4 | exit | leaf --
// `ref` is the INDEX (0..3) of the one-hot reference nucleotide (the loader
// builds ref_nuc as 1 << idx, mutation_annotated_tree.cpp:572).  `par` is the
// TRUE allele state of the parent genotype at this position (0 = no mutation
// above on the root path), recomputed by the flattener; Mutation::par_nuc of
// the reference (mutation_annotated_tree.hpp:48) is never read by the scorer.
// The two flag bits are only set on sweep-stream events.
constexpr uint32_t W_POS_MASK = 0xFFFFFu;
constexpr uint32_t W_EXIT = 1u << 30;   // event closes the subtree of the node that carries the word
constexpr uint32_t W_LEAF = 1u << 31;   // enter event of a leaf: no other node sees the allele
constexpr uint32_t W_PAD = 0x000FFFFFu; // padding event (position 0xFFFFF is reserved)
inline uint32_t w_pack(uint32_t pos, uint32_t ref_idx, uint32_t par, uint32_t mut) {
    return (pos & W_POS_MASK) | ((ref_idx & 3u) << 20) | ((par & 15u) << 22) | ((mut & 15u) << 26);
}
inline uint32_t w_pos(uint32_t w) { return w & W_POS_MASK; }
inline uint32_t w_refmask(uint32_t w) { return 1u << ((w >> 20) & 3u); }
inline uint32_t w_par(uint32_t w) { return (w >> 22) & 15u; }
inline uint32_t w_mut(uint32_t w) { return (w >> 26) & 15u; }

// ---- per-node static word ---------------------------------------------------
// nmut:14 | ncommon0:14 | leaf | masked | elig0 | root
constexpr uint32_t NS_CNT_MASK = 0x3FFFu;
constexpr uint32_t NS_LEAF = 1u << 28, NS_MASKED = 1u << 29, NS_ELIG0 = 1u << 30, NS_ROOT = 1u << 31;
constexpr int32_t SCORE_INF = 0x3FFFFFFF;

// ---- sweep-stream event meta byte: offset of the event's node in its block ---
constexpr uint8_t EV_OFF_MASK = 63;

constexpr uint32_t BLK_MAX_NODES = 64;    // one node per lane on the slow path
constexpr uint32_t BLK_MAX_EVENTS = 128;  // two event words per lane per load (blocks are padded to an even count)

struct BlkSum {          // read-independent summary of one sweep block
    int32_t base;        // min static score over statically eligible nodes (SCORE_INF if none)
    uint32_t rank;       // tie-break rank of that minimum
    uint32_t cnt;        // number of statically eligible nodes attaining `base`
    int32_t min_all;     // min static score over ALL nodes of the block (pruning bound)
    uint32_t node0, nn;  // first node / node count of the block (heavy path)
    uint32_t pad0, pad1; // 32-byte records: one s_load_dwordx8 per block
};

// Aggregate of a range of stream nodes, as the walk's range queries return it: the best statically
// eligible node (static score, its tie-break rank, how many tie it, and whether the node of that rank has
// mutations the sample does not share -- the has_unique flag of a node no listed position touches);
// identity = {SCORE_INF, ~0, 0, 0}.  16 bytes: one load.
struct SegNode {
    int32_t base;
    uint32_t rank;
    uint32_t cnt;
    uint32_t hu;
};
// One entry of a stream's position index: a mutation of stream node `node` at the list's position; the
// nodes that inherit the allele are (node, end); `up` = entry (same list) of the innermost enclosing
// mutation of the position or IX_NONE; `next_node` = node of the list's next entry (IX_NONE at the end);
// base / rank / nstat = the node's own record, so that a walk reads ONE 32-byte record per event.  The last
// entry of every list is a sentinel (node == IX_NONE).
struct IxEnt {
    uint32_t node, end, word, up;
    uint32_t next_node;
    int32_t base;
    uint32_t rank, nstat;
};
struct IxHead {            // per position: first entry of its list and that entry's node (IX_NONE: empty list)
    uint32_t off, first_node;
};
struct NodeRec {           // per stream node: static score, tie-break rank, flags (flatmat.hpp NS_*)
    int32_t base;
    uint32_t rank, nstat, pad;
};
constexpr uint32_t RQ_BLK = 16;
constexpr uint32_t IX_PRE_MIN_NODES = 8192, IX_PRE_MIN_LISTS = 3, IX_RANK_BITS = 24, IX_RANK_MASK = (1u << IX_RANK_BITS) - 1;              // nodes per block of the exact range query
constexpr uint32_t IX_NONE = 0xFFFFFFFFu;    // sentinel node index closing every position's list
constexpr uint8_t SP_NONE = 255, SP_CLAMP = 254;   // sparse table: no eligible node in the range / score >= 254

// One sweep stream: the events of an ancestor-closed subset of the nodes (a
// "crown": every node whose static score is <= tau, plus all their ancestors),
// or of the whole tree.  Node indices inside a stream are local (0..n-1, DFS
// order of the subset); nkey keeps the GLOBAL tie-break rank.
struct Stream {
    int32_t tau = 0x7FFFFFFF;          // reads with theta <= tau may use this stream
    uint32_t n = 0, NB = 0, cp_stride = 1;
    uint64_t E = 0;
    std::vector<int64_t> nkey;         // [n]  (base << 32) | global rank
    std::vector<uint32_t> nstat;       // [n]  (NS_ROOT only on local node 0 = the root)
    std::vector<uint32_t> ncnt;        // [n] window streams only: nodes an element stands for (1, or the ties of a pseudo-node)
    std::vector<uint32_t> blk_node0;   // [NB+1] blocks of <=64 consecutive nodes with <=128 events
    std::vector<uint32_t> blk_eoff;    // [NB+1] (event counts padded even)
    std::vector<BlkSum> blk_sum;       // [NB]
    std::vector<uint32_t> ev_word;     // [E]
    std::vector<uint8_t> ev_meta;      // [E]
    std::vector<uint8_t> ev_lb;        // [E] lower bound (clamped to 255) of the static score of every node the
                                       //     event can affect inside its block: enter -> min over the subtree of
                                       //     its node, exit -> min over the rest of the block
    // checkpoints: enter words of every node still open when a sequential sweep
    // reaches block i*cp_stride (node blk_node0[..]-1 unless it closes there, and its ancestors)
    std::vector<uint32_t> cp_off;      // [ncp+1]
    std::vector<uint32_t> cp_word;
    // ---- the same crown addressed by genome position, for the per-read walk (k_walk) ----
    // list of position p = entries [ix_head[p].off, ix_head[p+1].off) in stream order, the last one a sentinel
    std::vector<IxHead> ix_head;       // [max_pos + 2]
    std::vector<IxEnt> ix_ent;
    std::vector<uint8_t> ix_nest;      // [max_pos + 1] most entries of a position's list open at once (nested subtrees
                                       // INSIDE this stream), clamped to 255: bounds the stack of a walking read
    // ix_pre[0] = 1: the top byte of IxEnt::rank holds the pre-test value (sp encoding) of the nodes between the
    // list's previous entry and this one -- a superset of any range a walk finishes at this entry's node, so the
    // walk needs no sparse-table byte for it.  Only for trees of at most 2^24 nodes (ranks then fit 24 bits) and
    // streams of at least IX_PRE_MIN_NODES nodes (WEPP_IX_PRE_MIN_NODES; smaller ones stay in the caches and pass
    // the pre-test so often that the table byte fetched alongside the entry is the better deal).
    std::vector<uint32_t> ix_pre;
    std::vector<NodeRec> nrec;         // [n]
    // range queries over the statically eligible nodes.
    // (1) "can anything in [a, b) matter": a sparse table of the minimum score, sp[l * n + i] = min over
    //     [i, min(n, i + 2^l)) clamped to SP_CLAMP, SP_NONE when the range holds no eligible node.  The walk
    //     reads ONE byte: the entry at `a` of the first level whose 2^l reaches b -- a superset of the range
    //     (at most twice as long), so a miss is exact and a hit only means "ask the exact query".
    // (2) the exact aggregate of [a, b): nodes are grouped in blocks of RQ_BLK; rq_pre[i] / rq_suf[i]
    //     aggregate a node's block up to / from the node, and rq_dst is a disjoint sparse table over the
    //     block aggregates (row 0 = the blocks; row l, entry i = the aggregate from block i to the middle of
    //     its 2^(l+1)-aligned group of blocks, towards that middle): a range that spans blocks is suffix +
    //     table[l][first whole block] + table[l][last whole block] + prefix, l = the highest bit in which the
    //     two block indices differ -- four independent loads, whatever the range.
    uint32_t sp_levels = 1, rq_blocks = 0, rq_levels = 1;
    std::vector<uint8_t> sp;           // [sp_levels * n]
    std::vector<SegNode> rq_pre, rq_suf;   // [n]
    std::vector<SegNode> rq_dst;       // [rq_levels * rq_blocks]
    SegNode whole{};                   // aggregate of the whole stream: the answer for a read none of whose
                                       // positions is mutated in the stream
    // bytes one sweep reads whatever the reads are: event words, block offsets and summaries, and -- on
    // a crown (tau finite), where the per-event bounds are fetched with the event words -- one bound byte
    // per event; node keys / flags / event offsets are only touched by node-by-node evaluations
    uint64_t stream_bytes() const {
        return 4ull * E + (uint64_t)NB * (sizeof(BlkSum) + 4) + (tau != 0x7FFFFFFF ? E : 0ull);
    }
};

struct FlatMAT {
    uint32_t N = 0, n_leaves = 0, max_depth = 0, max_pos = 0;
    uint64_t M = 0, n_masked = 0;
    int32_t root_base = 0;             // static score of the root (D0(root) + masked-root cost)
    // node-major CSR in DFS pre-order (depth_first_expansion order,
    // mutation_annotated_tree.cpp:1143-1163 == .pb node_mutations order)
    std::vector<uint32_t> node_woff;   // [N+1]
    std::vector<uint32_t> words;       // [M] non-masked mutation words
    std::vector<int64_t> nkey;         // [N] (base << 32) | rank
    std::vector<uint32_t> nstat;       // [N]
    std::vector<uint32_t> rank2dfs;    // [N]
    std::vector<uint32_t> dfs2bfs;     // [N] BFS index j of each node (tie-break key, usher_common.cpp:391,400)
    std::vector<uint32_t> rank2bfs;    // [N] BFS index of the node with tie-break rank r (what a winner is reported as)
    std::vector<uint32_t> bfs2id;      // [N] caller id of bfs[k]            (host only)
    std::vector<uint32_t> dfs2id;      // [N] caller id of dfs[k]            (host only)
    std::vector<uint32_t> parent_dfs;  // [N] DFS index of parent (root: 0)  (host only)
    std::vector<uint32_t> dfs_end;     // [N] last DFS index of the subtree  (host only)
    std::vector<uint32_t> num_leaves;  // [N]                                (host only)
    std::vector<uint8_t> maxnest;      // [max_pos+1] most mutations at one position along any root path (clamped to 255):
                                       //             bounds the open intervals a walking read keeps on its stack
    // streams[0 .. n-2] = crowns of increasing tau, streams.back() = whole tree
    std::vector<Stream> streams;
    // wstreams[w] = the whole tree as the reads of genome window [w * WIN_STRIDE, w * WIN_STRIDE + WIN_SIZE) see it
    std::vector<Stream> wstreams;
    // wcrowns[w] = WINDOW CROWNS of genome window w, increasing tau (at most WC_MAX): the nodes n whose score for ANY
    // read confined to the window is at least out_w(n) <= tau -- out_w(n) = the positions OUTSIDE the window at which
    // the genotype n is scored against (its parent's, with n's own back-mutations applied) differs from the
    // reference: a read that lists nothing there pays one for each, whatever it lists inside the window -- plus
    // their ancestors.  A read of the window whose root score is r can only be placed on a node with out_w <= r, so
    // the crown with tau >= r holds every node that can win or tie: the bound is r itself, not r + |S| as for the
    // tree-wide crowns above (every listed position may lower a score by one there) -- an N-rich read, whose Ns cost
    // it nothing at the root, walks a crown of a few hundred nodes instead of one of a million.  Only the walk index
    // of these streams is used (positions of the window only); node indices are local, ranks global, as in `streams`.
    std::vector<std::vector<Stream>> wcrowns;
    // EPP event stream (epp_kernels.hip): every non-masked mutation twice, in the order a
    // pre-order walk applies and retracts it -- enter words when its node is entered, the same
    // words with W_EXIT once the node's subtree is done.  epp_node[i] = number of nodes
    // (pre-order indices) entered before event i takes effect: the node's own index for an
    // enter, one past the subtree's last index for an exit.  The genotype between two
    // consecutive events i, j is constant and belongs to the nodes [epp_node[i], epp_node[j]).
    std::vector<uint32_t> epp_word;    // [2M]
    std::vector<uint32_t> epp_node;    // [2M]
    // SEED SIGNATURES (seed_kernels.hip): the whole-tree stream cut into seed_chunks chunks of
    // seed_stride blocks (about a thousand nodes; every chunk starts at a checkpoint of the stream).  Nibble
    // (position p, chunk c) of seed_sig = OR of the mut_nuc masks of every mutation at p carried by a node of the
    // chunk or by an ancestor of the chunk's first node -- a superset of the alleles any genotype a node of the chunk
    // is scored against (usher_mapper.cpp:191-287) can hold at p.  A sample entry whose allele excludes its reference
    // base costs one at a node unless that genotype offers a compatible allele there (:293-389), so a sample with s0
    // such entries, H of which the chunk's signature could serve, scores at least s0 - H on every node of the chunk.
    // Row p = seed_row_words dwords, 8 chunks per dword; empty when the tree is too small or the table too large.
    uint32_t seed_stride = 0, seed_chunks = 0, seed_row_words = 0;
    std::vector<uint32_t> seed_sig;    // [(max_pos + 2) * seed_row_words]
    const Stream& full() const { return streams.back(); }
};

// window streams (flatmat.cpp): a read whose positions span at most WIN_SIZE - WIN_STRIDE = 1536 (a 1.2 kb
// amplicon) lies inside the window that starts at its first position rounded down to the stride
// (-DWEPP_WIN_SIZE / -DWEPP_WIN_STRIDE: a test-only build with windows of a few dozen positions lets the fuzz trees --
// masked nodes, multi-allelic alleles, repeated positions, back-mutations -- straddle window edges on the GPU,
// tools/build_variant.sh win64; the product is built with the defaults.)  Positions from MAX_WINDOWS * WIN_STRIDE on
// lie in no window: reads there take the tree-wide streams (wepp_mat_stats::window_uncovered_positions).
#ifndef WEPP_WIN_SIZE
#define WEPP_WIN_SIZE 2560
#endif
#ifndef WEPP_WIN_STRIDE
#define WEPP_WIN_STRIDE 1024
#endif
constexpr uint32_t WIN_SIZE = WEPP_WIN_SIZE, WIN_STRIDE = WEPP_WIN_STRIDE, MAX_WINDOWS = 32;
static_assert(WIN_STRIDE > 0 && WIN_SIZE >= 2 * WIN_STRIDE && WIN_SIZE <= 3 * WIN_STRIDE, "a position lies in at most three windows");
constexpr uint32_t MAX_STREAMS = 16;   // also the size of the stream arrays of wepp_mat_stats
// window crowns: at most WC_MAX per window, tau = root score + 0 .. WC_MAX_DTAU, none larger than WC_MAX_NODES nodes;
// in plan ids and diagnostics they all share ONE stream slot, the last one (the tree-wide streams use at most
// MAX_STREAMS - 1 slots): which window crown a read walks is a per-read value (k_route)
constexpr uint32_t WC_MAX = 7, WC_MAX_DTAU = 5, WC_MAX_NODES = 1u << 19, WC_SLOT = MAX_STREAMS - 1;

// seed chunks: blocks of the whole-tree stream per chunk (rounded up to the stream's checkpoint stride), most chunks
// (a sample's per-chunk counters are bytes in LDS), largest signature table built
constexpr uint32_t SEED_CHUNK_BLOCKS = 16, SEED_MAX_CHUNKS = 32768;
constexpr uint64_t SEED_MAX_SIG_BYTES = 1ull << 30;
constexpr uint32_t SEED_MAX_POS = (1u << 18) - 1;   // the position bitmap of a sample shares the workgroup's LDS with its counters

// flatten_tree calls of this process that built the full image (not topology_only): lets a test see that a
// multi-device run flattened once
uint64_t flatten_count();

// Returns WEPP_OK or an error code; `err` receives the message.
// topology_only: stop after the orders / parents / per-node flags (no tie-break ranks, sweep streams, EPP stream)
int flatten_tree(const wepp_tree_desc& t, FlatMAT& out, std::string& err, bool topology_only = false);

}  // namespace wepp

// the object behind wepp_flat_t (include/wepp_place.h): a flat image on the host, built once, uploaded to any number
// of devices (wepp_mat_upload) and inspected by the CPU tests (wepp_flat_get)
struct wepp_flat {
    wepp::FlatMAT f;
    std::vector<int32_t> wc_tau;       // (wepp_flat_get "wc_tau" / "wc_nodes": filled at the first request)
    std::vector<uint32_t> wc_nodes;
};
