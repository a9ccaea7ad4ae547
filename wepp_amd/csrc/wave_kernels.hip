// wave_kernels.hip -- a read with MANY events at its positions: lane = list entry, one wavefront per 64 entries.
//
// The per-read walk (walk_kernels.hip) visits a read's events one after the other -- a chain of dependent gathers,
// ~1 us each: fine for the reads of a sequencing run (0-16 events), slow for the one read in a thousand that lists a
// frequently mutated position (17 to a few hundred events in its stream).  Round 2 cut such walks into jobs of 8
// events (gather + scan + walk + combine: ~85 us of latency for ~1 000 reads, the longest chain of the default step).
// Here the events of ONE read are spread over the lanes of one wave and nothing is walked:
//   * every lane loads up to WW_R entries of the read's position lists (an entry = a mutation of a stream node at a
//     listed position, with the node's subtree end: flatmat.hpp IxEnt) -- coalesced, the lists are contiguous;
//   * an all-pairs pass over the read's E entries (broadcasts, no memory) gives every entry what a sequential walk
//     would know on arrival: c_S in front of its node = c_S of an empty path + the deltas of the entries whose subtree
//     holds the node; the other entries of the same node (a node that mutates several listed positions); and, for the
//     two stretches of untouched nodes that start behind an entry -- its descendants, from node + 1, and what follows
//     its subtree, from end --, their c_S, where they stop (the next node, descendant start or subtree end of any
//     entry) and whether another entry owns the same stretch;
//   * every lane scores its node (the formula of the walk and of the sweep's node-by-node path, usher_mapper.cpp:
//     191-265, 455-456) and asks the range queries of its stretches (one byte of the sparse table, then four 16-byte
//     loads: flatmat.hpp), all lanes at once;
//   * a wave reduction leaves (score, rank, count, has_unique).
// A handful of dependent memory round trips per read, whatever its events.  Exact: the same nodes get the same scores
// as in the sequential walk (tests/walk_model.py is the CPU model of that walk; the GPU parity tests cover this kernel
// through every batch that holds such reads, and test_reads_with_many_events_vs_oracle aims at it).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_mat.hpp"
#include "place_dev.hpp"

namespace wepp {

namespace {

constexpr uint32_t WW_R = WAVE_WALK_MAX_EVENTS / 64;      // entries per lane
constexpr int PK_BIAS = 2;
constexpr uint32_t WW_WAVES = WW_R;                        // waves of a block: together they hold the largest read
constexpr uint32_t WW_WGS = 1024;                         // persistent grid: the waves loop over the list

struct Cand {
    int bs;
    uint32_t br, cnt, hu;
};
__device__ __forceinline__ void cand_take(Cand& c, int sc, uint32_t rk, uint32_t kk, uint32_t hu) {
    if (sc < c.bs) { c.bs = sc; c.br = rk; c.cnt = kk; c.hu = hu; }
    else if (sc == c.bs) { c.cnt += kk; if (rk < c.br) { c.br = rk; c.hu = hu; } }
}

// best statically eligible node of the untouched nodes [pos, stop) of an arena slice whose running c_S is `c`: the
// sparse table's byte says whether anything there can reach the bound, then the exact aggregate (as in k_walk)
__device__ __forceinline__ void range_candidate(const DevWalk& ix, const WcInfo& wi, uint32_t pos, uint32_t stop, int c, Cand& best, uint32_t& bytes) {
    const uint32_t len = stop - pos;
    const uint32_t lvl = len > 1 ? 32u - (uint32_t)__builtin_clz(len - 1) : 0u;
    const uint32_t mn = ix.sp[(size_t)wi.sp_off + (size_t)lvl * wi.n + pos];
    bytes += 1;
    if (mn == SP_NONE || (mn < SP_CLAMP && (int)mn + c > best.bs)) return;
    const uint32_t last = stop - 1, ba = pos / RQ_BLK, bl = last / RQ_BLK;
    SegNode ag{SCORE_INF_DEV, 0xFFFFFFFFu, 0u, 0u};
    auto join = [&](const SegNode x) {
        if (x.base < ag.base) ag = x;
        else if (x.base == ag.base) { ag.cnt += x.cnt; if (x.rank < ag.rank) { ag.rank = x.rank; ag.hu = x.hu; } }
    };
    if (ba == bl) {
        bytes += 16;
        if (pos == ba * RQ_BLK) join(ix.rq_pre[wi.node_off + last]);
        else if (stop == wi.n || stop == (ba + 1) * RQ_BLK) join(ix.rq_suf[wi.node_off + pos]);
        else
            for (uint32_t i = pos; i < stop; i++) {
                const NodeRec x = ix.nrec[wi.node_off + i];
                if (x.nstat & NS_ELIG0_DEV) {
                    const uint32_t hu = (x.nstat & NS_ROOT_DEV) ? 0u : (x.nstat & NS_MASKED_DEV) ? 1u :
                                        (((x.nstat >> 14) & NS_CNT_MASK_DEV) < (x.nstat & NS_CNT_MASK_DEV) ? 1u : 0u);
                    join(SegNode{x.base, x.rank, 1u, hu});
                }
            }
    } else {
        const uint32_t lo = ba + 1, hi = bl - 1;
        const SegNode none{SCORE_INF_DEV, 0xFFFFFFFFu, 0u, 0u};
        const uint32_t L = lo < hi ? 31u - (uint32_t)__builtin_clz(lo ^ hi) : 0u;
        const SegNode* trow = ix.rq_dst + wi.dst_off + (size_t)L * wi.rq_blocks;
        const SegNode s1 = ix.rq_suf[wi.node_off + pos], s2 = ix.rq_pre[wi.node_off + last];
        const SegNode s3 = lo <= hi ? trow[lo] : none, s4 = lo < hi ? trow[hi] : none;
        join(s1); join(s2); join(s3); join(s4);
        bytes += 64;
    }
    if (ag.cnt && ag.base + c <= best.bs) cand_take(best, ag.base + c, ag.rank, ag.cnt, ag.hu);
}

}  // namespace

// the lists of one read, as every wave that works on it sees them: lane j < k holds list j
struct ReadLists {
    uint32_t k, E, w, off, len, start;
    int c0;
};
__device__ __forceinline__ ReadLists read_lists(const DevMAT& m, const DevWalk& ix, const WcInfo& wi, uint32_t lane, uint32_t rd,
                                                const uint32_t* __restrict__ read_off, const uint32_t* __restrict__ read_word) {
    ReadLists L;
    const uint32_t so = read_off[rd];
    L.k = read_off[rd + 1] - so;                         // (k <= WALK16_K: k_route)
    L.w = lane < L.k ? read_word[so + lane] : 0u;
    L.off = L.len = 0;
    if (lane < L.k && w_pos(L.w) <= m.max_pos) {
        const uint32_t o0 = ix.ix_head[wi.head_off + w_pos(L.w)].off, o1 = ix.ix_head[wi.head_off + w_pos(L.w) + 1].off;
        L.off = o0;
        L.len = o1 - o0 - 1u;                            // (every list ends in a sentinel)
    }
    const uint32_t incl = wave_scan_add_u32(L.len);
    L.start = incl - L.len;
    L.E = min((uint32_t)__builtin_amdgcn_readlane((int)incl, 63), WAVE_WALK_MAX_EVENTS);   // (k_route admits no more)
    L.c0 = (int)__popcll(__ballot(lane < L.k && !rw_missing(L.w) && (rw_mut(L.w) & rw_ref(L.w)) == 0));
    return L;
}
// entry i of the concatenated lists: its place in the index and the read's word for its position
__device__ __forceinline__ void locate(const ReadLists& L, uint32_t i, uint32_t& e, uint32_t& sw) {
    e = NONE; sw = 0;
    for (uint32_t j = 0; j < L.k; j++) {
        const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)L.start, (int)j), lj = (uint32_t)__builtin_amdgcn_readlane((int)L.len, (int)j);
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)L.off, (int)j), wj = (uint32_t)__builtin_amdgcn_readlane((int)L.w, (int)j);
        if (i - sj < lj) { e = oj + (i - sj); sw = wj; }
    }
}
// the three adjustments of an entry (the delta -2 .. 2, the other two -1 .. 1), each biased by PK_BIAS in a byte of its
// own, and a one in the top byte: the sum over the <= 16 entries of one node (one per listed position) stays inside
__device__ __forceinline__ uint32_t pack_adjust(const IxEnt& ent, uint32_t sw) {
    int d = 0, adj = 0, dcom = 0;
    // descendants take the allele; the root also scores itself with it (usher_mapper.cpp:266-271)
    if (ent.end > ent.node + 1 || ent.node == 0) d = enter_delta(ent.word, sw);
    own_adjust(ent.word, sw, adj, dcom);
    return (uint32_t)(d + PK_BIAS) | (uint32_t)(adj + PK_BIAS) << 8 | (uint32_t)(dcom + PK_BIAS) << 16 | 1u << 24;
}

// The work of ONE wave on one read: the wave scores entries 64 mine .. 64 mine + 63 (lane = entry) and their stretches
// against all E <= 64 R entries of the read, which it holds once more for the broadcasts (lane + 64 r).
template <uint32_t R>
__device__ __forceinline__ void wave_read(const DevWalk& ix, const WcInfo& wi, uint32_t lane, const ReadLists& L, uint32_t mine, Cand& best, uint32_t& bytes) {
    const uint32_t E = L.E;
    const int c0 = L.c0;
    uint32_t bnode[R], bend[R], bpk[R];
    uint32_t node = NONE, end = NONE, rank = 0, nst = 0;
    int base = 0;
#pragma unroll
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t i = lane + 64 * r;
        bnode[r] = bend[r] = NONE; bpk[r] = 0;
        uint32_t e, sw;
        locate(L, i, e, sw);
        if (i < E && e != NONE) {
            const IxEnt ent = ix.ix_ent[e];
            bnode[r] = ent.node;
            bend[r] = ent.end;
            bpk[r] = pack_adjust(ent, sw);
            if (r == mine) {
                node = ent.node;
                end = ent.end;
                base = ent.base;
                rank = wi.has_pre ? ent.rank & IX_RANK_MASK : ent.rank;
                nst = ent.nstat;
            }
        }
    }
    // ---- all pairs: what a sequential walk would know at every entry.  Subtrees nest, so for entry (n, e):
    //   cb  = the deltas of the entries whose subtree holds n strictly inside (nl < n < el);
    //   T   = the packed sum over the entries of the same node (the lowest of them owns the node and its stretches);
    //   the stretch of untouched descendants starts at n + 1 with c_S = cb + the node's own deltas and stops at the first
    //   entry node or subtree end at or behind n + 1 (its own end at the latest: a leaf's stretch is empty);
    //   the stretch behind the subtree starts at e with cB = the deltas of the subtrees that hold e strictly inside and
    //   stops at the first entry node at or behind e, or subtree end behind e; of the entries that end at e the lowest
    //   owns it.  Differences wrap to huge values when the cut lies in front of the start, so plain minima do.
    int cb = 0, cB = 0;
    uint32_t T = 0, stopA = NONE, stopB = NONE, first_node = NONE;      // (the stretch from node 0 on is nobody's: the first lane asks for it)
    unsigned long long lower_same = 0ull, lower_end = 0ull;
    const uint32_t mg = mine * 64 + lane, sA = node + 1u, e2 = end << 1;
#pragma unroll
    for (uint32_t rl = 0; rl < R; rl++) {
        const uint32_t nl_max = E > 64 * rl ? min(64u, E - 64 * rl) : 0u;
        for (uint32_t ll = 0; ll < nl_max; ll++) {
            const uint32_t nl = (uint32_t)__builtin_amdgcn_readlane((int)bnode[rl], (int)ll), el = (uint32_t)__builtin_amdgcn_readlane((int)bend[rl], (int)ll);
            const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)bpk[rl], (int)ll);
            const int dl = (int)(pl & 0xFFu) - PK_BIAS;
            const uint32_t g = rl * 64 + ll, nl1 = nl + 1u, span = el - nl1, nl2 = nl << 1, el2 = (el << 1) - 1u;
            first_node = min(first_node, nl);
            const unsigned long long lower = __ballot(g < mg);
            if (dl != 0) {
                if (node - nl1 < span) cb += dl;
                if (end - nl1 < span) cB += dl;
            }
            const bool same = nl == node;
            T += same ? pl : 0u;
            lower_same |= __ballot(same) & lower;
            lower_end |= __ballot(el == end) & lower;
            stopA = min(stopA, min(nl - sA, el - sA));
            stopB = min(stopB, min(nl2 - e2, el2 - e2));       // (entry nodes at e cut, ends behind e cut)
        }
    }
    // ---- every lane: its node, its stretches ----
    if (node != NONE) {
        const bool owner = !((lower_same >> lane) & 1ull);
        const uint32_t cnt = PK_BIAS * (T >> 24);
        const int dsumT = (int)(T & 0xFFu) - (int)cnt, adjT = (int)((T >> 8) & 0xFFu) - (int)cnt, dcomT = (int)((T >> 16) & 0xFFu) - (int)cnt;
        if (owner) {
            const uint32_t nmut = nst & NS_CNT_MASK_DEV, ncom0 = (nst >> 14) & NS_CNT_MASK_DEV;
            const bool leaf = nst & NS_LEAF_DEV, masked = nst & NS_MASKED_DEV, root = nst & NS_ROOT_DEV;
            const int c = c0 + cb;
            if (root) { if (base + c + dsumT <= best.bs) cand_take(best, base + c + dsumT, rank, 1u, 0u); }
            else if (!masked) {
                const int sc = base + c + adjT, ncom = (int)ncom0 + dcomT;
                const bool elig = leaf ? (ncom > 0) : (ncom > 0 || ncom == (int)nmut);     // usher_mapper.cpp:455-456
                if (elig && sc <= best.bs) cand_take(best, sc, rank, 1u, ncom < (int)nmut ? 1u : 0u);
            }
            const uint32_t eA = min(sA + stopA, wi.n);
            if (sA < eA) range_candidate(ix, wi, sA, eA, c0 + cb + dsumT, best, bytes);
        }
        if (!((lower_end >> lane) & 1ull)) {
            const uint32_t eB = stopB >= 0x80000000u ? wi.n : min(end + ((stopB + 1u) >> 1), wi.n);     // (no cut behind e: the keys are below 2 n)
            if (end < eB) range_candidate(ix, wi, end, eB, c0 + cB, best, bytes);
        }
    }
    if (mine == 0 && lane == 0 && first_node != 0u && wi.n > 0) range_candidate(ix, wi, 0u, min(first_node, wi.n), c0, best, bytes);
}

// list[0 .. count[0]): reads with <= 64 events, a wave each; list[n_reads - count[1] .. n_reads), from the back: reads
// with more, the four waves of a block each (a wave scores 64 entries against all of them: the pass over the pairs is
// the time of such a read, ~20 instructions a pair), first.
__global__ __launch_bounds__(64 * WW_WAVES) void k_walk_wave(DevMAT m, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, uint32_t n_reads,
                                                                const uint32_t* __restrict__ read_off, const uint32_t* __restrict__ read_word,
                                                                const int32_t* __restrict__ root_score, uint32_t* __restrict__ best_bfs_j,
                                                                int32_t* __restrict__ score_out, uint32_t* __restrict__ num_best,
                                                                uint32_t* __restrict__ flags, unsigned long long* __restrict__ work_counter,
                                                                const uint32_t* __restrict__ wsid) {
    static_assert(WAVE_WALK_MAX_EVENTS == 64 * WW_WAVES, "a block holds the largest read");
    __shared__ int part_score[WW_WAVES];
    __shared__ uint32_t part_total[WW_WAVES], part_rank[WW_WAVES], part_hu[WW_WAVES];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_small = (uint32_t)__builtin_amdgcn_readfirstlane((int)count[0]), n_big = (uint32_t)__builtin_amdgcn_readfirstlane((int)count[1]);
    const DevWalk ix = m.walks[WC_SLOT];                 // the walk arena: every stream is a slice of it
    uint32_t bytes = 0, wave_bytes = 0;     // what the lanes / the wave as a whole asked memory for
    auto reduce = [&](const Cand& best, int& smin, uint32_t& total, uint32_t& rmin, bool& hu) {
        smin = wave_min_i32(best.cnt ? best.bs : 0x7FFFFFFF);
        const bool at = best.cnt && best.bs == smin;
        total = wave_sum_u32(at ? best.cnt : 0u);
        rmin = wave_min_u32(at ? best.br : 0xFFFFFFFFu);
        hu = __ballot(at && best.br == rmin && best.hu) != 0ull;
    };
    auto emit = [&](uint32_t rd, int smin, uint32_t total, uint32_t rmin, bool hu) {
        if (best_bfs_j) best_bfs_j[rd] = m.rank2bfs[rmin < m.N ? rmin : 0u];
        if (score_out) score_out[rd] = smin;
        if (num_best) num_best[rd] = total;
        if (flags) flags[rd] = hu ? WEPP_FLAG_HAS_UNIQUE_DEV : 0u;
    };
    for (uint32_t it = blockIdx.x; it < n_big; it += gridDim.x) {
        const uint32_t rd = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[n_reads - 1u - it]);
        const WcInfo wi = m.wc_info[wsid[rd]];
        const ReadLists L = read_lists(m, ix, wi, lane, rd, read_off, read_word);
        Cand best{root_score[rd] + 1, 0xFFFFFFFFu, 0u, 0u};        // the root always competes: nothing worse can win or tie
        if (64 * wv < L.E) {
            if (L.E <= 128) wave_read<2>(ix, wi, lane, L, wv, best, bytes);
            else wave_read<WW_R>(ix, wi, lane, L, wv, best, bytes);
            wave_bytes += (wv == 0 ? 8 + 12 * L.k + 80 + 16 : 0) + 32 * min(64u, L.E - 64 * wv);
        }
        int smin; uint32_t total, rmin; bool hu;
        reduce(best, smin, total, rmin, hu);
        if (lane == 0) { part_score[wv] = smin; part_total[wv] = total; part_rank[wv] = rmin; part_hu[wv] = hu ? 1u : 0u; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t v = 1; v < WW_WAVES; v++) {
                if (part_score[v] < smin) { smin = part_score[v]; total = part_total[v]; rmin = part_rank[v]; hu = part_hu[v]; }
                else if (part_score[v] == smin && part_total[v]) { total += part_total[v]; if (part_rank[v] < rmin) { rmin = part_rank[v]; hu = part_hu[v]; } }
            }
            emit(rd, smin, total, rmin, hu);
        }
        __syncthreads();
    }
    // (the small reads are dealt from the END of the grid: the first workgroups hold the large ones)
    for (uint32_t it = gridDim.x * WW_WAVES - 1u - (blockIdx.x * WW_WAVES + wv); it < n_small; it += gridDim.x * WW_WAVES) {
        const uint32_t rd = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[it]);
        const WcInfo wi = m.wc_info[wsid[rd]];
        const ReadLists L = read_lists(m, ix, wi, lane, rd, read_off, read_word);
        Cand best{root_score[rd] + 1, 0xFFFFFFFFu, 0u, 0u};
        wave_read<1>(ix, wi, lane, L, 0u, best, bytes);
        wave_bytes += 8 + 12 * L.k + 32 * L.E + 80 + 16;
        int smin; uint32_t total, rmin; bool hu;
        reduce(best, smin, total, rmin, hu);
        if (lane == 0) emit(rd, smin, total, rmin, hu);
    }
    wave_bytes += wave_sum_u32(bytes);
    if (work_counter && lane == 0 && wave_bytes)
        atomicAdd(work_counter + WALK_COUNTERS + ((blockIdx.x * WW_WAVES + wv) & (WALK_COUNTERS - 1)), (unsigned long long)wave_bytes);
}

hipError_t launch_walk_wave(const DevMAT& m, const uint32_t* list, const uint32_t* count, uint32_t n_reads, const uint32_t* d_read_off,
                            const uint32_t* d_read_word, const int32_t* root_score, uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best,
                            uint32_t* flags, unsigned long long* work_counter, const uint32_t* wsid, hipStream_t stream) {
    hipLaunchKernelGGL(k_walk_wave, dim3(WW_WGS), dim3(64 * WW_WAVES), 0, stream, m, list, count, n_reads, d_read_off, d_read_word, root_score,
                       best_bfs_j, score, num_best, flags, work_counter, wsid);
    return hipGetLastError();
}

}  // namespace wepp
