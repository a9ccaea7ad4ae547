// place_kernels.hip -- CDNA4 (gfx950) kernels of the read-placement engine.
//
// What is computed: for every read, exactly what the two passes of the
// reference's per-sample loop leave behind (src/usher_common.cpp:386-446, each
// iteration being mapper2_body, src/usher_mapper.cpp:168-506): the minimum
// parsimony score over eligible nodes, the number of eligible nodes attaining
// it, and the winner under the (num_leaves, BFS index) tie-break.
//
// How (DESIGN.md sections 2-4): the score of node n for read S is
//     score(n) = base(n) + c_S(parent(n)) + adj_S(n)
// where base(n) is read-independent, c_S is the read-dependent correction of
// the parent genotype and adj_S(n) is non-zero only when n itself mutates a
// position listed in S.  c_S changes only at "events": entering / leaving the
// subtree of a node that mutates a position of S.  The tree is stored as a
// DFS-ordered event stream cut into blocks of <=64 nodes / <=128 events, with a
// read-independent summary per block.  One wavefront sweeps the stream once
// for a TILE of up to 64 reads (lane = read): blocks without an event for a
// read cost that read one summary update; blocks with events are re-evaluated
// node by node (lane = node) for just the reads concerned.
//
// Work skipping (exact): score(n) >= base(n) - |S| for every node, and the best
// score is <= the root's score, so a read with theta = score(root) + |S| only
// needs the "crown" of nodes with base <= theta (plus ancestors).  k_route
// computes theta per read, k_scatter groups the reads by the smallest crown
// stream that covers them, and k_sweep runs once per non-empty group on that
// stream (the whole tree being the last one).
//
// Integer work only: no MFMA.  The bound is the event stream (HBM/L2 bytes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_mat.hpp"

namespace wepp {

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

// -DWEPP_SWEEP_STATS: per-stream event counters of the sweep (a profiling build, never shipped):
// [tier][0] block visits, [1] blocks with a bitmap hit, [2] hit events, [3] (hit event, read) matches,
// [4] node-by-node evaluations, [5] of which reached the reduction, [6] blocks with a summary update, [7] waves
// and wave cycles (s_memtime) by section: [tier][8] set-up, [9] blocks without a hit, [10] hit blocks without
// a node-by-node evaluation, [11] hit blocks with one (light part), [12] the evaluations themselves,
// [13..15] parts of the set-up: until the bitmap is cleared, until the reads are staged, the checkpoint
#ifdef WEPP_SWEEP_STATS
constexpr int NSTAT = 24;
__device__ unsigned long long g_sweep_stats[MAX_STREAMS * NSTAT];
#define STAT_DECL uint32_t st_[8] = {0, 0, 0, 0, 0, 0, 0, 1}; unsigned long long tt_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    unsigned long long t0_ = __builtin_amdgcn_s_memtime()
#define STAT_ADD(i, v) st_[i] += (uint32_t)(v)
#define STAT_NOW() __builtin_amdgcn_s_memtime()
#define STAT_T(i, from) tt_[i] += __builtin_amdgcn_s_memtime() - (from)
#define STAT_WAIT() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAT_FLUSH(tier)                                                                      \
    if (lane == 0) {                                                                          \
        for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_sweep_stats[(tier) * NSTAT + i_], (unsigned long long)st_[i_]); \
        for (int i_ = 0; i_ < 16; i_++) atomicAdd(&g_sweep_stats[(tier) * NSTAT + 8 + i_], tt_[i_]);                 \
    }
#else
#define STAT_DECL
#define STAT_ADD(i, v)
#define STAT_NOW() 0ull
#define STAT_T(i, from)
#define STAT_WAIT()
#define STAT_FLUSH(tier)
#endif
#ifndef WEPP_DENSE_MIN_HITS
#define WEPP_DENSE_MIN_HITS 3
#endif
constexpr int DENSE_MIN_HITS = WEPP_DENSE_MIN_HITS;   // hit events per block from which the lane = event lookup pays
constexpr uint32_t DENSE_WAVES = wepp::DENSE_WAVES_PER_WG;
#ifndef WEPP_OWN_WORDS
#define WEPP_OWN_WORDS 4
#endif
constexpr uint32_t OWN_WORDS = WEPP_OWN_WORDS;        // read words per lane kept in registers by the plain sweep

// ---- word field helpers ------------------------------------------------------
// tree / event word: pos:20 | ref idx:2 | par:4 | mut:4 | exit | leaf (flatmat.hpp)
// read word:         pos:20 | ref:4 | mut:4 | missing           (wepp_place.h)
__device__ __forceinline__ uint32_t w_pos(uint32_t w) { return w & 0xFFFFFu; }
__device__ __forceinline__ uint32_t tw_ref(uint32_t w) { return 1u << ((w >> 20) & 3u); }
__device__ __forceinline__ uint32_t tw_par(uint32_t w) { return (w >> 22) & 15u; }
__device__ __forceinline__ uint32_t tw_mut(uint32_t w) { return (w >> 26) & 15u; }
__device__ __forceinline__ uint32_t rw_ref(uint32_t w) { return (w >> 20) & 15u; }
__device__ __forceinline__ uint32_t rw_mut(uint32_t w) { return (w >> 24) & 15u; }
__device__ __forceinline__ uint32_t rw_missing(uint32_t w) { return (w >> 28) & 1u; }

// f(x) = cost of allele state x for the read entry `s` minus its cost for an
// empty read; x == 0 means "no mutation on the root path" (then the read is
// compared with ITS OWN ref_nuc, usher_mapper.cpp:302-305,342).
__device__ __forceinline__ int f_state(uint32_t x, uint32_t tref, uint32_t s) {
    int c0 = (x != 0 && x != tref) ? 1 : 0;                                   // usher_mapper.cpp:426-437
    int cs = rw_missing(s) ? 0 : (((rw_mut(s) & (x ? x : rw_ref(s))) == 0) ? 1 : 0);  // :295,314-320,342
    return cs - c0;
}
// change of c_S for the descendants of a node carrying tree word `w`
__device__ __forceinline__ int enter_delta(uint32_t w, uint32_t s) {
    return f_state(tw_mut(w), tw_ref(w), s) - f_state(tw_par(w), tw_ref(w), s);
}
// own-score / common-count adjustments of the node carrying `w`
// (usher_mapper.cpp:205-264: "common" test with the sample vs. without it)
__device__ __forceinline__ void own_adjust(uint32_t w, uint32_t s, int& adj_score, int& adj_common) {
    const uint32_t ref = tw_ref(w), par = tw_par(w), mut = tw_mut(w);
    const int static_common = (mut == ref) ? 1 : 0;
    const int static_sub = static_common ? ((par != 0 && par != ref) ? 1 : 0) : 0;
    int actual_common, actual_sub;
    if (rw_missing(s)) { actual_common = 1; actual_sub = 0; }                  // :210-212
    else {
        actual_common = ((rw_mut(s) & mut) != 0) ? 1 : 0;                       // :215
        actual_sub = actual_common ? (((rw_mut(s) & (par ? par : rw_ref(s))) == 0) ? 1 : 0) : 0;
    }
    adj_score += static_sub - actual_sub;
    adj_common += actual_common - static_common;
}

// wave-wide minimum, returned wave-uniform: four DPP row shifts leave the minimum of every row of
// 16 lanes in its last lane, four readlanes and scalar mins finish (no LDS permutes)
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x111, 0xF, 0xF, false));   // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x112, 0xF, 0xF, false));   // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x114, 0xF, 0xF, false));   // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, 0x118, 0xF, 0xF, false));   // row_shr:8
    const int a = __builtin_amdgcn_readlane(v, 15), b = __builtin_amdgcn_readlane(v, 31);
    const int c = __builtin_amdgcn_readlane(v, 47), d = __builtin_amdgcn_readlane(v, 63);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xF, 0xF, false));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return min(min(a, b), min(c, d));
}

// wave-wide sum, returned wave-uniform (the same DPP pattern)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15) + (uint32_t)__builtin_amdgcn_readlane((int)v, 31) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 47) + (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// wave-wide inclusive scans (sum / max of unsigned values), every lane its own result: four DPP row shifts scan the
// rows of 16 lanes, two row broadcasts (lane 15 -> next row, lane 31 -> rows 2 and 3) carry across the rows
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return v;
}

// lower_bound over a position-sorted slice of read words; returns the entry
// with exactly `pos` or NONE.
template <typename SPtr>
__device__ __forceinline__ uint32_t find_entry(SPtr S, uint32_t off, uint32_t k, uint32_t pos) {
    uint32_t lo = 0, hi = k;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        uint32_t p = w_pos(S[off + mid]);
        if (p < pos) lo = mid + 1; else hi = mid;
    }
    if (lo < k) {
        uint32_t s = S[off + lo];
        if (w_pos(s) == pos) return s;
    }
    return NONE;
}

}  // namespace

// -----------------------------------------------------------------------------
// k_route: theta(read) = score(root) + |S| -> index of the smallest stream whose
// tau covers it.  Per-(block, tier) counts go to blk_counts, per-tier totals and
// the largest read of each tier to tier_info.
// -----------------------------------------------------------------------------
__global__ __launch_bounds__(ROUTE_THREADS) void k_route(DevMAT m, const uint32_t* __restrict__ read_off,
                                                          const uint32_t* __restrict__ read_word, uint32_t n_reads,
                                                          int use_crowns, uint32_t walk_max_events, uint32_t job_events,
                                                          uint32_t stack8, uint32_t stack16,
                                                          uint32_t* __restrict__ job_n, uint8_t* __restrict__ tier_of,
                                                          int32_t* __restrict__ root_score,
                                                          uint32_t* __restrict__ blk_counts,
                                                          uint32_t* __restrict__ tier_info,
                                                          uint32_t* __restrict__ slot_in_blk,
                                                          uint32_t* __restrict__ tier_info_next,
                                                          uint32_t* __restrict__ wsid) {
    __shared__ uint32_t cnt[MAX_PLANS], mx[MAX_PLANS], jobs_of[2 * MAX_STREAMS], open_of[4], events_of[2];
    // the counters of the NEXT call (the handle alternates between two sets) are cleared here: no memset
    // (two fill kernels, ~10 us) in front of every call
    if (blockIdx.x == 0 && threadIdx.x < TI_WORDS) tier_info_next[threadIdx.x] = 0;
    if (threadIdx.x < MAX_PLANS) { cnt[threadIdx.x] = 0; mx[threadIdx.x] = 0; }
    if (threadIdx.x < 4) open_of[threadIdx.x] = 0;
    if (threadIdx.x < 2) events_of[threadIdx.x] = 0;
    if (threadIdx.x < 2 * MAX_STREAMS) jobs_of[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t per = (n_reads + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per, hi = min(n_reads, lo + per);
    // four reads per thread and round: their offsets, then their first two words, are requested together
    // (one read after the other, every read cost its thread three memory round trips in a row)
    for (uint32_t r0 = lo + threadIdx.x; r0 < hi; r0 += 4 * blockDim.x) {
      uint32_t so4[4], k4[4], fw[4][2];
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
          const uint32_t r = r0 + u * blockDim.x;
          so4[u] = r < hi ? read_off[r] : 0u;
          k4[u] = r < hi ? read_off[r + 1] - so4[u] : 0u;
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; u++)
#pragma unroll
          for (uint32_t j = 0; j < 2; j++) fw[u][j] = k4[u] > j ? read_word[so4[u] + j] : 0u;
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
        const uint32_t r = r0 + u * blockDim.x;
        if (r >= hi) break;
        const uint32_t so = so4[u], k = k4[u];
        int c = 0;
        auto count = [&](uint32_t sw) { if (!rw_missing(sw)) c += ((rw_mut(sw) & rw_ref(sw)) == 0) ? 1 : 0; };
        if (k > 0) count(fw[u][0]);
        if (k > 1) count(fw[u][1]);
        for (uint32_t j = 2; j < k; j++) count(read_word[so + j]);
        for (uint32_t w = m.node_woff[0]; w < m.node_woff[1]; w++) {   // the root's own mutations
            const uint32_t tw = m.words[w];
            const uint32_t sw = find_entry(read_word, so, k, w_pos(tw));
            if (sw != NONE) c += enter_delta(tw, sw);
        }
        root_score[r] = m.root_base + c;       // the root always competes: an upper bound of the best score
        const int theta = m.root_base + c + (int)k;
        uint32_t t = m.n_streams - 1;
        if (use_crowns)
            for (uint32_t i = 0; i + 1 < m.n_streams; i++)
                if (theta <= m.tau[i]) { t = i; break; }
        // how the read is placed (device_mat.hpp): by walking its own events when it lists few positions and
        // the intervals it can hold open at once fit the walk's stack, by a sweep of the stream otherwise
        // A walk runs a read's events one after the other: reads with many events in their stream (a
        // frequently mutated position) are left to the sweeps, whose cost does not depend on it.
        uint32_t cls = PLAN_SWEEP;
        // how a read with at most WALK16_K entries would walk a stream whose position index starts at (ix_head,
        // ix_nest): plain, cut into jobs, or not at all (more open intervals than a walk's stack holds)
        auto classify = [&](const IxHead* ix_head, const uint8_t* ix_nest, uint32_t& nj_out, uint32_t& open_out, uint32_t& ev_out) -> uint32_t {
            uint32_t open_max = 0, events = 0, longest = 0;
            for (uint32_t j = 0; j < k; j++) {
                const uint32_t p = w_pos(j < 2 ? fw[u][j] : read_word[so + j]);
                if (p <= m.max_pos) {
                    open_max += (uint32_t)ix_nest[p];
                    const uint32_t len = ix_head[p + 1].off - ix_head[p].off - 1u;      // (every list ends in a sentinel)
                    events += len;
                    longest = max(longest, len);
                }
            }
            open_out = open_max;
            ev_out = events;
            nj_out = 0;
            // (stack8 <= WALK8_STACK, stack16 <= WALK16_STACK: the stack rows the walks' workgroups get; the few
            // reads that could hold more intervals open are left to the sweeps)
            if (events <= walk_max_events) {
                if (k <= WALK8_K && open_max <= stack8) return PLAN_WALK8;
                if (open_max <= stack16) return PLAN_WALK16;
                return PLAN_SWEEP;
            }
            if (open_max > stack16) return PLAN_SWEEP;
            // many events: jobs of about `job_events`, cut at quantiles of the longest list
            const uint32_t small = (k <= WALK8_K && open_max <= stack8) ? 1u : 0u;
            const uint32_t je = small ? (job_events & 0xFFFFu) : (job_events >> 16);   // (per class, capi.cpp)
            nj_out = min((events + je - 1) / je, longest);
            return small ? PLAN_WALKC8 : PLAN_WALKC16;
        };
        // a read inside one genome window: the window crown its ROOT score admits (flatmat.hpp: wcrowns) holds
        // every node that can win or tie -- far fewer than the tree-wide crown of theta = root score + |S|
        uint32_t sid = NONE, wi = 0;
        bool in_win = false;                   // all listed positions inside genome window wi
        if (use_crowns && k > 0) {
            const uint32_t p_lo = w_pos(fw[u][0]), p_hi = w_pos(k > 1 ? read_word[so + k - 1] : fw[u][0]);
            wi = p_lo / WIN_STRIDE;
            in_win = p_hi < wi * WIN_STRIDE + WIN_SIZE;
            if (in_win && wi < m.wc_windows) {
                const int rs = m.root_base + c;
                for (uint32_t i = 0; i < WC_MAX; i++) {
                    const WcInfo* q = m.wc_info + wi * WC_MAX + i;
                    const uint32_t qn = q->n;
                    if (!qn) break;
                    if (rs <= q->tau) { if (qn < m.walks[t].n) sid = wi * WC_MAX + i; break; }
                }
            }
        }
        if (walk_max_events && k <= WALK16_K) {
            uint32_t nj = 0, open_max = 0, events = 0;
            if (sid != NONE) {
                const WcInfo* q = m.wc_info + sid;
                const DevWalk& ar = m.walks[WC_SLOT];
                cls = classify(ar.ix_head + q->head_off, ar.ix_nest + q->nest_off, nj, open_max, events);
                if (cls != PLAN_SWEEP) { wsid[r] = sid; t = WC_SLOT; }
            }
            if (cls == PLAN_SWEEP) cls = classify(m.walks[t].ix_head, m.walks[t].ix_nest, nj, open_max, events);
            if (cls == PLAN_WALK8 || cls == PLAN_WALK16) {
                // the deepest stack a walk of the class can need in this call: its kernel's LDS request
                atomicMax(&open_of[cls], open_max);
            } else if (cls == PLAN_WALKC8 || cls == PLAN_WALKC16) {
                const uint32_t small = cls == PLAN_WALKC8 ? 1u : 0u;
                job_n[r] = nj;
                atomicAdd(&events_of[small ? 0 : 1], events);
                atomicAdd(&jobs_of[(small ? 0u : MAX_STREAMS) + t], nj);
                atomicMax(&open_of[small ? 2 : 3], open_max);
            }
        }
        // (use_crowns & 2 -- wepp_best_nodes, which lists nodes and so takes streams of real nodes only: every read
        // inside a window whose stream is the window's candidate crown takes it when it is the smaller one)
        if (cls == PLAN_SWEEP && in_win && wi < m.n_windows &&
            ((use_crowns & 2) ? m.win_n[wi] < m.walks[t].n
                              : k > WIN_MIN_ENTRIES ? (m.win_n[wi] < m.walks[t].n || t + 1 == m.n_streams) : (sid == NONE && t + 1 == m.n_streams))) {
            // many entries, all inside one genome window: a tile of such reads sweeps the window's stream -- the window's
            // candidates (a crown of a few thousand nodes, whatever the root score) or, for the reads no crown serves,
            // the whole tree as the window sees it
            cls = PLAN_WIN;
            t = wi;
        } else if (cls == PLAN_SWEEP && sid != NONE) {
            // it cannot walk (more than WALK16_K entries or too deep a stack): waves of its own sweep its window crown (k_sweep_arena)
            wsid[r] = sid;
            t = WC_SLOT;
        }
        t = plan_id(cls, t);
        tier_of[r] = (uint8_t)t;
        slot_in_blk[r] = atomicAdd(&cnt[t], 1u);     // position among this block's reads of the plan (k_scatter)
        atomicMax(&mx[t], k);
      }
    }
    __syncthreads();
    if (threadIdx.x < MAX_PLANS) {
        blk_counts[blockIdx.x * MAX_PLANS + threadIdx.x] = cnt[threadIdx.x];
        if (cnt[threadIdx.x]) {
            atomicAdd(&tier_info[TI_COUNT + threadIdx.x], cnt[threadIdx.x]);
            atomicMax(&tier_info[TI_MAXK + threadIdx.x], mx[threadIdx.x]);
        }
    }
    if (threadIdx.x < 2 * MAX_STREAMS && jobs_of[threadIdx.x]) atomicAdd(&tier_info[TI_JOBS + threadIdx.x], jobs_of[threadIdx.x]);
    if (threadIdx.x < 4 && open_of[threadIdx.x]) atomicMax(&tier_info[TI_OPEN + threadIdx.x], open_of[threadIdx.x]);
    if (threadIdx.x < 2 && events_of[threadIdx.x]) atomicAdd(&tier_info[TI_EVENTS + threadIdx.x], (events_of[threadIdx.x] + 63) >> 6);
}

// -----------------------------------------------------------------------------
// k_scatter: list[] = read indices grouped by tier (same block decomposition as
// k_route; a block's reads of one tier occupy a contiguous range).
// -----------------------------------------------------------------------------
__global__ __launch_bounds__(ROUTE_THREADS) void k_scatter(const uint8_t* __restrict__ tier_of,
                                                            const uint32_t* __restrict__ slot_in_blk, uint32_t n_reads,
                                                            const uint32_t* __restrict__ blk_counts,
                                                            uint32_t* __restrict__ tier_info,
                                                            uint32_t* __restrict__ list) {
    __shared__ uint32_t base[MAX_PLANS], before[MAX_PLANS];
    if (threadIdx.x < MAX_PLANS) before[threadIdx.x] = 0;
    __syncthreads();
    // reads of every tier in the blocks before this one: thread = (tier, one earlier block in 64), so that a
    // wave's sixteen-lane groups read whole 64-byte rows and only four lanes of a wave add to the same LDS
    // word (one thread per row with sixteen adds each serialised 64 lanes on every word)
    {
        const uint32_t t = threadIdx.x & (MAX_PLANS - 1), c0 = threadIdx.x / MAX_PLANS;
        uint32_t acc = 0;
        for (uint32_t b = c0; b < blockIdx.x; b += blockDim.x / MAX_PLANS) acc += blk_counts[b * MAX_PLANS + t];
        if (acc) atomicAdd(&before[t], acc);
    }
    __syncthreads();
    if (threadIdx.x < MAX_PLANS) {
        const uint32_t t = threadIdx.x;
        uint32_t off = 0;                       // start of plan t in the list
        for (uint32_t i = 0; i < t; i++) off += tier_info[TI_COUNT + i];
        base[t] = off + before[t];
        if (blockIdx.x == 0) {
            tier_info[TI_OFF + t] = off;
            if (t == MAX_PLANS - 1) tier_info[TI_OFF + MAX_PLANS] = off + tier_info[TI_COUNT + t];
        }
    }
    __syncthreads();
    const uint32_t per = (n_reads + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per, hi = min(n_reads, lo + per);
    // no atomics here.  Four reads per thread and round: their loads are issued together (a thread's
    // reads used to cost it one memory round trip after the other: 28 us per 1 M reads)
    for (uint32_t r0 = lo + threadIdx.x; r0 < hi; r0 += 4 * blockDim.x) {
        uint32_t t[4], sl[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t r = r0 + u * blockDim.x;
            t[u] = r < hi ? tier_of[r] : 0u;
            sl[u] = r < hi ? slot_in_blk[r] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t r = r0 + u * blockDim.x;
            if (r < hi) list[base[t[u]] + sl[u]] = r;
        }
    }
}

// -----------------------------------------------------------------------------
// k_first_pos: sort key of the reads of one stream's list = first listed position (reads that
// list nothing first).  Sorted by it, the reads of a tile list the same or neighbouring
// positions: the tile looks at an event of the stream once per DISTINCT position, every read
// that lists it takes the delta in the same instructions, and equal reads are evaluated once.
// -----------------------------------------------------------------------------
__global__ void k_first_pos(const uint32_t* __restrict__ list, uint32_t n, const uint32_t* __restrict__ read_off,
                            const uint32_t* __restrict__ read_word, uint32_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = list[i], so = read_off[r];
    keys[i] = (read_off[r + 1] > so) ? w_pos(read_word[so]) + 1u : 0u;
}

// sort key of the reads of a walk class's list: (stream, first listed position) -- the list is grouped by stream
// already (plan ids ascend), so a sort by this key reorders the reads inside every stream's range only
__global__ void k_walk_keys(const uint32_t* __restrict__ list, uint32_t n, const uint8_t* __restrict__ tier_of,
                            const uint32_t* __restrict__ read_off, const uint32_t* __restrict__ read_word,
                            uint32_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = list[i], so = read_off[r];
    const uint32_t first = (read_off[r + 1] > so) ? w_pos(read_word[so]) + 1u : 0u;
    keys[i] = (plan_index(tier_of[r]) << SORT_KEY_BITS) | first;
}

// -----------------------------------------------------------------------------
// The sweep.  grid = ntiles * nchunks single-wave workgroups.
// LDS: [bm_words] position bitmap of the tile, then (S_IN_LDS) the tile's read
// words.  part_* receive one (score, rank, count) per (chunk, read).
// -----------------------------------------------------------------------------
// S_IN_LDS: the tile's read words are staged in LDS (else read from global memory: reads
// longer than MAX_TILE_ENTRIES words).  DENSE: additionally keep a tile-sorted position
// index in LDS and resolve blocks with many hit events with lane = event (long reads).
// Plain LDS variant (short reads): the first OWN_WORDS words of a lane's read also sit in registers,
// so that an event whose position is in the tile's bitmap costs every read a few compares instead
// of a binary search through LDS (99.6 % of 150 bp reads list at most four positions; longer
// ones search the rest of their words in LDS).
// WIN (with DENSE; window plans of long reads): the reads of the tile all lie inside one genome window of
// WIN_SIZE positions from `key_cap` (= the window's first position).  Instead of the bitmap and the sorted keys
// the workgroup keeps, per window position, the 64-bit mask of the tile's reads that list it and the index of
// their words in a position-major copy of the tile's read words (the position bits of a copied word hold the
// lane of its read).  A block's hit events are then resolved with lane = (event, read) pair: the events' match
// counts are scanned, every pair finds its event through a marker array and adds its contribution to the
// read's accumulators -- two or three rounds of 64 pairs for a block of a 1.2 kb amplicon tile, where a loop
// over the events (lane = read) or over each event's reads (lane = event) takes tens of nearly empty rounds.
template <bool S_IN_LDS, bool DENSE, bool WIN = false>
__device__ __forceinline__ void sweep_tile(
    const DevStream& ms, uint32_t wg, uint32_t lds_word0, uint32_t bm_words, uint32_t max_pos, uint32_t ent_cap,
    uint32_t key_cap, const uint32_t* __restrict__ read_off,
    const uint32_t* __restrict__ read_word, const int32_t* __restrict__ root_score,
    const uint32_t* __restrict__ list, uint32_t n_list, uint32_t T,
    uint32_t ntiles, uint32_t blocks_per_chunk, int32_t* __restrict__ part_score, uint32_t* __restrict__ part_rank,
    uint32_t* __restrict__ part_cnt) {
    constexpr bool OWN = S_IN_LDS && !DENSE;
    static_assert(!WIN || (DENSE && S_IN_LDS), "the window table lives in the dense variant's workgroup");
    STAT_DECL;
    const DevStream& m = ms;
    // Plain variant: one wave = one tile of reads and one chunk of the stream; `wg` is the wave's
    // index among the sweeps of its plan and lds_word0 the start of its private LDS region (the
    // waves of a workgroup never interact: they are grouped only because a CU holds at most 16
    // LDS-using workgroups, i.e. 4 waves per SIMD with single-wave workgroups).
    // DENSE variant: a workgroup = one tile; its DENSE_WAVES waves share the tile's LDS structures
    // (so that the larger footprint does not cost occupancy) and each sweeps its own chunk.
    // LDS: [bm_words] bitmap | [ent_cap] read words (tile order) | DENSE only: [key_cap, pow2]
    // tile-sorted keys pos:19|idx:13 | [ent_cap] owner lane of each entry (bytes) | per wave [3*64] accumulators
    constexpr uint32_t NW = DENSE ? DENSE_WAVES : 1;
    extern __shared__ uint32_t lds[];
    uint32_t* bitmap = lds + lds_word0;
    uint32_t* S_lds = bitmap + bm_words;
    uint32_t* skey = S_lds + ent_cap;
    uint8_t* owner = reinterpret_cast<uint8_t*>(skey + key_cap);
    const uint32_t lane = threadIdx.x & 63;
    // wave-uniform values are pinned to scalar registers: block offsets, summaries and loop
    // control then run on the scalar unit
    const uint32_t wv = DENSE ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;
    int* acc = reinterpret_cast<int*>(owner + ent_cap) + wv * 192;   // net[64], H[64], bound[64] of this wave
    // DENSE: wtab[p - plo] = index of the first sorted key at position p (0xFFFF = none) for the
    // DENSE_WINDOW positions from the tile's smallest one: reads of one amplicon (batches sorted by
    // position) fall inside, and an event then finds its reads with one LDS read instead of a binary search
    uint16_t* wtab = reinterpret_cast<uint16_t*>(reinterpret_cast<int*>(owner + ent_cap) + NW * 192);
    // WIN (no bitmap): [ent_cap] read words, position-major | [WIN_TAB] read masks | [WIN_TAB] first word of the
    // position (16 bit) | per wave: [64] event records (16 B), net[64], H[64], bound[64], marker[64] | [NW] scan scratch.
    // Entry WIN_SIZE of the tables is a sentinel (mask 0) for positions outside the window (padding words).
    uint32_t* sval = lds + lds_word0;
    unsigned long long* tmask = reinterpret_cast<unsigned long long*>(sval + ent_cap);
    uint16_t* tstart = reinterpret_cast<uint16_t*>(tmask + WIN_TAB);
    uint4* wrec = reinterpret_cast<uint4*>(tstart + WIN_TAB) + wv * (WIN_WAVE_BYTES / 16);
    int* wacc = reinterpret_cast<int*>(wrec + 64);
    uint32_t* wmark = reinterpret_cast<uint32_t*>(wacc + 192);
    uint32_t* wscan = reinterpret_cast<uint32_t*>(reinterpret_cast<uint4*>(tstart + WIN_TAB) + NW * (WIN_WAVE_BYTES / 16));
    // the best score any wave of the workgroup has found for the read of lane l so far: the waves sweep different
    // chunks of the same stream for the same reads, and a bound found in one prunes the others
    int* wbest = reinterpret_cast<int*>(wscan + NW);
    const uint32_t win_lo = key_cap;

    const uint32_t tile = wg % ntiles;
    const uint32_t chunk = (wg / ntiles) * NW + wv;
    const uint32_t r0 = tile * T;                   // first list slot of the tile
    const uint32_t nr = min(T, n_list - r0);
    const bool have = lane < nr;
    const uint32_t rd = have ? list[r0 + lane] : 0;   // this lane's read
    STAT_WAIT(); STAT_T(8, t0_);
    const uint32_t so = have ? read_off[rd] : 0;
    const uint32_t my_k = have ? read_off[rd + 1] - so : 0;
    STAT_WAIT(); STAT_T(9, t0_);
    // everything the set-up needs from memory is requested here, in one go, so that the latencies
    // overlap: the first words of the read, the root's score, the checkpoint of the chunk start
    uint32_t pre[OWN_WORDS];
#pragma unroll
    for (uint32_t j = 0; j < OWN_WORDS; j++) pre[j] = (wv == 0 && my_k > j) ? read_word[so + j] : NONE;
    const int root_sc = have ? root_score[rd] : 0;
    const uint32_t b0 = chunk * blocks_per_chunk;
    const uint32_t b1 = min(m.NB, b0 + blocks_per_chunk);
    uint32_t cp_e0 = 0, cp_e1 = 0;
    if (b0 < m.NB) {
        const uint32_t cpi = b0 / m.cp_stride;
        cp_e0 = m.cp_off[cpi];
        cp_e1 = m.cp_off[cpi + 1];
    }
    const uint32_t cp_first = (cp_e0 + lane < cp_e1) ? m.cp_word[cp_e0 + lane] : 0;

    STAT_WAIT(); STAT_T(10, t0_);
    // exclusive prefix sum of the entry counts: where this lane's read sits in LDS
    uint32_t incl = my_k;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= (uint32_t)d) incl += o;
    }
    const uint32_t lds_off = incl - my_k;
    // plain variant: the wave's LDS operations complete in program order, so a compiler fence is
    // all the set-up needs; the dense variant's waves share the structures and take a barrier
    auto tile_sync = [&]() {
        if (DENSE) __syncthreads();
        else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    };
    const uint32_t tid = DENSE ? threadIdx.x : lane;

    if (!WIN)
        for (uint32_t i = tid; i < bm_words; i += 64 * NW) bitmap[i] = 0;
    // bm_words is a power of two >= (max_pos >> 5) + 1: positions beyond the tree's
    // last mutated site (and the padding word) alias into the map; a false positive
    // only costs a failed lookup in the reads.
    const uint32_t bm_mask = bm_words - 1;
    const uint32_t n_ent = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);   // read words of the tile
    uint32_t n2 = 1;                                   // bitonic network size (power of two >= n_ent)
    while (n2 < n_ent) n2 <<= 1;
    if (DENSE && !WIN) {
        for (uint32_t i = threadIdx.x; i < n2; i += 64 * NW) skey[i] = 0xFFFFFFFFu;
        for (uint32_t i = lane; i < 192; i += 64) acc[i] = 0;
    }
    if (WIN) {
        for (uint32_t i = threadIdx.x; i < WIN_TAB; i += 64 * NW) { tmask[i] = 0ull; tstart[i] = 0; }
        for (uint32_t i = lane; i < 192; i += 64) wacc[i] = 0;
        if (wv == 0) wbest[lane] = have ? root_sc + 1 : -(1 << 30);
    }
    STAT_T(11, t0_);
    tile_sync();
    STAT_T(5, t0_);
    const unsigned long long ts1_ = STAT_NOW();
    (void)ts1_;
    // WIN: a wave stages whole READS -- reads wv, wv + NW, ... of the tile, lane = word: one coalesced load per read
    // (each lane used to fetch its own read's words, NW apart: ~60 scattered dwords per read, every cache line asked for
    // by every wave) --, keeps the first 64 words of each in a register between the two passes (longer reads: read
    // again) and counts the read's c on the way (every wave used to read all the words of its lane's read once more
    // for c alone).  The counts travel through wave 0's accumulators (cleared again below).
    constexpr uint32_t WIN_OWN = (64 + NW - 1) / NW;     // reads per wave
    uint32_t wown[WIN ? WIN_OWN : 1];
    int c_win = 0;
    if (WIN) {
        uint32_t* tm32 = reinterpret_cast<uint32_t*>(tmask);
        int* cacc = reinterpret_cast<int*>(reinterpret_cast<uint4*>(tstart + WIN_TAB) + 64);      // wave 0's net[]
#pragma unroll
        for (uint32_t q = 0; q < WIN_OWN; q++) {
            const uint32_t r = wv + q * NW;                 // wave-uniform
            wown[q] = NONE;
            if (r < nr) {
                const uint32_t so_r = (uint32_t)__builtin_amdgcn_readlane((int)so, (int)r);
                const uint32_t k_r = (uint32_t)__builtin_amdgcn_readlane((int)my_k, (int)r);
                if (lane < k_r) wown[q] = read_word[so_r + lane];
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < WIN_OWN; q++) {
            const uint32_t r = wv + q * NW;
            if (r >= nr) continue;
            const uint32_t so_r = (uint32_t)__builtin_amdgcn_readlane((int)so, (int)r);
            const uint32_t k_r = (uint32_t)__builtin_amdgcn_readlane((int)my_k, (int)r);
            uint32_t cnt_r = 0;
            for (uint32_t j0 = 0; j0 < k_r; j0 += 64) {
                const bool on = j0 + lane < k_r;
                const uint32_t sw = j0 == 0 ? wown[q] : (on ? read_word[so_r + j0 + lane] : NONE);
                const uint32_t rel = w_pos(sw) - win_lo;
                if (on && rel < WIN_SIZE) atomicOr(&tm32[2 * rel + (r >> 5)], 1u << (r & 31));
                cnt_r += (uint32_t)__popcll(__ballot(on && !rw_missing(sw) && (rw_mut(sw) & rw_ref(sw)) == 0));
            }
            if (lane == 0) cacc[r] = (int)cnt_r;
        }
        __syncthreads();
        c_win = have ? cacc[lane] : 0;
        // tstart = exclusive prefix sum of the masks' populations (consecutive positions per thread)
        constexpr uint32_t PER = (WIN_SIZE + 64 * NW - 1) / (64 * NW);
        const uint32_t i0 = threadIdx.x * PER;
        uint32_t mine = 0;
        for (uint32_t i = i0; i < min(i0 + PER, WIN_SIZE); i++) mine += (uint32_t)__popcll(tmask[i]);
        uint32_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
            if (lane >= (uint32_t)d) inc += o;
        }
        if (lane == 63) wscan[wv] = inc;
        __syncthreads();
        if (wv == 0) wacc[lane] = 0;             // (every wave has read its reads' c by now)
        uint32_t run = inc - mine;
        for (uint32_t w2 = 0; w2 < wv; w2++) run += wscan[w2];
        for (uint32_t i = i0; i < min(i0 + PER, WIN_SIZE); i++) {
            tstart[i] = (uint16_t)run;
            run += (uint32_t)__popcll(tmask[i]);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < WIN_OWN; q++) {
            const uint32_t r = wv + q * NW;
            if (r >= nr) continue;
            const uint32_t so_r = (uint32_t)__builtin_amdgcn_readlane((int)so, (int)r);
            const uint32_t k_r = (uint32_t)__builtin_amdgcn_readlane((int)my_k, (int)r);
            const unsigned long long below_r = (1ull << r) - 1ull;
            for (uint32_t j0 = 0; j0 < k_r; j0 += 64) {
                const bool on = j0 + lane < k_r;
                const uint32_t w = j0 == 0 ? wown[q] : (on ? read_word[so_r + j0 + lane] : NONE);
                const uint32_t rel = w_pos(w) - win_lo;
                // (the copy keeps the allele fields; its position bits name the read's lane)
                if (on && rel < WIN_SIZE) sval[tstart[rel] + (uint32_t)__popcll(tmask[rel] & below_r)] = (w & 0xFFF00000u) | r;
            }
        }
    } else if (wv == 0) {
        auto stage = [&](uint32_t j, uint32_t w) {
            const uint32_t p = w_pos(w);
            if (S_IN_LDS) S_lds[lds_off + j] = w;
            if (DENSE) {
                skey[lds_off + j] = (p << 13) | (lds_off + j);
                owner[lds_off + j] = (uint8_t)lane;
            }
            if (p <= max_pos) atomicOr(&bitmap[(p >> 5) & bm_mask], 1u << (p & 31));
        };
#pragma unroll
        for (uint32_t j = 0; j < OWN_WORDS; j++)
            if (my_k > j) stage(j, pre[j]);
        for (uint32_t j = OWN_WORDS; j < my_k; j++) stage(j, read_word[so + j]);
    }
    tile_sync();
    // tile-wide position index: bitonic sort of the keys (once per tile, by the whole workgroup)
    if (DENSE && !WIN) {
        for (uint32_t k2 = 2; k2 <= n2; k2 <<= 1) {
            for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                for (uint32_t i = threadIdx.x; i < n2; i += 64 * NW) {
                    const uint32_t x = i ^ j2;
                    if (x > i) {
                        const uint32_t a = skey[i], b = skey[x];
                        if ((a > b) == ((i & k2) == 0)) { skey[i] = b; skey[x] = a; }
                    }
                }
                __syncthreads();
            }
        }
    }
    uint32_t plo = 0;
    if (DENSE && !WIN) {
        for (uint32_t i = threadIdx.x; i < DENSE_WINDOW; i += 64 * NW) wtab[i] = 0xFFFFu;
        __syncthreads();
        plo = n_ent ? (skey[0] >> 13) : 0u;
        for (uint32_t i = threadIdx.x; i < n_ent; i += 64 * NW) {
            const uint32_t p = skey[i] >> 13, rel = p - plo;
            if (rel < DENSE_WINDOW && (i == 0 || (skey[i - 1] >> 13) != p)) wtab[rel] = (uint16_t)i;
        }
        __syncthreads();
    }
    // from here on the waves of a workgroup never synchronise with each other again

    // Slice of this lane's read inside S (LDS copy or the global array).
    const uint32_t* S = S_IN_LDS ? (const uint32_t*)S_lds : read_word;
    const uint32_t my_off = S_IN_LDS ? lds_off : so;
    // position test.  `pos` is a position or a word whose low 20 bits are one: bm_words <= 2^15 (20-bit
    // positions), so the byte-offset mask also drops the bits above the position -- shift, and, LDS read,
    // bit-field extract (its offset operand uses the low five bits)
    // With one sweep per workgroup the bitmap is the first thing in the workgroup's LDS (these kernels
    // declare no static LDS; checked below): the byte offset then IS the LDS address, no base to add.
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    constexpr bool BM_AT_ZERO = SWEEP_WAVES == 1;
#ifdef WEPP_SWEEP_STATS   // (the check costs the scalar summary loads of the product build: stats build only)
    if (BM_AT_ZERO && (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)bitmap != 0u) __builtin_trap();
#endif
    const uint32_t bm_mask4 = bm_mask << 2;
    const char* bm_bytes = reinterpret_cast<const char*>(bitmap);
    auto bit = [&](uint32_t pos) -> bool {
#ifdef WEPP_EXP_NO_CONFLICT   // timing experiment only (wrong results): every lane reads its own bank
        const uint32_t off = (lane << 2) & bm_mask4;
#else
        const uint32_t off = (pos >> 3) & bm_mask4;
#endif
        const uint32_t word = BM_AT_ZERO ? *(lds_cu32*)(uintptr_t)off : *reinterpret_cast<const uint32_t*>(bm_bytes + off);
        return __builtin_amdgcn_ubfe(word, pos, 1u) != 0;
    };

    // the position of tree word `w` is listed by some read of the tile (WIN: its table entry, else the bitmap)
    auto win_mask = [&](uint32_t w) -> unsigned long long { return tmask[min(w_pos(w) - win_lo, WIN_SIZE)]; };
    auto hit = [&](uint32_t w) -> bool { return WIN ? win_mask(w) != 0ull : bit(w_pos(w)); };

    // OWN: the first OWN_WORDS words of this lane's read and their positions (an impossible
    // position where the read is shorter); tile_long = some read of the tile lists more
    uint32_t ow[OWN_WORDS], op[OWN_WORDS];
#pragma unroll
    for (uint32_t j = 0; j < OWN_WORDS; j++) { ow[j] = NONE; op[j] = NONE; }
    bool tile_long = false;
    if (OWN) {
#pragma unroll
        for (uint32_t j = 0; j < OWN_WORDS; j++)
            if (my_k > j) { ow[j] = pre[j]; op[j] = w_pos(ow[j]); }
        tile_long = __ballot(my_k > OWN_WORDS) != 0;
    }
    // WIN: the word of the read in lane `ln` at position P, or NONE
    auto win_entry = [&](uint32_t P, uint32_t ln) -> uint32_t {
        const uint32_t rel = P - win_lo;
        if (rel >= WIN_SIZE) return NONE;
        const unsigned long long mk = tmask[rel];
        if (!((mk >> ln) & 1ull)) return NONE;
        return sval[tstart[rel] + (uint32_t)__popcll(mk & ((1ull << ln) - 1ull))];
    };
    // this lane's read word at position P (wave-uniform), or NONE
    auto own_entry = [&](uint32_t P) -> uint32_t {
        if (WIN) return win_entry(P, lane);
        if (!OWN) return have ? find_entry(S, my_off, my_k, P) : NONE;
        uint32_t s = NONE;
#pragma unroll
        for (uint32_t j = 0; j < OWN_WORDS; j++) s = (op[j] == P) ? ow[j] : s;
        if (tile_long) {
            if (my_k > OWN_WORDS && s == NONE) s = find_entry(S, my_off + OWN_WORDS, my_k - OWN_WORDS, P);
        }
        return s;
    };

    // c for "no mutation anywhere on the path": every non-missing entry is
    // compared with its own reference allele (usher_mapper.cpp:302-305,342).
    int c = 0;
    if (WIN) c = c_win;                     // (counted while the words were staged; they sit position-major in LDS)
    else
        for (uint32_t j = 0; j < my_k; j++) {
            uint32_t s = S[my_off + j];
            if (!rw_missing(s)) c += ((rw_mut(s) & rw_ref(s)) == 0) ? 1 : 0;
        }

    STAT_T(6, ts1_);
    const unsigned long long ts2_ = STAT_NOW();
    (void)ts2_;
    // ---- state at the chunk start: enter words of every node still open there -
    {
        const uint32_t e0 = cp_e0, e1 = cp_e1;
        for (uint32_t e = e0; e < e1; e += 64) {
            const bool valid = e + lane < e1;
            const uint32_t w = e == e0 ? cp_first : (valid ? m.cp_word[e + lane] : 0);
            unsigned long long hm = __ballot(valid && hit(w));
            while (hm) {
                const int l = __builtin_ctzll(hm);
                hm &= hm - 1;
                const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)w, l);
                const uint32_t s = own_entry(w_pos(wl));
                if (s != NONE) c += enter_delta(wl, s);
            }
        }
    }

    // best score of this lane's read.  It starts one above the root's score: the root always
    // competes, so nothing worse can win or tie -- every chunk prunes against that bound from
    // its first block on (a chunk that finds nothing reports count 0 and loses in k_finalize).
    // Idle lanes of a partial tile hold INT_MIN: no block ever looks useful to them.
    int bs = have ? root_sc + 1 : -(1 << 30);
    uint32_t br = 0xFFFFFFFFu;  // its tie-break rank (smaller wins)
    uint32_t cnt = 0;           // eligible nodes attaining bs
    STAT_T(7, ts2_);
    STAT_T(0, t0_);

    // ---- node-by-node evaluation of one block for read r (lane = node) ----------
    // Everything it needs was fetched when the evaluation was decided (fetch_nodes):
    // w0/w1 + m0/m1 = this lane's two events and their node offsets, key/st = this
    // lane's node.
    // grp = the lanes whose read is word for word the read of lane r (they hold the same c, bs, br,
    // cnt at every point of the sweep): the evaluation is done once and its outcome taken by all
    // mm = this lane's two node offsets as loaded (byte 0: first event, byte 1: second)
    // WIN: mk0 / mk1 / ts0 / ts1 = the window table's entries of this lane's two events (read mask, first word), looked
    // up once per block: whether read r lists an event's position is a bit of its mask
    auto heavy_eval = [&](const BlkSum& sum, uint32_t e0, uint32_t e1, uint32_t w0, uint32_t w1, uint32_t mm,
                          int64_t key, uint32_t st, uint32_t wt, int r, unsigned long long grp,
                          unsigned long long mk0, unsigned long long mk1, uint32_t ts0, uint32_t ts1) {
        const uint32_t n0 = sum.node0;
        const uint32_t off_r = (uint32_t)__builtin_amdgcn_readlane((int)my_off, r);
        const uint32_t k_r = (uint32_t)__builtin_amdgcn_readlane((int)my_k, r);
        const int c_r = __builtin_amdgcn_readlane(c, r);
        const int bs_r = __builtin_amdgcn_readlane(bs, r);
        uint32_t r_ow[OWN_WORDS], r_op[OWN_WORDS];             // read r's first words (wave-uniform)
#pragma unroll
        for (uint32_t j = 0; j < OWN_WORDS; j++) {
            r_ow[j] = (uint32_t)__builtin_amdgcn_readlane((int)ow[j], r);
            r_op[j] = (uint32_t)__builtin_amdgcn_readlane((int)op[j], r);
        }
        int cadd = 0, adj = 0, dcom = 0;
        bool touched = false;
        auto apply = [&](uint32_t w, uint32_t mt, unsigned long long mk, uint32_t ts, bool tabled) {
            uint32_t sw;
            if (WIN && tabled) {
                sw = ((mk >> r) & 1ull) ? sval[ts + (uint32_t)__popcll(mk & ((1ull << r) - 1ull))] : NONE;
            } else if (OWN) {
                // read r's first two words come from its lane's registers (uniform), the rest from LDS
                const uint32_t p = w_pos(w);
                sw = NONE;
#pragma unroll
                for (uint32_t j = 0; j < OWN_WORDS; j++) sw = (p == r_op[j]) ? r_ow[j] : sw;
                if (k_r > OWN_WORDS) {
                    if (sw == NONE && bit(p)) sw = find_entry(S, off_r + OWN_WORDS, k_r - OWN_WORDS, p);
                }
            } else if (WIN) {
                sw = win_entry(w_pos(w), (uint32_t)r);
            } else {
                sw = bit(w_pos(w)) ? find_entry(S, off_r, k_r, w_pos(w)) : NONE;
            }
            unsigned long long hm = __ballot(sw != NONE);
            while (hm) {
                const int l = __builtin_ctzll(hm);
                hm &= hm - 1;
                const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)w, l);
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)mt, l) & EV_OFF_MASK_DEV;
                const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)sw, l);
                const int delta = enter_delta(wl, sl);
                if (wl & W_EXIT_DEV) {
                    // the subtree that carried wl ended just before node o
                    cadd += (lane >= o) ? -delta : 0;
                } else {
                    if (!(wl & W_LEAF_DEV)) {
                        // descendants of node o see the new allele; the root also scores
                        // itself with its own mutations applied (usher_mapper.cpp:266-271)
                        const bool is_root = (n0 + o) == 0;
                        cadd += (lane > o || (is_root && lane == o)) ? delta : 0;
                    }
                    int a1 = 0, a2 = 0;
                    own_adjust(wl, sl, a1, a2);            // uniform
                    if (lane == o) {
                        touched = true;
                        adj += a1;
                        dcom += a2;
                    }
                }
            }
        };
        apply(w0, mm, mk0, ts0, true);
        apply(w1, mm >> 8, mk1, ts1, true);
        for (uint32_t e = e0 + 128; e < e1; e += 64) {      // rare: a block with more than 128 events
            const bool valid = e + lane < e1;
            apply(valid ? m.ev_word[e + lane] : W_PAD_DEV, valid ? (uint32_t)m.ev_meta[e + lane] : 0u, 0ull, 0u, false);
        }
        const int base = (int)(key >> 32);
        const uint32_t rank = (uint32_t)(key & 0xFFFFFFFFll);
        const uint32_t nmut = st & NS_CNT_MASK_DEV;
        const uint32_t ncom0 = (st >> 14) & NS_CNT_MASK_DEV;
        const bool leaf = st & NS_LEAF_DEV, masked = st & NS_MASKED_DEV, root = st & NS_ROOT_DEV;
        bool elig;
        int score = base + c_r + cadd;
        if (root) elig = true;
        else if (masked) elig = false;
        else if (touched) {
            score += adj;
            const int ncom = (int)ncom0 + dcom;
            elig = leaf ? (ncom > 0) : (ncom > 0 || ncom == (int)nmut);     // usher_mapper.cpp:455-456
        } else elig = st & NS_ELIG0_DEV;
        elig = elig && (lane < sum.nn);
        STAT_ADD(4, 1);
        if (__ballot(elig && score <= bs_r)) {
            STAT_ADD(5, 1);
            const int smin = wave_min_i32(elig ? score : 0x7FFFFFFF);
            const bool at_min = elig && score == smin;
            // (window streams: an element stands for several nodes)
            const uint32_t cntb = m.ncnt ? wave_sum_u32(at_min ? wt : 0u) : (uint32_t)__popcll(__ballot(at_min));
            const uint32_t rmin = wave_min_u32(at_min ? rank : 0xFFFFFFFFu);
            if ((grp >> lane) & 1ull) {
                if (smin < bs) { bs = smin; br = rmin; cnt = cntb; }
                else if (smin == bs) { cnt += cntb; br = min(br, rmin); }
            }
        }
    };

    // ---- one block: lane = read ----------------------------------------------------
    // w0/w1 = this lane's two words of the block's first 128 events (W_PAD beyond e1)
    // lbw = this lane's two per-event bounds (crown streams; fetched with the event words)
    auto process_block = [&](uint32_t e0, uint32_t e1, uint32_t w0, uint32_t w1, uint32_t lbw, const BlkSum sum) {
        const unsigned long long tb_ = STAT_NOW();
        (void)tb_;
        unsigned long long mk0 = 0, mk1 = 0;      // WIN: the tile's reads that list the positions of this lane's events
        if (WIN) { mk0 = win_mask(w0); mk1 = win_mask(w1); }
        // WIN: what another wave of the workgroup has found for this lane's read bounds this chunk too.  A chunk that
        // takes the bound over holds no node of that score yet (count 0: k_finalize ignores it unless it finds one)
        int bs_in = bs;
        if (WIN) {
            const int g = wbest[lane];
            if (g < bs) { bs = g; br = 0xFFFFFFFFu; cnt = 0; }
            bs_in = bs;
        }
        auto publish = [&]() {
            if (WIN && bs < bs_in) atomicMin(&wbest[lane], bs);
        };
        const unsigned long long hm0 = WIN ? __ballot(mk0 != 0ull) : __ballot(bit(w0));
        const unsigned long long hm1 = WIN ? __ballot(mk1 != 0ull) : __ballot(bit(w1));
        STAT_ADD(0, 1);
        STAT_ADD(1, (hm0 | hm1) ? 1 : 0);
        STAT_ADD(2, __popcll(hm0) + __popcll(hm1));
        // reads without an event in this block: one summary update (a block without statically
        // eligible nodes has base = SCORE_INF and never passes the test)
        auto summary_update = [&](bool untouched) {
            const bool take = untouched && sum.base + c <= bs;
            STAT_ADD(6, __ballot(take) ? 1 : 0);
            if (__ballot(take)) {              // rare once a good node has been seen: skipped wave-wide
                __builtin_amdgcn_sched_barrier(0);   // keeps the update behind a real branch (no if-conversion)
                if (take) {
                    const int s = sum.base + c;
                    if (s < bs) { bs = s; br = sum.rank; cnt = sum.cnt; }
                    else { cnt += sum.cnt; br = min(br, sum.rank); }
                }
            }
        };
#ifdef WEPP_EXP_NO_HITS     // timing experiment only (wrong results): every block takes the no-hit path
        const bool any_hit = false;
#else
        // wave-uniform; false for most blocks.  One 64-bit OR and one compare: hits of either half, or bit 0
        // from the sign of 128 - (events of the block) (a block holds fewer than 2^31 events)
        const bool any_hit = (hm0 | hm1 | (unsigned long long)((128u - (e1 - e0)) >> 31)) != 0;
#endif
        if (!any_hit) {
            summary_update(true);
            publish();
            STAT_T(1, tb_);
            return;
        }
        __builtin_amdgcn_sched_barrier(0);     // the hit path stays out of line of the fast path
        int net = 0, H = 0, lbmin = 0x3FFFFFFF;
        bool touched = false;
        // for a hit event, the reads that list its position take its delta; lbl = lower
        // bound of the static score of the nodes this event can affect in the block
        auto light_hit = [&](uint32_t wl, int lbl) {
            const uint32_t s = own_entry(w_pos(wl));
            STAT_ADD(3, __popcll(__ballot(s != NONE)));
            if (s != NONE) {
                const int d = enter_delta(wl, s);
                touched = true;
                lbmin = min(lbmin, lbl);
                // H bounds how far the events can LOWER a score: an exit takes d away from the nodes
                // behind it (matters if d > 0); an enter gives d to the descendants (matters if d < 0) and
                // changes the node's own score, which takes no d, by at least -1: max(-d, 1) covers both
                if (wl & W_EXIT_DEV) { net -= d; H += max(d, 0); }
                else if (wl & W_LEAF_DEV) {
                    // a leaf has no descendants: only its own adjustment can lower its score, and it does
                    // (by one) only if the read shares the new allele and not the parent's (own_adjust:
                    // actual_sub); an N or another allele leaves the leaf's score where it was
                    const uint32_t a = rw_mut(s), par = tw_par(wl);
                    const bool lowers = !rw_missing(s) && (a & tw_mut(wl)) != 0 && (a & (par ? par : rw_ref(s))) == 0;
                    H += lowers ? 1 : 0;
                }
                else { net += d; H += max(-d, 1); }
            }
        };
        uint32_t mm = 0, st = 0, wt = 1;
        int64_t key = 0;
        bool fetched = false;
        auto fetch_nodes = [&]() {
            if (e0 + 2 * lane < e1) {
                mm = *reinterpret_cast<const uint16_t*>(m.ev_meta + e0 + 2 * lane);
            }
            if (lane < sum.nn) {
                key = m.nkey[sum.node0 + lane];
                st = m.nstat[sum.node0 + lane];
                if (m.ncnt) wt = m.ncnt[sum.node0 + lane];
            }
            fetched = true;
        };
        // (window tiles of long reads evaluate some node of nearly every block with a hit -- 8.7 reads per block on a
        // window's candidate crown --: the node data is requested here, and arrives while the pairs are resolved)
        if (WIN) fetch_nodes();
        uint32_t ts0 = 0, ts1 = 0;               // WIN: first word of this lane's events' positions in the position-major copy
        if (WIN) { ts0 = tstart[min(w_pos(w0) - win_lo, WIN_SIZE)]; ts1 = tstart[min(w_pos(w1) - win_lo, WIN_SIZE)]; }
        {
            // per-event bounds of this lane's two events.  Crown streams interleave low- and
            // high-score nodes, so the per-event bound (one more 2-byte load) is what prunes there;
            // on the whole-tree stream the block minimum already prunes ~95 % and costs no load.
            uint32_t lb0 = (uint32_t)max(sum.min_all, 0), lb1 = lb0;
            if (m.eager) {
                lb0 = lbw & 0xFFu;
                lb1 = lbw >> 8;
            }
            // crown streams hold only low-score nodes, so a hit nearly always ends in the
            // node-by-node path: start its loads before the lookups.  On the whole-tree
            // stream the bound prunes ~95 % of the hits and the loads are issued on demand.
            // (the node data is fetched when an evaluation is decided: fetched at the first hit, its loads sat
            // between the prefetched event words of the next blocks and every hit block waited for them)
            unsigned long long hm;
            if (DENSE && !WIN && __popcll(hm0) + __popcll(hm1) >= DENSE_MIN_HITS) {
                // many hit events (long reads): lane = event.  Every lane looks its event up in
                // the tile-sorted key array and adds its contribution to the owning read's
                // accumulators in LDS; all events of the block are resolved together.
                // both events of the lane are searched in one loop: two independent chains of
                // dependent LDS reads in flight instead of one
                const uint32_t p0 = w_pos(w0), p1 = w_pos(w1);
                const uint32_t want0 = p0 << 13, want1 = p1 << 13;
                const bool act0 = (hm0 >> lane) & 1ull, act1 = (hm1 >> lane) & 1ull;
                const uint32_t rel0 = p0 - plo, rel1 = p1 - plo;
                // inside the window: direct index (0xFFFF = no read lists the position: fails `i < n2` below)
                uint32_t lo0 = (act0 && rel0 < DENSE_WINDOW) ? wtab[rel0] : 0u;
                uint32_t lo1 = (act1 && rel1 < DENSE_WINDOW) ? wtab[rel1] : 0u;
                if (__ballot((act0 && rel0 >= DENSE_WINDOW) || (act1 && rel1 >= DENSE_WINDOW))) {
                    uint32_t b0s = 0, b1s = 0;
                    for (uint32_t step = n2 >> 1; step > 0; step >>= 1) {      // lower_bound, n2 is a power of two
                        const uint32_t k0 = skey[b0s + step - 1], k1 = skey[b1s + step - 1];
                        if (k0 < want0) b0s += step;
                        if (k1 < want1) b1s += step;
                    }
                    if (rel0 >= DENSE_WINDOW) lo0 = b0s;
                    if (rel1 >= DENSE_WINDOW) lo1 = b1s;
                }
                auto dense_apply = [&](uint32_t w, uint32_t p, uint32_t lb, uint32_t i, bool act) {
                    while (__ballot(act && i < n2 && (skey[min(i, n2 - 1)] >> 13) == p)) {
                        const uint32_t kv = skey[min(i, n2 - 1)];
                        if (act && i < n2 && (kv >> 13) == p) {
                            const uint32_t idx = kv & 8191u;
                            const int d = enter_delta(w, S_lds[idx]);
                            int dn, dh;
                            if (w & W_EXIT_DEV) { dn = -d; dh = max(d, 0); }
                            else if (w & W_LEAF_DEV) {
                                const uint32_t sl = S_lds[idx], a = rw_mut(sl), par = tw_par(w);
                                dn = 0;
                                dh = (!rw_missing(sl) && (a & tw_mut(w)) != 0 && (a & (par ? par : rw_ref(sl))) == 0) ? 1 : 0;
                            }
                            else { dn = d; dh = max(-d, 1); }
                            const uint32_t o = owner[idx];
                            if (dn) atomicAdd(&acc[o], dn);
                            if (dh) atomicAdd(&acc[64 + o], dh);
                            // touched marker + running min of the events' bounds (max of 2^30 - lb)
                            atomicMax(&acc[128 + o], 0x40000000 - (int)lb);
                        }
                        i++;
                    }
                };
                dense_apply(w0, p0, lb0, lo0, act0);
                dense_apply(w1, p1, lb1, lo1, act1);
                // the accumulators belong to this wave alone; its LDS operations complete in
                // program order, the barriers only stop the compiler from reordering them
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (acc[128 + lane]) {
                    touched = true;
                    lbmin = 0x40000000 - acc[128 + lane];
                    net = acc[lane];
                    H = acc[64 + lane];
                    acc[lane] = 0;
                    acc[64 + lane] = 0;
                    acc[128 + lane] = 0;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else if (WIN) {
                // lane = (event, read) pair.  Every lane counts the reads of its two events, the counts are
                // scanned; pair i belongs to the last event whose first pair is <= i (markers + max scan).
                const uint32_t n0 = (uint32_t)__popcll(mk0), n1 = (uint32_t)__popcll(mk1), nn = n0 + n1;
                const uint32_t inc = wave_scan_add_u32(nn), exc = inc - nn;
                const uint32_t npairs = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                STAT_ADD(3, npairs);
                // (an event record keeps the word's allele / flag bits; its position bits carry the event's bound)
                wrec[lane] = make_uint4((w0 & 0xFFF00000u) | min(lb0, 255u), (w1 & 0xFFF00000u) | min(lb1, 255u), exc | (n0 << 16),
                                        ts0 | (ts1 << 16));
                uint32_t carry = 0;
                for (uint32_t pb = 0; pb < npairs; pb += 64) {
                    wmark[lane] = 0;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (nn && exc - pb < 64u) wmark[exc - pb] = lane + 1;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t ev = max(wave_scan_max_u32(wmark[lane]), carry);     // event lane + 1 of this pair
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)ev, 63);
                    const uint32_t i = pb + lane;
                    if (i < npairs) {
                        const uint4 rc = wrec[ev - 1];
                        uint32_t k = i - (rc.z & 0xFFFFu);
                        const uint32_t n0e = rc.z >> 16;
                        const bool second = k >= n0e;
                        if (second) k -= n0e;
                        const uint32_t w = second ? rc.y : rc.x;
                        const uint32_t sl = sval[(second ? rc.w >> 16 : rc.w & 0xFFFFu) + k];
                        const int d = enter_delta(w, sl);
                        int dn, dh;
                        if (w & W_EXIT_DEV) { dn = -d; dh = max(d, 0); }
                        else if (w & W_LEAF_DEV) {
                            const uint32_t a = rw_mut(sl), par = tw_par(w);
                            dn = 0;
                            dh = (!rw_missing(sl) && (a & tw_mut(w)) != 0 && (a & (par ? par : rw_ref(sl))) == 0) ? 1 : 0;
                        }
                        else { dn = d; dh = max(-d, 1); }
                        const uint32_t o = sl & 63u;
                        if (dn) atomicAdd(&wacc[o], dn);
                        atomicAdd(&wacc[64 + o], dh + 0x10000);      // high half: matches of the read (touched marker)
                        atomicMax(&wacc[128 + o], 0x40000000 - (int)(w & 0xFFu));   // running min of the events' bounds
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                const int hcnt = wacc[64 + lane];
                if (hcnt) {
                    touched = true;
                    lbmin = 0x40000000 - wacc[128 + lane];
                    net = wacc[lane];
                    H = hcnt & 0xFFFF;
                    wacc[lane] = 0;
                    wacc[64 + lane] = 0;
                    wacc[128 + lane] = 0;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                hm = hm0;
                while (hm) {
                    const int l = __builtin_ctzll(hm);
                    hm &= hm - 1;
                    light_hit((uint32_t)__builtin_amdgcn_readlane((int)w0, l), __builtin_amdgcn_readlane((int)lb0, l));
                }
                hm = hm1;
                while (hm) {
                    const int l = __builtin_ctzll(hm);
                    hm &= hm - 1;
                    light_hit((uint32_t)__builtin_amdgcn_readlane((int)w1, l), __builtin_amdgcn_readlane((int)lb1, l));
                }
            }
            for (uint32_t e = e0 + 128; e < e1; e += 64) {      // rare: a block with more than 128 events
                const uint32_t w = (e + lane < e1) ? m.ev_word[e + lane] : W_PAD_DEV;
                hm = __ballot(hit(w));
                while (hm) {
                    const int l = __builtin_ctzll(hm);
                    hm &= hm - 1;
                    light_hit((uint32_t)__builtin_amdgcn_readlane((int)w, l), min(sum.min_all, 0));   // overflow events: block bound
                }
            }
        }
        summary_update(!touched);
        // reads with events: a node changed by the events scores at least
        // (min of the events' bounds) + c - H (|delta| per event bounds c, -1 per enter bounds
        // the node's own adjustment), a node they leave alone at least base + c.  Unless one
        // of the two can reach the current best, only c moves on; otherwise evaluate the
        // block node by node
        {
            const bool heavy = touched && ((lbmin + c - H <= bs) || (sum.base + c <= bs));
            unsigned long long hv = __ballot(heavy);
#ifdef WEPP_EXP_NO_HEAVY   // timing experiment only (wrong results): no node-by-node evaluation
            hv = 0;
#endif
            const unsigned long long th_ = STAT_NOW();
            (void)th_;
            STAT_T(hv ? 3 : 2, tb_);
            while (hv) {
                const int r = __builtin_ctzll(hv);
                // reads of the tile identical to read r (sorted batches put them side by side)
                unsigned long long grp = 1ull << r;
                if (OWN) {
                    const uint32_t k_r = (uint32_t)__builtin_amdgcn_readlane((int)my_k, r);
                    if (k_r <= OWN_WORDS) {
                        bool same = my_k == k_r;
#pragma unroll
                        for (uint32_t j = 0; j < OWN_WORDS; j++)
                            same = same && ow[j] == (uint32_t)__builtin_amdgcn_readlane((int)ow[j], r);
                        grp = __ballot(same) & hv;
                    }
                }
                hv &= ~grp;
                if (!fetched) fetch_nodes();
                heavy_eval(sum, e0, e1, w0, w1, mm, key, st, wt, r, grp, mk0, mk1, ts0, ts1);
            }
            STAT_T(4, th_);
            c += net;
        }
        publish();
    };

    // ---- the sweep: groups of 60 blocks (their 61 event offsets sit in one
    // vector register), four blocks' loads issued together ----------------------
    for (uint32_t bb = b0; bb < b1; bb += 60) {
        const uint32_t ng = min(60u, b1 - bb);
        const uint32_t eo_vec = (lane <= ng) ? m.blk_eoff[bb + lane] : 0;
        for (uint32_t j = 0; j < ng; j += 4) {
            uint32_t e[5];
#pragma unroll
            for (int q = 0; q < 5; q++) e[q] = (uint32_t)__builtin_amdgcn_readlane((int)eo_vec, (int)min(j + q, ng));
            uint2 ww[4];
            uint32_t lbw[4];
            BlkSum sm[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // Every lane loads:
                // with loads that may or may not be issued the compiler cannot count the outstanding ones
                // and waits for all four blocks before the first.  Crown streams: a hit nearly always
                // needs the per-event bounds; fetching them here keeps a dependent load off the hit path
                // (the whole-tree stream ignores them).
                // (the device copies of ev_word / ev_lb / blk_sum carry EV_TAIL_PAD padding events and
                // SUM_TAIL_PAD summaries behind the last one: no index needs a clamp)
                // lanes past the block's events load padding words (the first two of the tail padding)
                const uint32_t idr = e[q] + 2 * lane;
                const uint32_t idx = idr < e[q + 1] ? idr : m.e_pad;
                ww[q] = *reinterpret_cast<const uint2*>(m.ev_word + idx);
                lbw[q] = *reinterpret_cast<const uint16_t*>(m.ev_lb + idx);
                sm[q] = m.blk_sum[bb + j + q];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (j + q >= ng) continue;
                process_block(e[q], e[q + 1], ww[q].x, ww[q].y, lbw[q], sm[q]);
            }
        }
    }

    STAT_FLUSH(m.tier);
    if (have) {
        const size_t o = (size_t)chunk * n_list + r0 + lane;
        part_score[o] = bs;
        part_rank[o] = br;
        part_cnt[o] = cnt;
    }
}

// one stream per launch (dense variant, reads too long for LDS)
template <bool S_IN_LDS, bool DENSE, bool WIN = false>
__global__ __launch_bounds__(DENSE ? 64 * DENSE_WAVES : 64) void k_sweep(
    DevStream m, uint32_t bm_words, uint32_t max_pos, uint32_t ent_cap, uint32_t key_cap,
    const uint32_t* __restrict__ read_off, const uint32_t* __restrict__ read_word,
    const int32_t* __restrict__ root_score, const uint32_t* __restrict__ list, uint32_t n_list, uint32_t T,
    uint32_t ntiles, uint32_t blocks_per_chunk, int32_t* __restrict__ part_score, uint32_t* __restrict__ part_rank,
    uint32_t* __restrict__ part_cnt) {
    sweep_tile<S_IN_LDS, DENSE, WIN>(m, blockIdx.x, 0u, bm_words, max_pos, ent_cap, key_cap, read_off, read_word, root_score,
                                     list, n_list, T, ntiles, blocks_per_chunk, part_score, part_rank, part_cnt);
}

// the reads that cannot walk (more than WALK16_K entries, or too many open intervals) but lie inside one genome
// window: waves of its own (ARENA_CHUNKS per read) sweep the read's WINDOW CROWN (wsid: a few hundred to a few ten
// thousand nodes) instead of a 64-read tile sweeping the tree-wide stream of theta = root score + |S|
__global__ __launch_bounds__(64) void k_sweep_arena(const DevStream* __restrict__ wc_streams, const uint32_t* __restrict__ wsid,
                                                    uint32_t bm_words, uint32_t max_pos, uint32_t ent_cap,
                                                    const uint32_t* __restrict__ read_off, const uint32_t* __restrict__ read_word,
                                                    const int32_t* __restrict__ root_score, const uint32_t* __restrict__ list,
                                                    uint32_t n_list, int32_t* __restrict__ part_score, uint32_t* __restrict__ part_rank,
                                                    uint32_t* __restrict__ part_cnt) {
    // a wave per (read, chunk of its crown): ARENA_CHUNKS chunks cut at the crown's checkpoints, chunk-major like
    // every sweep launch; a crown with fewer checkpoints leaves its last chunks empty (a partial that counts nothing).
    // (One wave per read took 0.7 ms for a read on a 100 K-node crown: the tail of a 0.3 ms step.)
    const uint32_t i = blockIdx.x % n_list, chunk = blockIdx.x / n_list;
    const uint32_t sid = (uint32_t)__builtin_amdgcn_readfirstlane((int)wsid[list[i]]);
    const DevStream st = wc_streams[sid];
    const uint32_t bpc = ((st.ncp + ARENA_CHUNKS - 1) / ARENA_CHUNKS) * st.cp_stride;
    if (chunk * bpc >= st.NB) {
        if (threadIdx.x == 0) {
            const size_t o = (size_t)chunk * n_list + i;
            part_score[o] = SCORE_INF_DEV;
            part_rank[o] = 0xFFFFFFFFu;
            part_cnt[o] = 0;
        }
        return;
    }
    sweep_tile<true, false>(st, blockIdx.x, 0u, bm_words, max_pos, ent_cap, 0u, read_off, read_word, root_score, list, n_list, 1u, n_list,
                            bpc, part_score, part_rank, part_cnt);
}

// all the plain (short-read) plans of one placement call in ONE launch: the sweeps of the
// different streams run side by side instead of queueing behind the hardware queues.  A
// workgroup is SWEEP_WAVES independent waves, each with its own (tile, chunk) and LDS region.
__global__ __launch_bounds__(64 * SWEEP_WAVES) void k_sweep_multi(SweepPlans pl, uint32_t bm_words, uint32_t max_pos,
                                                                  uint32_t lds_words_per_wave,
                                                                  const uint32_t* __restrict__ read_off,
                                                                  const uint32_t* __restrict__ read_word,
                                                                  const int32_t* __restrict__ root_score) {
    const uint32_t wv = SWEEP_WAVES == 1 ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t unit = blockIdx.x * SWEEP_WAVES + wv;
    if (unit >= pl.p[pl.n - 1].wg_end) return;
    uint32_t p = 0;
    while (p + 1 < pl.n && unit >= pl.p[p].wg_end) p++;
    const SweepPlanDev& q = pl.p[p];
    const uint32_t wg0 = p ? pl.p[p - 1].wg_end : 0;
    sweep_tile<true, false>(q.st, unit - wg0, wv * lds_words_per_wave, bm_words, max_pos, q.ent_cap, 0u, read_off,
                                  read_word, root_score, q.list, q.n_list, q.T, q.ntiles, q.bpc, q.part_score,
                                  q.part_rank, q.part_cnt);
}

// -----------------------------------------------------------------------------
// finalize: combine the chunks of a read, map the winner back to the
// reference's BFS index and recompute its has_unique flag
// (usher_mapper.cpp:184,199,262,472,492).
// -----------------------------------------------------------------------------
// the per-read outputs from a read's best (score, rank, count): the reference's BFS index of the winner and
// its has_unique flag recomputed from its own mutations (usher_mapper.cpp:184,199,262,472,492)
__device__ __forceinline__ void emit_result(const DevMAT& m, uint32_t r, const uint32_t* __restrict__ read_off,
                                            const uint32_t* __restrict__ read_word, int bs, uint32_t br, uint32_t cnt,
                                            uint32_t* __restrict__ best_bfs_j, int32_t* __restrict__ score,
                                            uint32_t* __restrict__ num_best, uint32_t* __restrict__ flags) {
    // the root always competes, so some chunk reports it or better; the clamp only keeps a broken
    // invariant from becoming an out-of-bounds read
    const uint32_t d = m.rank2dfs[br < m.N ? br : 0u];
    const uint32_t st = m.nstat[d];
    uint32_t hu = 0;
    if (!(st & NS_ROOT_DEV)) {
        if (st & NS_MASKED_DEV) hu = 1;
        else {
            int ncom = (int)((st >> 14) & NS_CNT_MASK_DEV);
            int dummy = 0;
            const uint32_t so = read_off[r], k = read_off[r + 1] - so;
            for (uint32_t w = m.node_woff[d]; w < m.node_woff[d + 1]; w++) {
                const uint32_t tw = m.words[w];
                const uint32_t s = find_entry(read_word, so, k, w_pos(tw));
                if (s != NONE) own_adjust(tw, s, dummy, ncom);
            }
            hu = (ncom < (int)(st & NS_CNT_MASK_DEV)) ? 1u : 0u;
        }
    }
    if (best_bfs_j) best_bfs_j[r] = m.dfs2bfs[d];
    if (score) score[r] = bs;
    if (num_best) num_best[r] = cnt;
    if (flags) flags[r] = hu ? WEPP_FLAG_HAS_UNIQUE_DEV : 0u;
}

template <uint32_t LPR>
__device__ __forceinline__ void finalize_reads(const DevMAT& m, uint32_t blk, const uint32_t* __restrict__ read_off,
                           const uint32_t* __restrict__ read_word, const uint32_t* __restrict__ list,
                           uint32_t n_list, uint32_t nchunks, const int32_t* __restrict__ part_score,
                           const uint32_t* __restrict__ part_rank, const uint32_t* __restrict__ part_cnt,
                           uint32_t* __restrict__ best_bfs_j, int32_t* __restrict__ score,
                           uint32_t* __restrict__ num_best, uint32_t* __restrict__ flags) {
    // LPR lanes per read (finalize_lanes_per_read: 1, 4, 16 or 64 by the number of chunks): a wave holds
    // 64 / LPR consecutive list entries, lane = sub * (64 / LPR) + entry, so that every load of a partial
    // covers consecutive entries of LPR chunk rows; the LPR lanes of an entry stride over its chunks and
    // are combined with xor shuffles
    constexpr uint32_t RPW = 64 / LPR;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t sub = lane / RPW;
    const uint32_t i = ((blk * blockDim.x + threadIdx.x) >> 6) * RPW + lane % RPW;
    const bool valid = i < n_list;
    if (LPR == 1 && !valid) return;                          // (no shuffles: a lane may leave)
    int bs = 0x7FFFFFFF;
    uint32_t br = 0xFFFFFFFFu, cnt = 0;
    // the three values of a chunk are loaded together, and two chunks per round: a thread's chunks used to
    // cost it up to three memory round trips each, one after the other
    auto take = [&](int s, uint32_t pr, uint32_t pc) {
        if (pc == 0) return;                                 // this chunk found nothing within the bound
        if (s < bs) { bs = s; br = pr; cnt = pc; }
        else if (s == bs) { cnt += pc; br = min(br, pr); }
    };
    if (valid) {
        uint32_t ch = sub;
        for (; ch + LPR < nchunks; ch += 2 * LPR) {
            const size_t o0 = (size_t)ch * n_list + i, o1 = (size_t)(ch + LPR) * n_list + i;
            const int s0 = part_score[o0], s1 = part_score[o1];
            const uint32_t c0 = part_cnt[o0], c1 = part_cnt[o1];
            const uint32_t r0 = part_rank[o0], r1 = part_rank[o1];
            take(s0, r0, c0);
            take(s1, r1, c1);
        }
        if (ch < nchunks) {
            const size_t o = (size_t)ch * n_list + i;
            const int s = part_score[o];
            const uint32_t pc = part_cnt[o], pr = part_rank[o];
            take(s, pr, pc);
        }
    }
#pragma unroll
    for (uint32_t msk = RPW; msk < 64; msk <<= 1) {
        const int os = __shfl_xor(bs, (int)msk, 64);
        const uint32_t orr = (uint32_t)__shfl_xor((int)br, (int)msk, 64);
        const uint32_t oc = (uint32_t)__shfl_xor((int)cnt, (int)msk, 64);
        if (oc) {
            if (os < bs) { bs = os; br = orr; cnt = oc; }
            else if (os == bs) { cnt += oc; br = min(br, orr); }
        }
    }
    if (sub != 0 || !valid) return;
    emit_result(m, list[i], read_off, read_word, bs, br, cnt, best_bfs_j, score, num_best, flags);
}

template <uint32_t LPR>
__global__ void k_finalize(DevMAT m, const uint32_t* __restrict__ read_off,
                           const uint32_t* __restrict__ read_word, const uint32_t* __restrict__ list,
                           uint32_t n_list, uint32_t nchunks, const int32_t* __restrict__ part_score,
                           const uint32_t* __restrict__ part_rank, const uint32_t* __restrict__ part_cnt,
                           uint32_t* __restrict__ best_bfs_j, int32_t* __restrict__ score,
                           uint32_t* __restrict__ num_best, uint32_t* __restrict__ flags) {
    finalize_reads<LPR>(m, blockIdx.x, read_off, read_word, list, n_list, nchunks, part_score, part_rank,
                                  part_cnt, best_bfs_j, score, num_best, flags);
}

// the finalizes of all fused plans in one launch (fin_end = first block after a plan)
__global__ void k_finalize_multi(DevMAT m, SweepPlans pl, const uint32_t* __restrict__ read_off,
                                 const uint32_t* __restrict__ read_word, uint32_t* __restrict__ best_bfs_j,
                                 int32_t* __restrict__ score, uint32_t* __restrict__ num_best,
                                 uint32_t* __restrict__ flags) {
    uint32_t p = 0;
    while (p + 1 < pl.n && blockIdx.x >= pl.p[p].fin_end) p++;
    const SweepPlanDev& q = pl.p[p];
    const uint32_t blk = blockIdx.x - (p ? pl.p[p - 1].fin_end : 0);
#define FIN(L) finalize_reads<L>(m, blk, read_off, read_word, q.list, q.n_list, q.nchunks, q.part_score, q.part_rank, \
                                 q.part_cnt, best_bfs_j, score, num_best, flags)
    switch (finalize_lanes_per_read(q.nchunks)) {
        case 1: FIN(1); break;
        case 4: FIN(4); break;
        case 16: FIN(16); break;
        default: FIN(64); break;
    }
#undef FIN
}

// -----------------------------------------------------------------------------
// k_walk: a read visits only the events of the positions IT lists.
//
// lane = read.  The stream's position index (DevWalk) gives, per listed position, the mutations of
// the stream's nodes at that position in stream order, each with the node's index and the end of its
// subtree: the read merges its (at most KW) lists, keeps the intervals it has entered on a stack
// (they are nested: subtrees), and between two consecutive events -- where its running c_S is
// constant and no node carries one of its positions -- asks a range query for the best statically
// eligible node: a sparse table of the minimum static score says whether anything in the range can
// reach the read's best (almost never), and only then a segment tree gives the exact (score, rank,
// count).  The node of an event is evaluated with the formula of the sweep's node-by-node path.
// Work per read ~ events at its positions in the stream, instead of the whole stream per tile:
// 2 events instead of 270 blocks on the 17 K-node crown, ~3.5 K instead of 250 K blocks on the
// whole tree for a read with three entries.  Same results (tests/walk_model.py is the CPU model).
// -----------------------------------------------------------------------------
template <int KW, int SD, bool CHUNKED>
__global__ __launch_bounds__(64 * WALK_WAVES) void k_walk(DevMAT m, WalkPlans pl, WalkJobs jb, uint32_t sd_rows,
                                              const uint32_t* __restrict__ read_off,
                                              const uint32_t* __restrict__ read_word,
                                              const int32_t* __restrict__ root_score, uint32_t* __restrict__ best_bfs_j,
                                              int32_t* __restrict__ score_out, uint32_t* __restrict__ num_best,
                                              uint32_t* __restrict__ flags, unsigned long long* __restrict__ work_counter,
                                              const uint32_t* __restrict__ wsid) {
    // wave-private LDS: the allele fields of the read words, 16 bits each [KW / 2][64]; the list cursors [KW][64];
    // the interval stack [sd_rows][64] -- sd_rows = the deepest stack a read of this launch can need (k_route's
    // maximum of open_max over the class, <= SD).  The walk waits on memory: what it gains from a wave more
    // per SIMD is nearly proportional, and its LDS request is what limits them.
    extern __shared__ uint32_t lds_all[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* S16 = lds_all + wv * (KW / 2 + KW + sd_rows) * 64;
    uint32_t* cur_l = S16 + (KW / 2) * 64;
    uint32_t* stk = cur_l + KW * 64;
    // the read word of list j rebuilt from its 9 allele bits (the position is not needed again)
    auto sword = [&](int j) -> uint32_t { return ((S16[(j >> 1) * 64 + lane] >> ((j & 1) * 16)) & 0x1FFu) << 20; };
    const uint32_t unit = blockIdx.x * WALK_WAVES + wv;
    if (unit >= pl.p[pl.n - 1].wave_end) return;
#ifdef WEPP_WALK_STATS   // (profiling build: wave cycles by phase into the work counters, tools/walk_probe.py prints them)
    unsigned long long ts_[6];
    ts_[0] = __builtin_amdgcn_s_memtime();
#define WALK_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); ts_[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WALK_STAMP(i)
#endif
    // what the walk asks memory for, in bytes (DESIGN.md 4.2 "algorithmic bytes"): the rare requests are added per
    // lane (lane_bytes), the two of the loop body -- a 32-byte index entry per node event, a sparse-table byte per
    // range pre-test -- are counted per wave with a population count of the lanes that issue them
    uint32_t lane_bytes = 0, n_ent = 0, n_spb = 0;
    uint32_t pi = 0;
    while (pi + 1 < pl.n && unit >= pl.p[pi].wave_end) pi++;
    const WalkPlanDev& q = pl.p[pi];
    // Workgroups are handed to the eight XCDs round-robin (workgroup b runs on XCD b % 8), each with its own L2.  A
    // plan's waves start at a workgroup index that is a multiple of 8 and their number is a multiple of 16
    // (walk_plan_waves), so the plan's workgroup i is on XCD i % 8: XCD x takes the x-th CONTIGUOUS eighth of the
    // plan's tiles -- neighbouring reads of the position-sorted list, i.e. the same few amplicons' lists of the index.
    // (Per plan, not per launch: the plans of a launch differ in cost per read by orders of magnitude.)
    uint32_t tile;
    {
        const uint32_t first = pi ? pl.p[pi - 1].wave_end : 0u;
        const uint32_t wgs = (q.wave_end - first) / WALK_WAVES, wg = (unit - first) / WALK_WAVES;
        tile = ((wg % WALK_XCDS) * (wgs / WALK_XCDS) + wg / WALK_XCDS) * WALK_WAVES + wv;
    }
    const DevWalk ix = m.walks[q.tier];
    const uint32_t slot = tile * 64 + lane;
    const bool have = slot < q.n_list;
    // plain: a lane = a read of the plan's list.  CHUNKED: a lane = a job = (read, chunk of its walk)
    uint32_t rd = 0, chunk = 0, n_chunks = 1, job = 0;
    if (CHUNKED) {
        // the read a job belongs to = the last list position whose first job is <= job (job_off ascends).
        // The wave's jobs are consecutive: its first job is located by a bisection on wave-uniform values
        // (scalar loads), the other lanes' reads lie within the next 64 list positions, whose offsets go to
        // LDS for a short per-lane bisection.
        const uint32_t job_first = q.job0 + tile * 64;
        uint32_t lo = 0, hi = jb.n_list;                  // invariant: job_off[lo] <= job_first < job_off[hi] (or hi == n_list)
        while (hi - lo > 1) {
            const uint32_t mid = (uint32_t)__builtin_amdgcn_readfirstlane((int)((lo + hi) >> 1));
            if (jb.job_off[mid] <= job_first) lo = mid; else hi = mid;
        }
        const uint32_t lp0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
        stk[lane] = lp0 + 1 + lane < jb.n_list ? jb.job_off[lp0 + 1 + lane] : 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (have) {
            job = job_first + lane;
            uint32_t a = 0, b = 64;                       // number of the 64 offsets that are <= job
            while (a < b) {
                const uint32_t mid = (a + b) >> 1;
                if (stk[mid] <= job) a = mid + 1; else b = mid;
            }
            const uint32_t lp = lp0 + a;
            rd = q.list[lp];
            chunk = job - (a ? stk[a - 1] : jb.job_off[lp0]);
            n_chunks = jb.job_n[rd];
            lane_bytes += 4 + 4 + 4;                  // its job offset, list entry and job count
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        rd = have ? q.list[slot] : 0u;
    }
    WALK_STAMP(1);          // job / read decoded
    const uint32_t so = have ? read_off[rd] : 0u;
    const uint32_t k = have ? read_off[rd + 1] - so : 0u;
    const int root_sc = have ? root_score[rd] : 0;
    // The stream's numbers, per LANE: the plans of slot WC_SLOT hold reads of different window crowns (k_route picked
    // one per read: wsid), whose structures are slices of the arena the slot's DevWalk points at (device_mat.hpp:
    // WcInfo); on every other plan the offsets are zero and the numbers the plan's own.
    uint32_t L_n = ix.n, L_rqb = ix.rq_blocks, L_last = ix.last_ent, L_pre = ix.has_pre, L_node = 0, L_head = 0, L_dst = 0;
    size_t L_sp = 0;
    SegNode L_whole = ix.whole;
    if (q.tier == WC_SLOT && have) {
        const WcInfo wi = m.wc_info[wsid[rd]];
        L_n = wi.n; L_rqb = wi.rq_blocks; L_last = wi.last_ent; L_pre = wi.has_pre; L_node = wi.node_off; L_head = wi.head_off;
        L_dst = wi.dst_off; L_sp = (size_t)wi.sp_off; L_whole = wi.whole;
        lane_bytes += 4 + 64;
    }
    // list entry, two offsets, root score; per listed position its word and list head (chunked: also the next head)
    if (have) lane_bytes += (CHUNKED ? 0u : 4u) + 8u + 4u + k * (CHUNKED ? 20u : 12u);

    // ---- set-up: the read's words, the start of every position's list, its first node ----
    uint32_t head[KW];
    int c = 0;
    uint32_t long_off = 0, long_len = 0;     // CHUNKED: the read's longest list (the chunks are its quantiles)
    uint32_t s_pair = 0;
#pragma unroll
    for (int j = 0; j < KW; j++) {
        head[j] = NONE;
        uint32_t s9 = 0;
        if ((uint32_t)j < k) {
            const uint32_t w = read_word[so + j];
            const uint32_t p = w_pos(w);
            // a position beyond the tree's last mutated one has no list: the last sentinel stands in
            IxHead h{L_last, NONE};
            if (p <= m.max_pos) h = ix.ix_head[L_head + p];
            const uint32_t e = h.off;
            s9 = (w >> 20) & 0x1FFu;
            cur_l[j * 64 + lane] = e;
            if (CHUNKED) {
                const uint32_t len = p <= m.max_pos ? ix.ix_head[L_head + p + 1].off - e - 1u : 0u;
                if (len > long_len) { long_len = len; long_off = e; }
            } else {
                head[j] = h.first_node;
            }
            if (!rw_missing(w)) c += ((rw_mut(w) & rw_ref(w)) == 0) ? 1 : 0;
        }
        if (j & 1) S16[(j >> 1) * 64 + lane] = s_pair | (s9 << 16);
        else s_pair = s9;
    }
    WALK_STAMP(2);          // words and list heads staged
    uint32_t n = L_n;               // one past the last node this lane looks at
    uint32_t pos = 0;               // next node nobody has looked at
    uint32_t sp = 0;                // open intervals on the stack
    uint32_t top_end = NONE;
    int top_d = 0;
    if (CHUNKED) {
        // this job's nodes [pos, n): cut at the quantiles of the longest list (k_route made sure it has at
        // least n_chunks entries)
        if (chunk + 1 < n_chunks) { n = ix.ix_ent[long_off + (uint32_t)(((uint64_t)(chunk + 1) * long_len) / n_chunks)].node; lane_bytes += 4; }
        if (chunk) { pos = ix.ix_ent[long_off + (uint32_t)(((uint64_t)chunk * long_len) / n_chunks)].node; lane_bytes += 4; }
        // ---- the state of a sequential walk when it reaches `pos` ----
        // every list's cursor at its first entry >= pos: binary searches, four lists side by side (their
        // loads in flight together); cursors and words live in LDS, so the loops over the lists stay rolled
        const bool mid_stream = have && chunk != 0;
#pragma unroll 1
        for (uint32_t j0 = 0; j0 < (uint32_t)KW; j0 += 4) {
            if (!__ballot(mid_stream && j0 < k)) break;
            uint32_t lo[4], hi[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                lo[u] = 0; hi[u] = 0;
                if (j0 + u < k && mid_stream) {
                    const uint32_t p = w_pos(read_word[so + j0 + u]);
                    lo[u] = cur_l[(j0 + u) * 64 + lane];
                    hi[u] = p <= m.max_pos ? ix.ix_head[L_head + p + 1].off - 1u : lo[u];       // (the sentinel stays out)
                    lane_bytes += 4 + 4;
                }
            }
            // first entry of the list goes to the stack region for a moment: the predecessor test below needs it
            bool searching = true;
            const uint32_t first0 = lo[0], first1 = lo[1], first2 = lo[2], first3 = lo[3];
            while (__ballot(searching)) {
                uint32_t probe[4];
                searching = false;
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) probe[u] = lo[u] < hi[u] ? ix.ix_ent[(lo[u] + hi[u]) >> 1].node : 0u;
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    if (lo[u] < hi[u]) {
                        lane_bytes += 4;
                        const uint32_t mid = (lo[u] + hi[u]) >> 1;
                        if (probe[u] < pos) lo[u] = mid + 1; else hi[u] = mid;
                        searching = searching || lo[u] < hi[u];
                    }
                }
            }
            // the intervals open at `pos`: a list's predecessor entry if its subtree reaches past pos, else the
            // first one up its chain of enclosing entries that does, and every entry enclosing that one
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                if (j0 + u < k && mid_stream) {
                    const uint32_t first = u == 0 ? first0 : u == 1 ? first1 : u == 2 ? first2 : first3;
                    const uint32_t sw = sword((int)(j0 + u));
                    cur_l[(j0 + u) * 64 + lane] = lo[u];
                    uint32_t e = lo[u] > first ? lo[u] - 1u : NONE;
                    IxEnt ent{};
                    while (e != NONE) {
                        ent = ix.ix_ent[e];
                        lane_bytes += 32;
                        if (ent.end > pos) break;
                        e = ent.up;
                    }
                    while (e != NONE) {
                        const int d = enter_delta(ent.word, sw);
                        c += d;
                        if (d != 0) {
                            // insertion by subtree end, outermost at the bottom (the intervals are nested)
                            const uint32_t end = ent.end;
                            uint32_t at = sp;
                            while (at > 0 && (stk[(at - 1) * 64 + lane] >> WALK_DELTA_BITS) < end) { stk[at * 64 + lane] = stk[(at - 1) * 64 + lane]; at--; }
                            stk[at * 64 + lane] = (end << WALK_DELTA_BITS) | (uint32_t)(d + (int)WALK_DELTA_BIAS);
                            sp++;
                        }
                        e = ent.up;
                        if (e != NONE) { ent = ix.ix_ent[e]; lane_bytes += 32; }
                    }
                }
            }
        }
        if (sp) {
            const uint32_t e = stk[(sp - 1) * 64 + lane];
            top_end = e >> WALK_DELTA_BITS;
            top_d = (int)(e & ((1u << WALK_DELTA_BITS) - 1u)) - (int)WALK_DELTA_BIAS;
        }
#pragma unroll
        for (int j = 0; j < KW; j++)
            if ((uint32_t)j < k) head[j] = ix.ix_ent[cur_l[j * 64 + lane]].node;
        lane_bytes += 4 * k;
    }
    if (!have) pos = n;
    WALK_STAMP(3);          // (chunked) start state found
    int bs = root_sc + 1;          // the root always competes: nothing worse can win or tie
    uint32_t br = 0xFFFFFFFFu, cnt = 0, bhu = 0;
    uint32_t iters = 0;

    // a candidate: score, tie-break rank, how many nodes it stands for, has_unique of the node of that rank
    auto take = [&](int sc, uint32_t rk, uint32_t kk, uint32_t hu) {
        if (sc < bs) { bs = sc; br = rk; cnt = kk; bhu = hu; }
        else if (sc == bs) { cnt += kk; if (rk < br) { br = rk; bhu = hu; } }
    };
    if (!CHUNKED) {
        // none of the read's positions is mutated in this stream (most reads of the small crowns): every node
        // scores base + c, and the stream-wide aggregate is the answer
        uint32_t any = head[0];
#pragma unroll
        for (int j = 1; j < KW; j++) any = min(any, head[j]);
        if (pos < n && any == NONE) {
            if (L_whole.cnt && L_whole.base + c <= bs) take(L_whole.base + c, L_whole.rank, L_whole.cnt, L_whole.hu);
            pos = n;
        }
    }

    // small streams: nearly every range between two events holds a node that can tie the best (a crown is
    // made of low-score nodes), so the exact query is issued at once, with the other loads of the iteration;
    // large ones ask the sparse table first (there nearly every range fails it)
    const bool eager = L_n <= m.walk_eager_nodes;
#ifdef WEPP_WALK_STATS
    uint32_t st_live = 0, st_pass = 0;
#endif
    while (__ballot(pos < n)) {
        iters++;
#ifdef WEPP_WALK_STATS
        st_live += (uint32_t)__popcll(__ballot(pos < n));
#endif
        uint32_t i_next = head[0];
#pragma unroll
        for (int j = 1; j < KW; j++) i_next = min(i_next, head[j]);
        const bool live = pos < n;
        const uint32_t stop = min(min(i_next, top_end), n);
        const bool at_node = live && i_next < top_end && i_next < n;
        const unsigned long long b_at = __ballot(at_node);
        n_ent += (uint32_t)__popcll(b_at);
        // ---- everything this iteration reads from memory is requested here, together ----
        // the node of the next event: its record, the list entry that carries it and that list's next node
        IxEnt ent{};
        uint32_t sw = 0, ecur = 0;
        int js = 0;
        if (at_node) {
#pragma unroll
            for (int j = KW - 1; j >= 0; j--) js = head[j] == i_next ? j : js;
            sw = sword(js);
            ecur = cur_l[js * 64 + lane];
            ent = ix.ix_ent[ecur];             // 32 bytes: the mutation, the list's next node and the node's own record
        }
        // the sparse-table byte of [pos, stop): the minimum over [pos, pos + 2^lvl), the first level that reaches
        // `stop`.  A lane whose range ends at the fetched entry's node asks that entry's byte first (use_pre);
        // every other lane's table byte is requested here, with the entry, not behind it
        const bool ranged = live && stop > pos;
        const bool use_pre = ranged && !eager && L_pre && at_node && k >= IX_PRE_MIN_LISTS;
        size_t sp_at = 0;
        uint32_t mn_early = SP_NONE;
        if (ranged && !eager) {
            const uint32_t len = stop - pos;
            const uint32_t lvl = len > 1 ? 32u - (uint32_t)__builtin_clz(len - 1) : 0u;
            sp_at = L_sp + (size_t)lvl * L_n + pos;
            if (!use_pre) mn_early = ix.sp[sp_at];
        }
        n_spb += (uint32_t)__popcll(__ballot(ranged && !eager && !use_pre));
        // ---- the nodes [pos, stop): none of them carries a listed position, c is constant ----
        if (live && stop > pos) {
            const uint32_t last = stop - 1;
            const uint32_t ba = pos / RQ_BLK, bl = last / RQ_BLK;
            bool pass = true;
            if (!eager) {
                // the range ends at the node of the entry fetched above: that entry's byte is the minimum of a
                // superset (everything since its list's previous entry), no table byte needed unless it passes
                // (a read with one or two lists gains nothing: its ranges ARE the ranges between its list's entries, and
                // when such a range cannot be skipped the table byte would be fetched after the entry instead of with it)
                bool by_entry = false;
                uint32_t mn = mn_early;
                if (use_pre) {
                    const uint32_t pb = ent.rank >> IX_RANK_BITS;
                    by_entry = pb == SP_NONE || (pb < SP_CLAMP && (int)pb + c > bs);
                    if (!by_entry) { mn = ix.sp[sp_at]; lane_bytes += 1; }   // (rare on the large streams: the byte of the table after all)
                }
                pass = !by_entry && mn != SP_NONE && (mn >= SP_CLAMP || (int)mn + c <= bs);
            }
#ifdef WEPP_WALK_STATS
            st_pass += (uint32_t)__popcll(__ballot(pass));
#endif
            if (__ballot(pass)) {
                if (pass) {
                    // exact aggregate of the statically eligible nodes of [pos, stop): suffix of the first node's
                    // block, disjoint sparse table over the whole blocks in between, prefix of the last node's
                    // block -- four independent 16-byte loads (flatmat.hpp)
                    SegNode ag{SCORE_INF_DEV, 0xFFFFFFFFu, 0u, 0u};
                    auto join = [&](const SegNode x) {
                        if (x.base < ag.base) ag = x;
                        else if (x.base == ag.base) { ag.cnt += x.cnt; if (x.rank < ag.rank) { ag.rank = x.rank; ag.hu = x.hu; } }
                    };
                    if (ba == bl) {
                        // inside one block: its prefix up to the last node, unless the range starts behind the
                        // block's first node -- then node by node
                        lane_bytes += 16;
                        if (pos == ba * RQ_BLK) join(ix.rq_pre[L_node + last]);
                        else if (stop == L_n || stop == (ba + 1) * RQ_BLK) join(ix.rq_suf[L_node + pos]);
                        else
                            for (uint32_t i = pos; i < stop; i++) {
                                if (i > pos) lane_bytes += 16;
                                const NodeRec x = ix.nrec[L_node + i];
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
                        const SegNode* trow = ix.rq_dst + L_dst + (size_t)L * L_rqb;
                        const SegNode s1 = ix.rq_suf[L_node + pos], s2 = ix.rq_pre[L_node + last];
                        const SegNode s3 = lo <= hi ? trow[lo] : none, s4 = lo < hi ? trow[hi] : none;
                        join(s1); join(s2); join(s3); join(s4);
                        lane_bytes += 32 + (lo <= hi ? 16 : 0) + (lo < hi ? 16 : 0);
                    }
                    if (ag.cnt && ag.base + c <= bs) take(ag.base + c, ag.rank, ag.cnt, ag.hu);
                }
            }
            pos = stop;
        }
        if (live && pos < n) {
            if (!at_node) {
                // the innermost open interval ends here: its nodes are behind us
                c -= top_d;
                sp--;
                if (sp) {
                    const uint32_t e = stk[(sp - 1) * 64 + lane];
                    top_end = e >> WALK_DELTA_BITS;
                    top_d = (int)(e & ((1u << WALK_DELTA_BITS) - 1u)) - (int)WALK_DELTA_BIAS;
                } else { top_end = NONE; top_d = 0; }
            }
        }
        if (b_at) {
            if (at_node) {
                // every listed mutation the node carries (nearly always one: the entry fetched above)
                const uint32_t node = i_next;
                int adj = 0, dcom = 0, dsum = 0;
                const uint32_t end = ent.end;
                bool more = true;
                while (more) {
                    cur_l[js * 64 + lane] = ecur + 1;
                    own_adjust(ent.word, sw, adj, dcom);
                    // descendants take the allele; the root also scores itself with it (usher_mapper.cpp:266-271)
                    if (end > node + 1 || node == 0) dsum += enter_delta(ent.word, sw);
                    more = false;
                    int jn = 0;
#pragma unroll
                    for (int j = KW - 1; j >= 0; j--) {
                        if (j == js) head[j] = ent.next_node;
                        if (head[j] == node) { more = true; jn = j; }
                    }
                    if (more) {
                        js = jn;
                        sw = sword(js);
                        ecur = cur_l[js * 64 + lane];
                        ent = ix.ix_ent[ecur];
                        lane_bytes += 32;
                    }
                }
                const uint32_t nst = ent.nstat;
                const uint32_t nmut = nst & NS_CNT_MASK_DEV, ncom0 = (nst >> 14) & NS_CNT_MASK_DEV;
                const bool leaf = nst & NS_LEAF_DEV, masked = nst & NS_MASKED_DEV, root = nst & NS_ROOT_DEV;
                bool elig;
                int sc;
                uint32_t hu = 0;
                if (root) { elig = true; sc = ent.base + c + dsum; }
                else if (masked) { elig = false; sc = 0; }
                else {
                    sc = ent.base + c + adj;
                    const int ncom = (int)ncom0 + dcom;
                    elig = leaf ? (ncom > 0) : (ncom > 0 || ncom == (int)nmut);     // usher_mapper.cpp:455-456
                    hu = ncom < (int)nmut ? 1u : 0u;                                // :184,199,262
                }
                if (elig && sc <= bs) take(sc, L_pre ? ent.rank & IX_RANK_MASK : ent.rank, 1u, hu);
                if (dsum != 0 && end > node + 1) {
                    // k_route admits a read only if its open intervals always fit (sum of ix_nest <= SD)
                    stk[sp * 64 + lane] = (end << WALK_DELTA_BITS) | (uint32_t)(dsum + (int)WALK_DELTA_BIAS);
                    sp++;
                    top_end = end;
                    top_d = dsum;
                }
                c += dsum;
                pos = node + 1;
            }
        }
    }
    WALK_STAMP(4);          // walked
    if (have) {
        if (CHUNKED) {
            jb.part_score[job] = bs;
            jb.part_rank[job] = br;
            jb.part_cnt[job] = (cnt << 1) | bhu;     // (the job's has_unique rides in bit 0)
            lane_bytes += 12;
        } else {
            lane_bytes += 4 + 16;                    // rank -> BFS index, the four results
            // the root always competes, so br is a rank; the clamp only keeps a broken invariant in bounds
            if (best_bfs_j) best_bfs_j[rd] = m.rank2bfs[br < m.N ? br : 0u];
            if (score_out) score_out[rd] = bs;
            if (num_best) num_best[rd] = cnt;
            if (flags) flags[rd] = bhu ? WEPP_FLAG_HAS_UNIQUE_DEV : 0u;
        }
    }
    // (1024 counters: thousands of waves adding to ONE address queue up at the memory side)
#ifdef WEPP_WALK_STATS
    WALK_STAMP(5);          // results written
    if (work_counter && lane == 0) {
        unsigned long long* wc = work_counter + (CHUNKED ? 16 : 0);
        for (int i = 0; i < 5; i++) atomicAdd(wc + i, ts_[i + 1] - ts_[i]);
        atomicAdd(wc + 5, 1ull);
        atomicAdd(wc + 6, (unsigned long long)iters);
        // lane-level counts: live lanes summed over the iterations, node events, table bytes, exact queries, lanes with work
        atomicAdd(wc + 7, (unsigned long long)st_live);
        atomicAdd(wc + 8, (unsigned long long)n_ent);
        atomicAdd(wc + 9, (unsigned long long)n_spb);
        atomicAdd(wc + 10, (unsigned long long)st_pass);
        atomicAdd(wc + 11, (unsigned long long)__popcll(__ballot(have)));
    }
#else
    // plain walks count in the first half of the slots, chunked ones in the second; the bytes in a second array of
    // WALK_COUNTERS slots behind the iterations
    {
        const uint32_t wave_bytes = wave_sum_u32(lane_bytes);
        if (work_counter && lane == 0) {
            const uint32_t at = (CHUNKED ? WALK_COUNTERS / 2 : 0) + (unit & (WALK_COUNTERS / 2 - 1));
            atomicAdd(work_counter + at, (unsigned long long)iters);
            atomicAdd(work_counter + WALK_COUNTERS + at, (unsigned long long)wave_bytes + 32ull * n_ent + n_spb);
        }
    }
#endif
}

// job counts in list order (the input of the scan)
__global__ void k_gather_jobs(const uint32_t* __restrict__ list, uint32_t n_list, const uint32_t* __restrict__ job_n,
                              uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_list) out[i] = job_n[list[i]];
}
// a wave per 64 reads of the chunked class: a read with few jobs is combined by its own lane, one with many
// by the whole wave (lane-strided loads, butterfly reduction).  part_cnt = (count << 1) | has_unique.
__global__ __launch_bounds__(256) void k_finalize_jobs(DevMAT m, const uint32_t* __restrict__ list, uint32_t n_list, WalkJobs jb,
                                uint32_t* __restrict__ best_bfs_j, int32_t* __restrict__ score,
                                uint32_t* __restrict__ num_best, uint32_t* __restrict__ flags) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n_list;
    const uint32_t r = valid ? list[i] : 0u;
    const uint32_t j0 = valid ? jb.job_off[i] : 0u, nj = valid ? jb.job_n[r] : 0u;
    int bs = 0x7FFFFFFF;
    uint32_t br = 0xFFFFFFFFu, cnt = 0, bhu = 0;
    auto take = [&](int& b, uint32_t& rk, uint32_t& ct, uint32_t& h, int s, uint32_t pr, uint32_t pc, uint32_t ph) {
        if (pc == 0) return;
        if (s < b) { b = s; rk = pr; ct = pc; h = ph; }
        else if (s == b) { ct += pc; if (pr < rk) { rk = pr; h = ph; } }
    };
    constexpr uint32_t SMALL = 48;
    if (nj <= SMALL) {
        // (four partials per round: their loads are in flight together)
        uint32_t c = 0;
        for (; c + 4 <= nj; c += 4) {
            uint32_t pc[4], pr[4];
            int ps[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) { pc[u] = jb.part_cnt[j0 + c + u]; ps[u] = jb.part_score[j0 + c + u]; pr[u] = jb.part_rank[j0 + c + u]; }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) take(bs, br, cnt, bhu, ps[u], pr[u], pc[u] >> 1, pc[u] & 1u);
        }
        for (; c < nj; c++) {
            const uint32_t pc = jb.part_cnt[j0 + c];
            take(bs, br, cnt, bhu, jb.part_score[j0 + c], jb.part_rank[j0 + c], pc >> 1, pc & 1u);
        }
    }
    unsigned long long big = __ballot(nj > SMALL);
    while (big) {
        const int l = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t bj0 = (uint32_t)__builtin_amdgcn_readlane((int)j0, l), bnj = (uint32_t)__builtin_amdgcn_readlane((int)nj, l);
        int ws = 0x7FFFFFFF;
        uint32_t wr = 0xFFFFFFFFu, wc = 0, wh = 0;
        for (uint32_t c = lane; c < bnj; c += 64) {
            const uint32_t pc = jb.part_cnt[bj0 + c];
            take(ws, wr, wc, wh, jb.part_score[bj0 + c], jb.part_rank[bj0 + c], pc >> 1, pc & 1u);
        }
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) {
            const int os = __shfl_xor(ws, msk, 64);
            const uint32_t orr = (uint32_t)__shfl_xor((int)wr, msk, 64), oc = (uint32_t)__shfl_xor((int)wc, msk, 64);
            const uint32_t oh = (uint32_t)__shfl_xor((int)wh, msk, 64);
            take(ws, wr, wc, wh, os, orr, oc, oh);
        }
        if ((int)lane == l) { bs = ws; br = wr; cnt = wc; bhu = wh; }
    }
    if (valid) {
        if (best_bfs_j) best_bfs_j[r] = m.rank2bfs[br < m.N ? br : 0u];
        if (score) score[r] = bs;
        if (num_best) num_best[r] = cnt;
        if (flags) flags[r] = bhu ? WEPP_FLAG_HAS_UNIQUE_DEV : 0u;
    }
}

// -----------------------------------------------------------------------------
// k_scores: the -p mode (--write-parsimony-scores-per-node): node_set_difference
// of EVERY node for every read, in BFS order, +1 for nodes that do not compete
// (usher_common.cpp:403-409, usher_mapper.cpp:449-451,500-505).  One wave per
// (read, chunk of the whole-tree stream), lane = node; an R x N output only makes
// sense for small batches, so no tiling and no pruning here.
// -----------------------------------------------------------------------------
// EMIT (wepp_best_nodes: best_j_vec, usher_common.cpp:376-381, filled at usher_mapper.cpp:475-476,497): the same
// evaluation of every node of the read's OWN stream (the crown k_route picked: every node that can reach the read's
// best score is in it), nothing written per node; a node that competes and attains the read's best score -- known from
// the placement call -- is appended to the read's slice of the list (BFS index; slot from a per-read counter, the
// host sorts a slice afterwards).  `list` = the reads routed to this stream.
struct BestOut {
    const uint32_t* list;            // reads of this stream
    const int32_t* best;             // [n_reads] best score of every read (wepp_place_batch)
    const unsigned long long* off;   // [n_reads + 1] CSR over the reads
    uint32_t* cursor;                // [n_reads] slots taken
    uint32_t* nodes;                 // the list
    const uint32_t* rank2bfs;        // tie-break rank -> BFS index (a stream's nodes carry their global rank)
};
template <bool EMIT>
__global__ __launch_bounds__(64) void k_scores(DevStream m, const uint32_t* __restrict__ dfs2bfs,
                                               const uint32_t* __restrict__ read_off,
                                               const uint32_t* __restrict__ read_word, uint32_t n_reads,
                                               uint32_t blocks_per_chunk, int32_t* __restrict__ out, BestOut bo) {
    const uint32_t lane = threadIdx.x;
    const uint32_t r = EMIT ? bo.list[blockIdx.x % n_reads] : blockIdx.x % n_reads;
    const uint32_t chunk = blockIdx.x / n_reads;
    const uint32_t so = read_off[r], k = read_off[r + 1] - so;
    int c = 0;                                    // wave-uniform running c_S
    for (uint32_t j = 0; j < k; j++) {
        const uint32_t sw = read_word[so + j];
        if (!rw_missing(sw)) c += ((rw_mut(sw) & rw_ref(sw)) == 0) ? 1 : 0;
    }
    const uint32_t b0 = chunk * blocks_per_chunk;
    const uint32_t b1 = min(m.NB, b0 + blocks_per_chunk);
    {
        const uint32_t cpi = b0 / m.cp_stride;
        const uint32_t e0 = m.cp_off[cpi], e1 = m.cp_off[cpi + 1];
        for (uint32_t e = e0; e < e1; e += 64) {
            int d = 0;
            if (e + lane < e1) {
                const uint32_t w = m.cp_word[e + lane];
                const uint32_t sw = find_entry(read_word, so, k, w_pos(w));
                if (sw != NONE) d = enter_delta(w, sw);
            }
#pragma unroll
            for (int msk = 32; msk >= 1; msk >>= 1) d += __shfl_xor(d, msk, 64);
            c += d;
        }
    }
    for (uint32_t b = b0; b < b1; b++) {
        const BlkSum sum = m.blk_sum[b];
        const uint32_t e0 = m.blk_eoff[b], e1 = m.blk_eoff[b + 1];
        const bool nvalid = lane < sum.nn;
        const int64_t key = nvalid ? m.nkey[sum.node0 + lane] : 0;
        const uint32_t st = nvalid ? m.nstat[sum.node0 + lane] : 0;
        int cadd = 0, adj = 0, dcom = 0, net = 0;
        bool touched = false;
        for (uint32_t e = e0; e < e1; e += 64) {
            const bool valid = e + lane < e1;
            const uint32_t w = valid ? m.ev_word[e + lane] : W_PAD_DEV;
            const uint32_t mt = valid ? (uint32_t)m.ev_meta[e + lane] : 0;
            const uint32_t sw = (valid && w != W_PAD_DEV) ? find_entry(read_word, so, k, w_pos(w)) : NONE;
            unsigned long long hm = __ballot(sw != NONE);
            while (hm) {
                const int l = __builtin_ctzll(hm);
                hm &= hm - 1;
                const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)w, l);
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)mt, l) & EV_OFF_MASK_DEV;
                const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)sw, l);
                const int delta = enter_delta(wl, sl);
                if (wl & W_EXIT_DEV) {
                    cadd += (lane >= o) ? -delta : 0;
                    net -= delta;
                } else {
                    if (!(wl & W_LEAF_DEV)) {
                        const bool is_root = (sum.node0 + o) == 0;
                        cadd += (lane > o || (is_root && lane == o)) ? delta : 0;
                        net += delta;
                    }
                    if (lane == o) {
                        touched = true;
                        own_adjust(wl, sl, adj, dcom);
                    }
                }
            }
        }
        const int base = (int)(key >> 32);
        const uint32_t nmut = st & NS_CNT_MASK_DEV;
        const uint32_t ncom0 = (st >> 14) & NS_CNT_MASK_DEV;
        const bool leaf = st & NS_LEAF_DEV, masked = st & NS_MASKED_DEV, root = st & NS_ROOT_DEV;
        bool elig;
        int score = base + c + cadd;
        if (root) elig = true;
        else if (masked) elig = false;
        else if (touched) {
            score += adj;
            const int ncom = (int)ncom0 + dcom;
            elig = leaf ? (ncom > 0) : (ncom > 0 || ncom == (int)nmut);
        } else elig = st & NS_ELIG0_DEV;
        if (EMIT) {
            if (nvalid && elig && score == bo.best[r]) {
                const uint32_t slot = atomicAdd(&bo.cursor[r], 1u);
                if (bo.off[r] + slot < bo.off[r + 1]) bo.nodes[bo.off[r] + slot] = bo.rank2bfs[(uint32_t)key];
            }
        } else if (nvalid) out[(size_t)r * m.n + dfs2bfs[sum.node0 + lane]] = elig ? score : score + 1;
        c += net;
    }
}

// -----------------------------------------------------------------------------
// k_imputed: allele imputed for an ambiguous read entry at the chosen node
// (usher_mapper.cpp:293-378 with compute_vecs).  The node's genotype as the
// scorer sees it: its own mutation at the position if it is "common" with the
// read (:205-236; none of its mutations when the node is masked, :198-201; all
// of them for the root, :266-271), else the most recent mutation on the path
// above it (:276-287).  One thread per (read, entry) pair.
// -----------------------------------------------------------------------------
__global__ void k_imputed(DevMAT m, const uint32_t* __restrict__ read_off, const uint32_t* __restrict__ read_word,
                          const uint32_t* __restrict__ best_bfs_j, const uint32_t* __restrict__ pairs,
                          uint32_t n_pairs, uint8_t* __restrict__ nuc_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const uint32_t r = pairs[2 * i], wi = pairs[2 * i + 1];
    const uint32_t s = read_word[wi];
    const uint32_t pos = w_pos(s), a = rw_mut(s), sref = rw_ref(s);
    uint32_t d = m.bfs2dfs[best_bfs_j[r]];
    uint32_t anc = 0;          // allele of the genotype at pos, 0 = no mutation found
    bool first = true;
    for (;;) {
        const uint32_t st = m.nstat[d];
        const bool root = st & NS_ROOT_DEV;
        const bool own_ok = !first || root || !(st & NS_MASKED_DEV);
        if (own_ok) {
            for (uint32_t w = m.node_woff[d]; w < m.node_woff[d + 1]; w++) {
                const uint32_t tw = m.words[w];
                if (w_pos(tw) != pos) continue;
                // the placement node itself contributes only a mutation shared with the read
                if (first && !root && (a & tw_mut(tw)) == 0) break;
                anc = tw_mut(tw);
                break;
            }
        }
        if (anc || root) break;
        d = m.parent_dfs[d];
        first = false;
    }
    const bool found_pos = anc != 0;
    const bool found = found_pos && (a & anc) != 0;
    const bool has_ref = (a & sref) != 0;
    uint32_t out;
    if (found) out = anc;                               // :323-335
    else if (!found_pos && has_ref) out = sref;         // :342-351
    else out = has_ref ? sref : (a & (0u - a));         // :357-377 (lowest set bit)
    nuc_out[i] = (uint8_t)out;
}

// -----------------------------------------------------------------------------
// k_excess: node_excess_mutations of a (sample, node) pair as mapper2_body appends them with
// compute_vecs -- usher prints the first `score` of them for the optimal nodes in the last
// column of parsimony-scores.tsv (usher_common.cpp:555-574):
//   (0) the node's own mutations the sample shares (usher_mapper.cpp:223-228, :253-258),
//   (1) the sample's alleles that E does not offer, in sample order (:357-388),
//   (2) E's non-reference alleles at positions the sample does not list, in position
//       order (:394-446),
// E = the genotype the scorer sees at the node: its own shared mutations (none when the node
// is masked, all of them for the root), then the most recent mutation per position on the
// path above.  One thread per pair, run twice: count, then emit at the offsets the host
// derived from the counts.
// -----------------------------------------------------------------------------
namespace {
// does the node's own mutation `tw` enter E for a sample entry `s` (NONE = position not listed)?
__device__ __forceinline__ bool own_in_E(uint32_t tw, uint32_t s) {
    if (s == NONE) return tw_mut(tw) == tw_ref(tw);                 // :245 back-mutation to the reference
    return !rw_missing(s) && (rw_mut(s) & tw_mut(tw)) != 0;         // :211-216 (a missing base shares but is not recorded)
}
// allele of E at `pos` (0 = none); s = the sample's entry at pos or NONE
__device__ uint32_t allele_of_E(const DevMAT& m, uint32_t d, uint32_t pos, uint32_t s) {
    bool first = true;
    for (;;) {
        const uint32_t st = m.nstat[d];
        const bool root = st & NS_ROOT_DEV;
        if (!first || root || !(st & NS_MASKED_DEV)) {
            for (uint32_t w = m.node_woff[d]; w < m.node_woff[d + 1]; w++) {
                const uint32_t tw = m.words[w];
                if (w_pos(tw) != pos) continue;
                if (first && !root && !own_in_E(tw, s)) break;
                return tw_mut(tw);
            }
        }
        if (root) return 0;
        d = m.parent_dfs[d];
        first = false;
    }
}
}  // namespace

__global__ void k_excess(DevMAT m, const uint32_t* __restrict__ read_off, const uint32_t* __restrict__ read_word,
                         const uint32_t* __restrict__ pair_read, const uint32_t* __restrict__ pair_bfs_j,
                         uint32_t n_pairs, const unsigned long long* __restrict__ out_off, uint32_t* __restrict__ counts,
                         uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const bool emit = out_off != nullptr;
    uint32_t* o = emit ? out + out_off[i] : nullptr;
    const uint32_t r = pair_read[i];
    const uint32_t s0 = read_off[r], k = read_off[r + 1] - s0;
    const uint32_t d0 = m.bfs2dfs[pair_bfs_j[i]];
    const uint32_t st0 = m.nstat[d0];
    const bool root0 = st0 & NS_ROOT_DEV;
    uint32_t n = 0;
    // (0) the node's own shared mutations
    if (!root0 && !(st0 & NS_MASKED_DEV)) {
        for (uint32_t w = m.node_woff[d0]; w < m.node_woff[d0 + 1]; w++) {
            const uint32_t tw = m.words[w];
            if (!own_in_E(tw, find_entry(read_word, s0, k, w_pos(tw)))) continue;
            const uint32_t par = tw_par(tw) ? tw_par(tw) : tw_ref(tw);
            if (emit) o[n] = w_pos(tw) | (tw_ref(tw) << 20) | (par << 24) | (tw_mut(tw) << 28);
            n++;
        }
    }
    // (1) the sample's own alleles
    for (uint32_t j = 0; j < k; j++) {
        const uint32_t s = read_word[s0 + j];
        if (rw_missing(s)) continue;
        const uint32_t pos = w_pos(s), a = rw_mut(s), sref = rw_ref(s);
        const uint32_t anc = allele_of_E(m, d0, pos, s);
        const bool found_pos = anc != 0, found = found_pos && (a & anc) != 0, has_ref = (a & sref) != 0;
        if (found || (!found_pos && has_ref)) continue;
        const uint32_t mnuc = has_ref ? sref : (a & (0u - a));
        const uint32_t par = found_pos ? anc : sref;
        if (mnuc == par) continue;
        if (emit) o[n] = pos | (sref << 20) | (par << 24) | (mnuc << 28);
        n++;
    }
    // (2) back-mutations: E's alleles at positions the sample does not list
    const uint32_t n1 = n;
    uint32_t d = d0;
    bool first = true;
    for (;;) {
        const uint32_t st = m.nstat[d];
        const bool root = st & NS_ROOT_DEV;
        if (!first || root || !(st & NS_MASKED_DEV)) {
            for (uint32_t w = m.node_woff[d]; w < m.node_woff[d + 1]; w++) {
                const uint32_t tw = m.words[w];
                const uint32_t pos = w_pos(tw), mut = tw_mut(tw), ref = tw_ref(tw);
                if (mut == ref) continue;                                   // :421
                if (find_entry(read_word, s0, k, pos) != NONE) continue;   // :417-419, :423
                if (first && !root) continue;     // an own mutation at an unlisted position is in E only as mut == ref
                // E keeps the most recent mutation of a position only
                bool earlier = false;
                {
                    uint32_t e = d0;
                    bool ef = true;
                    while (e != d && !earlier) {
                        const uint32_t est = m.nstat[e];
                        if (!ef || !(est & NS_MASKED_DEV))
                            for (uint32_t x = m.node_woff[e]; x < m.node_woff[e + 1]; x++)
                                if (w_pos(m.words[x]) == pos && (!ef || own_in_E(m.words[x], NONE))) earlier = true;
                        e = m.parent_dfs[e];
                        ef = false;
                    }
                }
                if (earlier) continue;
                if (emit) {
                    // insertion by position among the back-mutations
                    uint32_t q = n;
                    while (q > n1 && (o[q - 1] & 0xFFFFFu) > pos) { o[q] = o[q - 1]; q--; }
                    o[q] = pos | (ref << 20) | (mut << 24) | (ref << 28);
                }
                n++;
            }
        }
        if (root) break;
        d = m.parent_dfs[d];
        first = false;
    }
    if (!emit) counts[i] = n;
}

// -----------------------------------------------------------------------------
// launchers (called from capi.cpp)
// -----------------------------------------------------------------------------
hipError_t launch_route(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word, uint32_t n_reads,
                        int use_crowns, uint32_t walk_max_events, uint32_t job_events, uint32_t stack8, uint32_t stack16,
                        uint32_t* job_n, uint8_t* tier_of,
                        int32_t* root_score, uint32_t* blk_counts, uint32_t* tier_info, uint32_t* slot_in_blk,
                        uint32_t* tier_info_next, uint32_t* wsid, hipStream_t stream) {
    hipLaunchKernelGGL(k_route, dim3(ROUTE_BLOCKS), dim3(ROUTE_THREADS), 0, stream, m, d_read_off, d_read_word,
                       n_reads, use_crowns, walk_max_events, job_events, std::min(stack8, WALK8_STACK), std::min(stack16, WALK16_STACK),
                       job_n, tier_of, root_score, blk_counts, tier_info, slot_in_blk, tier_info_next, wsid);
    return hipGetLastError();
}

hipError_t launch_first_pos(const uint32_t* list, uint32_t n, const uint32_t* d_read_off, const uint32_t* d_read_word,
                            uint32_t* keys, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_first_pos, dim3((n + 255) / 256), dim3(256), 0, stream, list, n, d_read_off, d_read_word, keys);
    return hipGetLastError();
}

hipError_t launch_walk_keys(const uint32_t* list, uint32_t n, const uint8_t* tier_of, const uint32_t* d_read_off,
                            const uint32_t* d_read_word, uint32_t* keys, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_walk_keys, dim3((n + 255) / 256), dim3(256), 0, stream, list, n, tier_of, d_read_off, d_read_word, keys);
    return hipGetLastError();
}

hipError_t launch_scatter(const uint8_t* tier_of, const uint32_t* slot_in_blk, uint32_t n_reads, const uint32_t* blk_counts,
                          uint32_t* tier_info, uint32_t* list, hipStream_t stream) {
    hipLaunchKernelGGL(k_scatter, dim3(ROUTE_BLOCKS), dim3(ROUTE_THREADS), 0, stream, tier_of, slot_in_blk, n_reads,
                       blk_counts, tier_info, list);
    return hipGetLastError();
}

hipError_t launch_sweep(const DevMAT& m, const DevStream& st, const uint32_t* d_read_off,
                        const uint32_t* d_read_word, const int32_t* root_score, const uint32_t* list,
                        uint32_t n_list, uint32_t T,
                        uint32_t ntiles, uint32_t nchunks, uint32_t blocks_per_chunk, bool s_in_lds, bool dense,
                        bool win_table, uint32_t ent_cap, uint32_t key_cap, uint32_t lds_bytes, int32_t* part_score,
                        uint32_t* part_rank, uint32_t* part_cnt, hipStream_t stream) {
    // nchunks is a multiple of DENSE_WAVES_PER_WG for the dense variant (capi.cpp)
    const dim3 grid(dense ? ntiles * (nchunks / DENSE_WAVES_PER_WG) : ntiles * nchunks);
    const dim3 block(dense ? 64 * DENSE_WAVES_PER_WG : 64);
#define WEPP_SWEEP(A, B, C, CAP, KCAP)                                                                                \
    hipLaunchKernelGGL((k_sweep<A, B, C>), grid, block, lds_bytes, stream, st, m.bm_words, m.max_pos, CAP, KCAP,       \
                       d_read_off, d_read_word, root_score, list, n_list, T, ntiles, blocks_per_chunk, part_score,    \
                       part_rank, part_cnt)
    // (win_table: key_cap carries the window's first position)
    if (s_in_lds && dense && win_table) WEPP_SWEEP(true, true, true, ent_cap, key_cap);
    else if (s_in_lds && dense) WEPP_SWEEP(true, true, false, ent_cap, key_cap);
    else if (s_in_lds) WEPP_SWEEP(true, false, false, ent_cap, 0u);
    else WEPP_SWEEP(false, false, false, 0u, 0u);
#undef WEPP_SWEEP
    return hipGetLastError();
}

hipError_t launch_sweep_arena(const DevMAT& m, const DevStream* wc_streams, const uint32_t* wsid, const uint32_t* d_read_off,
                              const uint32_t* d_read_word, const int32_t* root_score, const uint32_t* list, uint32_t n_list,
                              uint32_t ent_cap, uint32_t lds_bytes, int32_t* part_score, uint32_t* part_rank, uint32_t* part_cnt,
                              hipStream_t stream) {
    if (n_list == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sweep_arena, dim3(n_list * ARENA_CHUNKS), dim3(64), lds_bytes, stream, wc_streams, wsid, m.bm_words, m.max_pos, ent_cap,
                       d_read_off, d_read_word, root_score, list, n_list, part_score, part_rank, part_cnt);
    return hipGetLastError();
}

hipError_t launch_sweep_multi(const DevMAT& m, const SweepPlans& pl, const uint32_t* d_read_off,
                              const uint32_t* d_read_word, const int32_t* root_score, uint32_t lds_bytes,
                              hipStream_t stream) {
    if (pl.n == 0) return hipSuccess;
    // lds_bytes = the largest request of one sweep; every wave of a workgroup gets that much
    const uint32_t units = pl.p[pl.n - 1].wg_end;
    hipLaunchKernelGGL(k_sweep_multi, dim3((units + SWEEP_WAVES - 1) / SWEEP_WAVES), dim3(64 * SWEEP_WAVES),
                       lds_bytes * SWEEP_WAVES, stream, pl, m.bm_words, m.max_pos, lds_bytes / 4, d_read_off, d_read_word,
                       root_score);
    return hipGetLastError();
}

hipError_t launch_finalize_multi(const DevMAT& m, const SweepPlans& pl, const uint32_t* d_read_off,
                                 const uint32_t* d_read_word, uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best,
                                 uint32_t* flags, hipStream_t stream) {
    if (pl.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_finalize_multi, dim3(pl.p[pl.n - 1].fin_end), dim3(256), 0, stream, m, pl, d_read_off,
                       d_read_word, best_bfs_j, score, num_best, flags);
    return hipGetLastError();
}

hipError_t launch_finalize(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word,
                           const uint32_t* list, uint32_t n_list, uint32_t nchunks, const int32_t* part_score,
                           const uint32_t* part_rank, const uint32_t* part_cnt, uint32_t* best_bfs_j,
                           int32_t* score, uint32_t* num_best, uint32_t* flags, hipStream_t stream) {
    const uint32_t lpr = finalize_lanes_per_read(nchunks);
    const dim3 grid(finalize_blocks(n_list, nchunks)), block(256);
#define FIN(L) hipLaunchKernelGGL(k_finalize<L>, grid, block, 0, stream, m, d_read_off, d_read_word, list, n_list, nchunks, \
                                  part_score, part_rank, part_cnt, best_bfs_j, score, num_best, flags)
    if (lpr == 1) FIN(1);
    else if (lpr == 4) FIN(4);
    else if (lpr == 16) FIN(16);
    else FIN(64);
#undef FIN
    return hipGetLastError();
}

// LDS of a walk workgroup: WALK_WAVES waves, each KW / 2 rows of read alleles + KW rows of cursors + sd_rows of
// stack (64 lanes x 4 bytes a row)
static uint32_t walk_lds_bytes(uint32_t kw, uint32_t sd_rows) { return WALK_WAVES * (kw / 2 + kw + sd_rows) * 256; }
// stack rows of a launch: the deepest stack its reads can need (k_route), at least one row for the job decode's
// scratch, never more than the class admits
static uint32_t walk_stack_rows(uint32_t open_max, uint32_t sd) { return std::min(sd, std::max(open_max, 2u)); }

hipError_t launch_walk(const DevMAT& m, const WalkPlans& pl, uint32_t cls, uint32_t open_max, const uint32_t* d_read_off,
                       const uint32_t* d_read_word, const int32_t* root_score, uint32_t* best_bfs_j, int32_t* score,
                       uint32_t* num_best, uint32_t* flags, unsigned long long* work_counter, const uint32_t* wsid, hipStream_t stream) {
    if (pl.n == 0) return hipSuccess;
    const uint32_t waves = pl.p[pl.n - 1].wave_end;
    const dim3 grid((waves + WALK_WAVES - 1) / WALK_WAVES), block(64 * WALK_WAVES);
    const WalkJobs none{};
    if (cls == PLAN_WALK8) {
        const uint32_t sd = walk_stack_rows(open_max, WALK8_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK8_K, (int)WALK8_STACK, false>), grid, block, walk_lds_bytes(WALK8_K, sd), stream, m, pl,
                           none, sd, d_read_off, d_read_word, root_score, best_bfs_j, score, num_best, flags, work_counter, wsid);
    } else {
        const uint32_t sd = walk_stack_rows(open_max, WALK16_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK16_K, (int)WALK16_STACK, false>), grid, block, walk_lds_bytes(WALK16_K, sd), stream, m, pl,
                           none, sd, d_read_off, d_read_word, root_score, best_bfs_j, score, num_best, flags, work_counter, wsid);
    }
    return hipGetLastError();
}

hipError_t launch_gather_jobs(const uint32_t* list, uint32_t n_list, const uint32_t* job_n, uint32_t* out, hipStream_t stream) {
    if (n_list == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_jobs, dim3((n_list + 255) / 256), dim3(256), 0, stream, list, n_list, job_n, out);
    return hipGetLastError();
}

hipError_t launch_walk_jobs(const DevMAT& m, const WalkPlans& pl, uint32_t cls, uint32_t open_max, const WalkJobs& jb,
                            const uint32_t* d_read_off, const uint32_t* d_read_word, const int32_t* root_score,
                            unsigned long long* work_counter, const uint32_t* wsid, hipStream_t stream) {
    if (pl.n == 0) return hipSuccess;
    const uint32_t waves = pl.p[pl.n - 1].wave_end;
    const dim3 grid((waves + WALK_WAVES - 1) / WALK_WAVES), block(64 * WALK_WAVES);
    if (cls == PLAN_WALKC8) {
        const uint32_t sd = walk_stack_rows(open_max, WALK8_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK8_K, (int)WALK8_STACK, true>), grid, block, walk_lds_bytes(WALK8_K, sd), stream, m, pl, jb,
                           sd, d_read_off, d_read_word, root_score, (uint32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr,
                           (uint32_t*)nullptr, work_counter, wsid);
    } else {
        const uint32_t sd = walk_stack_rows(open_max, WALK16_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK16_K, (int)WALK16_STACK, true>), grid, block, walk_lds_bytes(WALK16_K, sd), stream, m, pl, jb,
                           sd, d_read_off, d_read_word, root_score, (uint32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr,
                           (uint32_t*)nullptr, work_counter, wsid);
    }
    return hipGetLastError();
}

hipError_t launch_finalize_jobs(const DevMAT& m, const uint32_t* list, uint32_t n_list, const WalkJobs& jb,
                                const uint32_t* d_read_off, const uint32_t* d_read_word, uint32_t* best_bfs_j,
                                int32_t* score, uint32_t* num_best, uint32_t* flags, hipStream_t stream) {
    if (n_list == 0) return hipSuccess;
    hipLaunchKernelGGL(k_finalize_jobs, dim3((n_list + 255) / 256), dim3(256), 0, stream, m, list, n_list, jb,
                       best_bfs_j, score, num_best, flags);
    return hipGetLastError();
}

hipError_t launch_scores(const DevMAT& m, const DevStream& full, const uint32_t* d_read_off,
                         const uint32_t* d_read_word, uint32_t n_reads, int32_t* d_out, hipStream_t stream) {
    // enough waves to fill the chip, cut at checkpoints
    uint32_t nchunks = std::max<uint32_t>(1, (8192 + n_reads - 1) / n_reads);
    nchunks = std::min(nchunks, full.ncp);
    const uint32_t bpc = ((full.ncp + nchunks - 1) / nchunks) * full.cp_stride;
    nchunks = (full.NB + bpc - 1) / bpc;
    hipLaunchKernelGGL(k_scores<false>, dim3(n_reads * nchunks), dim3(64), 0, stream, full, m.dfs2bfs, d_read_off,
                       d_read_word, n_reads, bpc, d_out, BestOut{});
    return hipGetLastError();
}

// the optimal nodes of the reads `list` (routed to stream `st`), see k_scores<true>
hipError_t launch_best_nodes(const DevMAT& m, const DevStream& st, const uint32_t* d_read_off, const uint32_t* d_read_word,
                             const uint32_t* list, uint32_t n_list, const int32_t* d_best, const unsigned long long* d_off,
                             uint32_t* d_cursor, uint32_t* d_nodes, hipStream_t stream) {
    if (n_list == 0) return hipSuccess;
    // enough waves to fill the chip, cut at checkpoints, no chunk shorter than 8 blocks
    uint32_t nchunks = std::max<uint32_t>(1, (8192 + n_list - 1) / n_list);
    nchunks = std::min<uint32_t>(nchunks, std::max<uint32_t>(1, st.NB / 8));
    nchunks = std::min(nchunks, st.ncp);
    const uint32_t bpc = ((st.ncp + nchunks - 1) / nchunks) * st.cp_stride;
    nchunks = (st.NB + bpc - 1) / bpc;
    if ((uint64_t)n_list * nchunks >= (1ull << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_scores<true>, dim3(n_list * nchunks), dim3(64), 0, stream, st, (const uint32_t*)nullptr, d_read_off,
                       d_read_word, n_list, bpc, (int32_t*)nullptr, BestOut{list, d_best, d_off, d_cursor, d_nodes, m.rank2bfs});
    return hipGetLastError();
}

hipError_t launch_imputed(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word,
                          const uint32_t* d_best_bfs_j, const uint32_t* d_pairs, uint32_t n_pairs,
                          uint8_t* d_nuc, hipStream_t stream) {
    if (n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_imputed, dim3((n_pairs + 255) / 256), dim3(256), 0, stream, m, d_read_off, d_read_word,
                       d_best_bfs_j, d_pairs, n_pairs, d_nuc);
    return hipGetLastError();
}

hipError_t launch_excess(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word,
                         const uint32_t* d_pair_read, const uint32_t* d_pair_bfs_j, uint32_t n_pairs,
                         const unsigned long long* d_out_off, uint32_t* d_counts, uint32_t* d_out, hipStream_t stream) {
    if (n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_excess, dim3((n_pairs + 127) / 128), dim3(128), 0, stream, m, d_read_off, d_read_word,
                       d_pair_read, d_pair_bfs_j, n_pairs, d_out_off, d_counts, d_out);
    return hipGetLastError();
}

#ifdef WEPP_SWEEP_STATS
extern "C" int wepp_debug_sweep_stats(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sweep_stats), sizeof(unsigned long long) * MAX_STREAMS * NSTAT) != hipSuccess) return 1;
    if (reset) {
        static unsigned long long zero[MAX_STREAMS * NSTAT];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_stats), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

hipError_t sweep_set_max_lds(uint32_t bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)k_sweep<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_sweep<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_sweep<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_sweep_arena, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_sweep_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_sweep<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace wepp
