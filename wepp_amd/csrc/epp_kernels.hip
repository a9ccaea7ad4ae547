// epp_kernels.hip -- WEPP's own read placement on CDNA4: for every read, the windowed
// parsimony against every haplotype (node) of the tree, the minimum, the set of nodes that
// attain it (the EPPs) and the haplotype scores / per-bin read counts accumulated from them.
// Reference: single_read_tree + wepp_filter::cartesian_map, src/WEPP/initial_filter.cpp:41-239
// (distance = haplotype::mutation_distance, src/WEPP/haplotype.hpp:123-173).
//
// Closed form.  For read r with window [s, e] and node n with root-path genotype G_n:
//   D_r(n) = #{listed non-N positions of r} + sum over mutations m on the root path of n with
//            s <= m.pos <= e of  cost_r(m.mut) - cost_r(m.par),
//   cost_r(x at p) = p listed in r ? (N ? 0 : x != allele_r(p)) : (x != ref(p))   (exact equality,
//   initial_filter.cpp:66-68), which is what the reference's list of mismatching positions has
//   as its length at n (:90).  The EPP event stream (flatmat.hpp) applies +delta when a
//   pre-order walk enters the mutation's node and -delta when it leaves the subtree, so a linear
//   scan carries D_r for 64 reads at once (lane = read).  Between two consecutive events the
//   distance is constant and belongs to a contiguous range of pre-order node indices: minima,
//   multiplicities and score updates are handled per RANGE, never per node.
//
//   k_epp_select_*  cut the MAT's event stream down to the events inside a genome window
//                   (the role of the reference's range trees, arena.cpp:68-152)
//   k_epp_sweep<1>  per (tile of 64 reads, chunk of a window stream): minimum of the relative
//                   distance over the chunk's nodes, how many nodes attain it, net change
//   k_epp_combine   per read: distance at every chunk start, global minimum (max_parismony),
//                   multiplicity, node_score delta (initial_filter.hpp:54-57)
//   k_epp_sweep<2>  same walk; where a read's distance equals its minimum the node range gets
//                   +delta .. -delta in a difference array (score: 64-bit fixed point, so that
//                   the prefix sum cancels exactly and the result does not depend on the order of
//                   the atomics; per-bin read counts: int) and, for reads with few EPPs, the
//                   node indices go to the read's EPP list (:205-210)
//   k_epp_scan_*    prefix sums of the difference arrays -> score, mapped_read_counts, divergence
// Integer, HBM/L2-bound streaming; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "epp.hpp"

namespace wepp {

namespace {

constexpr uint32_t W_POS = 0xFFFFFu;
constexpr uint32_t W_EXIT_BIT = 1u << 30;
constexpr uint32_t PAD_WORD = 0x000FFFFFu;
constexpr int INT_INF = 0x7FFFFFFF;

// Wave-wide reductions, returned wave-uniform: four DPP row shifts leave the result of every row of
// 16 lanes in its last lane, four readlanes and scalar operations finish (no LDS permutes).
#define WEPP_DPP_SHR(old, x, n) __builtin_amdgcn_update_dpp((int)(old), (int)(x), 0x110 + (n), 0xF, 0xF, false)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)WEPP_DPP_SHR(-1, v, 1));
    v = min(v, (uint32_t)WEPP_DPP_SHR(-1, v, 2));
    v = min(v, (uint32_t)WEPP_DPP_SHR(-1, v, 4));
    v = min(v, (uint32_t)WEPP_DPP_SHR(-1, v, 8));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)WEPP_DPP_SHR(0, v, 1));
    v = max(v, (uint32_t)WEPP_DPP_SHR(0, v, 2));
    v = max(v, (uint32_t)WEPP_DPP_SHR(0, v, 4));
    v = max(v, (uint32_t)WEPP_DPP_SHR(0, v, 8));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += WEPP_DPP_SHR(0, v, 1);
    v += WEPP_DPP_SHR(0, v, 2);
    v += WEPP_DPP_SHR(0, v, 4);
    v += WEPP_DPP_SHR(0, v, 8);
    return __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31) + __builtin_amdgcn_readlane(v, 47) +
           __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#define WEPP_SUM64_STEP(n)                                                                                   \
    {                                                                                                        \
        const uint32_t lo = (uint32_t)WEPP_DPP_SHR(0, (uint32_t)v, n), hi = (uint32_t)WEPP_DPP_SHR(0, (uint32_t)(v >> 32), n); \
        v += (long long)(((unsigned long long)hi << 32) | lo);                                               \
    }
    WEPP_SUM64_STEP(1) WEPP_SUM64_STEP(2) WEPP_SUM64_STEP(4) WEPP_SUM64_STEP(8)
#undef WEPP_SUM64_STEP
    long long r = 0;
#pragma unroll
    for (int l = 15; l < 64; l += 16)
        r += (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l));
    return r;
}
#undef WEPP_DPP_SHR
// groups that may hold position p: ws ascending, we_max = running maximum of we
__device__ __forceinline__ void candidate_groups(const uint32_t* ws, const uint32_t* wemax, uint32_t G, uint32_t p,
                                                 uint32_t& g_lo, uint32_t& g_hi_excl) {
    uint32_t lo = 0, hi = G;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (ws[mid] <= p) lo = mid + 1; else hi = mid; }
    g_hi_excl = lo;                         // groups [0, lo) start at or before p
    lo = 0; hi = G;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (wemax[mid] < p) lo = mid + 1; else hi = mid; }
    g_lo = lo;                              // groups before lo end before p
}

}  // namespace

// ---------------------------------------------------------------------------------------
// window selection
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_epp_select_count(const uint32_t* __restrict__ ev_word, uint64_t n_events,
                                                         const EppGroup* __restrict__ groups,
                                                         const uint32_t* __restrict__ we_max, uint32_t G, uint32_t nblk,
                                                         uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s_ws[EPP_MAX_GROUPS], s_we[EPP_MAX_GROUPS], s_wemax[EPP_MAX_GROUPS], s_c[EPP_MAX_GROUPS];
    const uint32_t lane = threadIdx.x;
    for (uint32_t g = lane; g < G; g += 64) { s_ws[g] = groups[g].ws; s_we[g] = groups[g].we; s_wemax[g] = we_max[g]; s_c[g] = 0; }
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * EPP_SEL_EVENTS;
    for (uint32_t i = lane; i < EPP_SEL_EVENTS; i += 64) {
        const uint64_t e = base + i;
        if (e >= n_events) break;
        const uint32_t p = ev_word[e] & W_POS;
        uint32_t g0, g1;
        candidate_groups(s_ws, s_wemax, G, p, g0, g1);
        for (uint32_t g = g0; g < g1; g++)
            if (s_we[g] >= p) atomicAdd(&s_c[g], 1u);
    }
    __syncthreads();
    for (uint32_t g = lane; g < G; g += 64) cnt[(size_t)g * nblk + blockIdx.x] = s_c[g];
}

// one workgroup per window: exclusive scan of its row of block counts, total -> totals[g]
__global__ __launch_bounds__(256) void k_epp_select_scan(uint32_t* __restrict__ cnt, uint32_t nblk,
                                                         uint32_t* __restrict__ totals) {
    __shared__ uint32_t s[256];
    __shared__ uint32_t carry;
    uint32_t* row = cnt + (size_t)blockIdx.x * nblk;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nblk; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < nblk ? row[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 256; d <<= 1) {
            const uint32_t t = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t c0 = carry;
        if (i < nblk) row[i] = c0 + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = c0 + s[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(64) void k_epp_select_scatter(const uint32_t* __restrict__ ev_word,
                                                           const uint32_t* __restrict__ ev_node, uint64_t n_events,
                                                           const EppGroup* __restrict__ groups,
                                                           const uint32_t* __restrict__ we_max, uint32_t G,
                                                           uint32_t nblk, const uint32_t* __restrict__ cnt,
                                                           uint32_t* __restrict__ st_word,
                                                           uint32_t* __restrict__ st_node) {
    __shared__ uint32_t s_ws[EPP_MAX_GROUPS], s_we[EPP_MAX_GROUPS], s_wemax[EPP_MAX_GROUPS];
    __shared__ unsigned long long s_cur[EPP_MAX_GROUPS];
    const uint32_t lane = threadIdx.x;
    for (uint32_t g = lane; g < G; g += 64) {
        s_ws[g] = groups[g].ws; s_we[g] = groups[g].we; s_wemax[g] = we_max[g];
        s_cur[g] = groups[g].soff + cnt[(size_t)g * nblk + blockIdx.x];
    }
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * EPP_SEL_EVENTS;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t i0 = 0; i0 < EPP_SEL_EVENTS; i0 += 64) {
        const uint64_t e = base + i0 + lane;
        if (base + i0 >= n_events) break;
        uint32_t w = PAD_WORD, nd = 0, g = 0xFFFFFFFFu, g1 = 0;
        if (e < n_events) {
            w = ev_word[e];
            nd = ev_node[e];
            const uint32_t p = w & W_POS;
            uint32_t g0;
            candidate_groups(s_ws, s_wemax, G, p, g0, g1);
            g = g0;
            while (g < g1 && s_we[g] < p) g++;
            if (g >= g1) g = 0xFFFFFFFFu;
        }
        // the events of one window must keep their order: windows are served one at a time,
        // lanes (= consecutive events) take consecutive slots
        while (true) {
            const uint32_t gmin = wave_min_u32(g);
            if (gmin == 0xFFFFFFFFu) break;
            const bool mine = g == gmin;
            const unsigned long long mask = __ballot(mine);
            const unsigned long long at = s_cur[gmin];
            if (mine) {
                const unsigned long long slot = at + (unsigned long long)__popcll(mask & lt);
                st_word[slot] = w;
                st_node[slot] = nd;
                const uint32_t p = w & W_POS;
                g++;
                while (g < g1 && s_we[g] < p) g++;
                if (g >= g1) g = 0xFFFFFFFFu;
            }
            __syncthreads();
            if (lane == 0) s_cur[gmin] = at + (unsigned long long)__popcll(mask);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------
// sweep
// ---------------------------------------------------------------------------------------
// RPL = reads per lane: a tile is 64 * RPL reads consecutive in window order (slot q of lane l is
// read tile * 64 * RPL + q * 64 + l), so that the serial per-event work of the walk is shared by
// 64 * RPL reads.
template <int PASS, int RPL>
__global__ __launch_bounds__(64) void k_epp_sweep(EppSweepArgs a) {
    extern __shared__ uint32_t lds[];
    uint32_t* bm = lds;                       // [bm_words] positions listed by some read of the tile, relative to its window
    uint32_t* tab = lds + a.bm_words;         // [tab_rows][RPL][64] allele of read (q, lane) at window offset o: nibble o & 7
                                              //                of tab[((o >> 3) * RPL + q) * 64 + lane], 0 = not listed
    const uint32_t lane = threadIdx.x;
    const uint32_t job = blockIdx.x;
    // group of the job
    uint32_t g;
    {
        uint32_t lo = 0, hi = a.G;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.groups[mid].job0 <= job) lo = mid + 1; else hi = mid; }
        g = lo - 1;
    }
    const EppGroup gr = a.groups[g];
    const uint32_t local = job - gr.job0;
    const uint32_t tl = local % gr.ntiles, c = local / gr.ntiles;
    uint32_t sidx[RPL], rd[RPL], rs[RPL], span_r[RPL];
    bool have[RPL];
    uint32_t tws = 0xFFFFFFFFu, twe = 0;
#pragma unroll
    for (int q = 0; q < RPL; q++) {
        sidx[q] = ((gr.tile0 + tl) * RPL + q) * 64 + lane;
        have[q] = sidx[q] < a.R;
        rd[q] = have[q] ? a.order[sidx[q]] : 0;
        rs[q] = have[q] ? (uint32_t)a.start[rd[q]] : 0xFFFFFFFFu;
        const uint32_t re = have[q] ? (uint32_t)a.end[rd[q]] : 0;
        span_r[q] = have[q] ? re - rs[q] : 0;
        tws = min(tws, rs[q]);
        twe = max(twe, re);
    }
    // tile window, tile bitmap and the per-read allele table
    tws = wave_min_u32(tws);
    twe = wave_max_u32(twe);
    const uint32_t tspan = twe - tws;
    for (uint32_t i = lane; i < a.bm_words; i += 64) bm[i] = 0;
    for (uint32_t i = 0; i < a.tab_rows * RPL; i++) tab[i * 64 + lane] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPL; q++) {
        if (!have[q]) continue;
        const uint32_t off = a.read_off[rd[q]], k = a.read_off[rd[q] + 1] - off;
        for (uint32_t j = 0; j < k; j++) {
            const uint32_t w = a.read_word[off + j];
            const uint32_t o = (w & W_POS) - rs[q];
            if (o <= span_r[q]) {              // a position outside the read's window never changes its distance
                tab[((o >> 3) * RPL + q) * 64 + lane] |= ((w >> 24) & 15u) << ((o & 7) * 4);
                const uint32_t rel = (w & W_POS) - tws;
                atomicOr(&bm[rel >> 5], 1u << (rel & 31));
            }
        }
    }
    __syncthreads();

    // this job's piece of the window stream
    const uint32_t* sw = a.st_word + gr.soff;
    const uint32_t* sn = a.st_node + gr.soff;
    const uint32_t e0 = c * a.chunk_events;
    const uint32_t e1 = min(gr.n_events, e0 + a.chunk_events);
    const uint32_t p_end = e1 < gr.n_events ? sn[e1] : a.N;       // first node index that is not this chunk's
    uint32_t p_prev = (c == 0 || e0 >= gr.n_events) ? 0u : sn[e0];
    if (c > 0 && e0 >= gr.n_events) p_prev = a.N;                 // empty trailing chunk: no nodes
    const size_t row0 = (size_t)job * RPL * 64 + lane;            // + q * 64

    int D[RPL], mn[RPL];
    uint32_t cnt[RPL];
    // pass 2 state: per read its minimum, score delta, degree, bin and EPP-list cursor;
    // wave-uniform the reads whose range is open and the sums of the last flip
    int best[RPL];
    long long dfx[RPL];
    int deg[RPL];
    uint32_t bucket[RPL];
    unsigned long long list_at[RPL];
    unsigned long long open_mask[RPL], list_mask[RPL], c_t[RPL], c_on[RPL];
    long long c_vs = 0;
    int c_nb = 0, c_cs[2] = {0, 0};
    uint32_t c_b[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < RPL; q++) {
        D[q] = 0; mn[q] = INT_INF; cnt[q] = 0;
        best[q] = INT_INF;                     // slots without a read never match
        dfx[q] = 0; deg[q] = 0; bucket[q] = 0; list_at[q] = 0;
        open_mask[q] = 0; list_mask[q] = 0; c_t[q] = 0; c_on[q] = 0;
        if (PASS == 2) {
            D[q] = a.part_net[row0 + q * 64];  // distance at the chunk's start
            bool keep = false;
            if (have[q]) {
                best[q] = a.best[sidx[q]];
                dfx[q] = a.delta_fx[sidx[q]];
                deg[q] = a.degree[rd[q]];
                bucket[q] = min((uint32_t)a.start[rd[q]] / a.bin_size, EPP_BINS - 1);
                const uint64_t eb = a.epp_base[rd[q]];
                keep = eb != ~0ull;
                list_at[q] = eb + a.part_cnt[row0 + q * 64];
            }
            list_mask[q] = __ballot(keep);
        }
    }

    // flips of pass 2: the reads in `tmask` start (those also in `onmask`) or stop matching at node p
    auto flip = [&](const unsigned long long (&tmask)[RPL], const unsigned long long (&onmask)[RPL], uint32_t p) {
        // A mutation's enter and exit events usually flip the same reads in opposite directions
        // (always when its node is a leaf): the sums of the previous flip are reused.
        bool same_t = true, same_dir = true, opp_dir = true;
#pragma unroll
        for (int q = 0; q < RPL; q++) {
            same_t = same_t && tmask[q] == c_t[q];
            same_dir = same_dir && onmask[q] == c_on[q];
            opp_dir = opp_dir && onmask[q] == (c_t[q] & ~c_on[q]);
        }
        if (!(same_t && (same_dir || opp_dir)) || c_nb > 2) {
            long long v = 0;
            bool t_l[RPL], on_l[RPL];
#pragma unroll
            for (int q = 0; q < RPL; q++) {
                t_l[q] = (tmask[q] >> lane) & 1ull;
                on_l[q] = (onmask[q] >> lane) & 1ull;
                v += t_l[q] ? (on_l[q] ? dfx[q] : -dfx[q]) : 0;
            }
            const long long vs = wave_sum_i64(v);
            c_vs = ((long long)__builtin_amdgcn_readfirstlane((int)(vs >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vs);
            c_nb = 0;
            if (a.diff_cnt) {
                unsigned long long pending[RPL];
#pragma unroll
                for (int q = 0; q < RPL; q++) pending[q] = tmask[q];
                while (true) {
                    // bin of the first read still pending
                    uint32_t bb = 0xFFFFFFFFu;
#pragma unroll
                    for (int q = RPL - 1; q >= 0; q--)
                        if (pending[q]) bb = (uint32_t)__builtin_amdgcn_readlane((int)bucket[q], __builtin_ctzll(pending[q]));
                    if (bb == 0xFFFFFFFFu) break;
                    int cv = 0;
#pragma unroll
                    for (int q = 0; q < RPL; q++) {
                        const bool same = t_l[q] && bucket[q] == bb;
                        cv += same ? (on_l[q] ? deg[q] : -deg[q]) : 0;
                        pending[q] &= ~__ballot(same);
                    }
                    const int cs = __builtin_amdgcn_readfirstlane(wave_sum_i32(cv));
                    if (c_nb < 2) { c_b[c_nb] = bb; c_cs[c_nb] = cs; }
                    else if (lane == 0) atomicAdd(&a.diff_cnt[(size_t)p * EPP_BINS + bb], cs);
                    c_nb++;
                }
            }
#pragma unroll
            for (int q = 0; q < RPL; q++) { c_t[q] = tmask[q]; c_on[q] = onmask[q]; }
        } else if (!same_dir) {
            c_vs = -c_vs;
            c_cs[0] = -c_cs[0];
            c_cs[1] = -c_cs[1];
#pragma unroll
            for (int q = 0; q < RPL; q++) c_on[q] = onmask[q];
        }
        if (lane == 0) {
            atomicAdd(&a.diff_score[p], (unsigned long long)c_vs);
            if (a.diff_cnt) {
                if (c_nb > 0) atomicAdd(&a.diff_cnt[(size_t)p * EPP_BINS + c_b[0]], c_cs[0]);
                if (c_nb > 1) atomicAdd(&a.diff_cnt[(size_t)p * EPP_BINS + c_b[1]], c_cs[1]);
            }
        }
    };

    // the nodes [p_prev, p) carry the current distances
    auto credit = [&](uint32_t p) {
        const uint32_t gap = p - p_prev;
        if (gap == 0) return;
        if (PASS == 1) {
#pragma unroll
            for (int q = 0; q < RPL; q++) {
                cnt[q] = D[q] < mn[q] ? 0u : cnt[q];
                mn[q] = min(mn[q], D[q]);
                cnt[q] += D[q] == mn[q] ? gap : 0u;
            }
        } else {
            unsigned long long match_mask[RPL], tmask[RPL], onmask[RPL];
            unsigned long long any_t = 0, any_l = 0;
#pragma unroll
            for (int q = 0; q < RPL; q++) {
                match_mask[q] = __ballot(D[q] == best[q]);
                tmask[q] = match_mask[q] ^ open_mask[q];
                onmask[q] = tmask[q] & match_mask[q];
                any_t |= tmask[q];
                any_l |= match_mask[q] & list_mask[q];
            }
            if (any_t) {
                flip(tmask, onmask, p_prev);
#pragma unroll
                for (int q = 0; q < RPL; q++) open_mask[q] = match_mask[q];
            }
            if (any_l) {
#pragma unroll
                for (int q = 0; q < RPL; q++) {
                    if (((match_mask[q] & list_mask[q]) >> lane) & 1ull) {
                        for (uint32_t x = 0; x < gap; x++) a.epp_nodes[list_at[q] + x] = p_prev + x;
                        list_at[q] += gap;
                    }
                }
            }
        }
        p_prev = p;
    };

    // The walk is serial in the events but most of what an event needs does not depend on the
    // read: it is computed lane = event for 64 events at a time, the events inside the tile's
    // window are packed to the low lanes, and the serial loop only broadcasts three registers
    // per event.
    for (uint32_t b0 = e0; b0 < e1; b0 += 64) {
        const uint32_t i = b0 + lane;
        const uint32_t w = i < e1 ? sw[i] : PAD_WORD;
        const uint32_t nd = i < e1 ? sn[i] : 0;
        const uint32_t wpos = w & W_POS;
        const uint32_t rel = wpos - tws;
        const bool in_tile = rel <= tspan;
        const bool listed = in_tile && ((bm[rel >> 5] >> (rel & 31)) & 1u);
        // distance change for a read that does not list the position (it shows the reference base);
        // bits 8.. keep what a read that lists it needs: mut, genotype above, exit flag
        uint32_t info;
        {
            const uint32_t refm = 1u << ((w >> 20) & 3u);
            const uint32_t par = (w >> 22) & 15u, mut = (w >> 26) & 15u;
            const uint32_t pare = par ? par : refm;            // genotype above the mutation
            int dref = (int)(mut != refm) - (int)(pare != refm);
            if (w & W_EXIT_BIT) dref = -dref;
            info = (uint32_t)(dref & 0xFF) | (mut << 8) | (pare << 12) | (((w >> 30) & 1u) << 16);
        }
        const unsigned long long m_in = __ballot(in_tile);
        if (m_in == 0) continue;
        const uint32_t n_in = (uint32_t)__popcll(m_in);
        const int dst = (int)(in_tile ? (uint32_t)__popcll(m_in & ((1ull << lane) - 1ull)) : 63u) << 2;
        const uint32_t c_pos = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)wpos);
        const uint32_t c_info = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)info);
        const uint32_t c_nd = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)nd);
        const bool c_listed = __builtin_amdgcn_ds_permute(dst, (int)listed) != 0 && lane < n_in;
        const unsigned long long m_hit = __ballot(c_listed);
        for (uint32_t j = 0; j < n_in; j++) {
            const uint32_t ep = (uint32_t)__builtin_amdgcn_readlane((int)c_nd, (int)j);
            credit(ep);
            const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)c_pos, (int)j);
            const uint32_t ei = (uint32_t)__builtin_amdgcn_readlane((int)c_info, (int)j);
            const int dref = (int)(int8_t)(ei & 0xFF);
            if ((m_hit >> j) & 1ull) {
                // some read of the tile lists this position: each read looks its own allele up
                const uint32_t mut = (ei >> 8) & 15u, pare = (ei >> 12) & 15u;
#pragma unroll
                for (int q = 0; q < RPL; q++) {
                    const uint32_t o = pos - rs[q];
                    const bool in_win = o <= span_r[q];        // only inside the read's own window (:56,60)
                    const uint32_t oo = in_win ? o : 0;
                    const uint32_t al = (tab[((oo >> 3) * RPL + q) * 64 + lane] >> ((oo & 7) * 4)) & 15u;
                    int dl = (int)(mut != al) - (int)(pare != al);
                    if ((ei >> 16) & 1u) dl = -dl;
                    dl = al == 15u ? 0 : dl;                   // N matches anything (:68)
                    const int d = al ? dl : dref;
                    D[q] += in_win ? d : 0;
                }
            } else {
#pragma unroll
                for (int q = 0; q < RPL; q++) D[q] += (pos - rs[q] <= span_r[q]) ? dref : 0;
            }
        }
    }
    credit(p_end);
    if (PASS == 1) {
#pragma unroll
        for (int q = 0; q < RPL; q++) {
            a.part_min[row0 + q * 64] = mn[q];
            a.part_cnt[row0 + q * 64] = cnt[q];
            a.part_net[row0 + q * 64] = D[q];
        }
    } else {
        unsigned long long any_o = 0, zero[RPL];
#pragma unroll
        for (int q = 0; q < RPL; q++) { any_o |= open_mask[q]; zero[q] = 0; }
        if (any_o) flip(open_mask, zero, p_end);   // ranges still open at the chunk's end stop at its last node
    }
}

// one thread per sorted read
__global__ __launch_bounds__(256) void k_epp_combine(EppSweepArgs a, uint32_t tiles_per_group, uint32_t rpl) {
    const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= a.R) return;
    const uint32_t t = sidx / (64 * rpl), q = (sidx / 64) % rpl, lane = sidx & 63;
    const EppGroup gr = a.groups[t / tiles_per_group];
    const uint32_t tl = t - gr.tile0;
    const uint32_t r = a.order[sidx];
    // root_mutations: the listed non-N positions (initial_filter.cpp:118-123)
    int D = 0;
    for (uint32_t j = a.read_off[r]; j < a.read_off[r + 1]; j++) D += ((a.read_word[j] >> 24) & 15u) != 15u;
    int best = INT_INF;
    uint32_t mult = 0;
    for (uint32_t c = 0; c < gr.nchunks; c++) {
        const size_t row = (((size_t)gr.job0 + (size_t)c * gr.ntiles + tl) * rpl + q) * 64 + lane;
        const int mn = a.part_min[row];
        if (mn != INT_INF) {
            const int v = D + mn;
            if (v < best) { best = v; mult = a.part_cnt[row]; }
            else if (v == best) mult += a.part_cnt[row];
        }
        const int net = a.part_net[row];
        a.part_net[row] = D;
        D += net;
    }
    uint32_t cursor = 0;
    for (uint32_t c = 0; c < gr.nchunks; c++) {
        const size_t row = (((size_t)gr.job0 + (size_t)c * gr.ntiles + tl) * rpl + q) * 64 + lane;
        const int mn = a.part_min[row];
        const uint32_t cn = a.part_cnt[row];
        a.part_cnt[row] = cursor;
        if (mn != INT_INF && a.part_net[row] + mn == best) cursor += cn;
    }
    a.best[sidx] = best;
    a.mult[sidx] = mult;
    // node_score, initial_filter.hpp:54-57
    const double delta = (double)a.degree[r] / (double)((long long)(1 + best) * (long long)mult);
    a.delta_fx[sidx] = __double2ll_rn(delta * a.fx_scale);
}

// ---------------------------------------------------------------------------------------
// prefix sums of the difference arrays
// ---------------------------------------------------------------------------------------
constexpr uint32_t SCAN_TILE = 2048;   // nodes per workgroup (256 threads x 8)

__global__ __launch_bounds__(256) void k_epp_scan_sums(uint32_t N, const unsigned long long* __restrict__ diff_score,
                                                       const int* __restrict__ diff_cnt,
                                                       long long* __restrict__ blk_score, int* __restrict__ blk_cnt) {
    __shared__ long long s64[256];
    __shared__ int s32[EPP_BINS];
    const uint32_t n0 = blockIdx.x * SCAN_TILE, n1 = min(N, n0 + SCAN_TILE);
    long long acc = 0;
    for (uint32_t i = n0 + threadIdx.x; i < n1; i += 256) acc += (long long)diff_score[i];
    s64[threadIdx.x] = acc;
    if (threadIdx.x < EPP_BINS) s32[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        if (threadIdx.x < d) s64[threadIdx.x] += s64[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) blk_score[blockIdx.x] = s64[0];
    if (diff_cnt) {
        // thread t sums bin (t % 50) over the nodes t / 50, t / 50 + 5, ...
        const uint32_t bin = threadIdx.x % EPP_BINS, sub = threadIdx.x / EPP_BINS;
        if (sub < 5) {
            int a = 0;
            for (uint32_t i = n0 + sub; i < n1; i += 5) a += diff_cnt[(size_t)i * EPP_BINS + bin];
            atomicAdd(&s32[bin], a);
        }
        __syncthreads();
        if (threadIdx.x < EPP_BINS) blk_cnt[(size_t)blockIdx.x * EPP_BINS + threadIdx.x] = s32[threadIdx.x];
    }
}

// single workgroup: exclusive scan of the block sums (in place)
__global__ __launch_bounds__(256) void k_epp_scan_blocks(uint32_t nblk, long long* __restrict__ blk_score,
                                                         int* __restrict__ blk_cnt, bool with_cnt) {
    __shared__ long long s[256];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nblk; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const long long v = i < nblk ? blk_score[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 256; d <<= 1) {
            const long long t = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        const long long c0 = carry;
        if (i < nblk) blk_score[i] = c0 + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = c0 + s[255];
        __syncthreads();
    }
    if (with_cnt && threadIdx.x < EPP_BINS) {
        int run = 0;
        for (uint32_t b = 0; b < nblk; b++) {
            const int v = blk_cnt[(size_t)b * EPP_BINS + threadIdx.x];
            blk_cnt[(size_t)b * EPP_BINS + threadIdx.x] = run;
            run += v;
        }
    }
}

__global__ __launch_bounds__(256) void k_epp_scan_apply(uint32_t N, const unsigned long long* __restrict__ diff_score,
                                                        const int* __restrict__ diff_cnt,
                                                        const long long* __restrict__ blk_score,
                                                        const int* __restrict__ blk_cnt, double inv_scale,
                                                        double* __restrict__ score, const int* __restrict__ true_counts,
                                                        int* __restrict__ counts, double* __restrict__ divergence) {
    __shared__ long long s[256];
    __shared__ int run[EPP_BINS];
    __shared__ int tc[EPP_BINS];
    __shared__ int active;
    const uint32_t n0 = blockIdx.x * SCAN_TILE, n1 = min(N, n0 + SCAN_TILE);
    // score: thread t owns 8 consecutive nodes
    {
        long long v[8];
        long long acc = 0;
        const uint32_t i0 = n0 + threadIdx.x * 8;
#pragma unroll
        for (int q = 0; q < 8; q++) { v[q] = (i0 + q < n1) ? (long long)diff_score[i0 + q] : 0; acc += v[q]; }
        s[threadIdx.x] = acc;
        __syncthreads();
        for (uint32_t d = 1; d < 256; d <<= 1) {
            const long long t = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        long long runv = blk_score[blockIdx.x] + s[threadIdx.x] - acc;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            runv += v[q];
            if (i0 + q < n1) score[i0 + q] = (double)runv * inv_scale;
        }
    }
    if (!diff_cnt) return;
    if (threadIdx.x < EPP_BINS) {
        run[threadIdx.x] = blk_cnt[(size_t)blockIdx.x * EPP_BINS + threadIdx.x];
        tc[threadIdx.x] = true_counts[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        int act = 0;
        for (uint32_t b = 0; b < EPP_BINS; b++) act += true_counts[b] != 0;
        active = act;
    }
    __syncthreads();
    // per-bin running sums: thread b < 50 walks its bin over the tile's nodes (the loads of the 50
    // threads are one contiguous 200-byte row per node)
    if (threadIdx.x < EPP_BINS) {
        int rv = run[threadIdx.x];
        for (uint32_t i = n0; i < n1; i++) {
            rv += diff_cnt[(size_t)i * EPP_BINS + threadIdx.x];
            if (counts) counts[(size_t)i * EPP_BINS + threadIdx.x] = rv;
            if (divergence) {
                // wepp_filter::cartesian_map, initial_filter.cpp:224-233
                const double prop = (double)rv / (double)tc[threadIdx.x];
                const unsigned long long over = __ballot(prop > 0.5 / 100);
                if (threadIdx.x == 0) divergence[i] = (double)__popcll(over) / (double)active;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
hipError_t launch_epp_select_count(const uint32_t* ev_word, uint64_t n_events, const EppGroup* groups,
                                   const uint32_t* we_max, uint32_t G, uint32_t nblk, uint32_t* cnt,
                                   hipStream_t stream) {
    if (nblk == 0) return hipSuccess;
    hipLaunchKernelGGL(k_epp_select_count, dim3(nblk), dim3(64), 0, stream, ev_word, n_events, groups, we_max, G, nblk, cnt);
    return hipGetLastError();
}
hipError_t launch_epp_select_scan(uint32_t* cnt, uint32_t G, uint32_t nblk, uint32_t* totals, hipStream_t stream) {
    hipLaunchKernelGGL(k_epp_select_scan, dim3(G), dim3(256), 0, stream, cnt, nblk, totals);
    return hipGetLastError();
}
hipError_t launch_epp_select_scatter(const uint32_t* ev_word, const uint32_t* ev_node, uint64_t n_events,
                                     const EppGroup* groups, const uint32_t* we_max, uint32_t G, uint32_t nblk,
                                     const uint32_t* cnt, uint32_t* st_word, uint32_t* st_node, hipStream_t stream) {
    if (nblk == 0) return hipSuccess;
    hipLaunchKernelGGL(k_epp_select_scatter, dim3(nblk), dim3(64), 0, stream, ev_word, ev_node, n_events, groups, we_max,
                       G, nblk, cnt, st_word, st_node);
    return hipGetLastError();
}
template <int PASS, int RPL>
static hipError_t launch_sweep_variant(const EppSweepArgs& a, uint32_t lds_bytes, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)k_epp_sweep<PASS, RPL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_epp_sweep<PASS, RPL>), dim3(a.n_jobs), dim3(64), lds_bytes, stream, a);
    return hipGetLastError();
}
hipError_t launch_epp_sweep(const EppSweepArgs& a, int pass, uint32_t rpl, uint32_t lds_bytes, hipStream_t stream) {
    if (a.n_jobs == 0) return hipSuccess;
    if (rpl == 1) return pass == 1 ? launch_sweep_variant<1, 1>(a, lds_bytes, stream) : launch_sweep_variant<2, 1>(a, lds_bytes, stream);
    if (rpl == 4) return pass == 1 ? launch_sweep_variant<1, 4>(a, lds_bytes, stream) : launch_sweep_variant<2, 4>(a, lds_bytes, stream);
    return hipErrorInvalidValue;
}
hipError_t launch_epp_combine(const EppSweepArgs& a, uint32_t tiles_per_group, uint32_t rpl, hipStream_t stream) {
    if (a.R == 0) return hipSuccess;
    hipLaunchKernelGGL(k_epp_combine, dim3((a.R + 255) / 256), dim3(256), 0, stream, a, tiles_per_group, rpl);
    return hipGetLastError();
}
size_t epp_finish_scratch_bytes(uint32_t N) {
    const size_t nblk = ((size_t)N + SCAN_TILE - 1) / SCAN_TILE;
    return nblk * 8 + nblk * EPP_BINS * 4 + 64;
}
hipError_t launch_epp_finish(uint32_t N, const unsigned long long* diff_score, double inv_scale, double* score,
                             const int* diff_cnt, const int* true_counts, int* counts, double* divergence,
                             void* scratch, hipStream_t stream) {
    const uint32_t nblk = (N + SCAN_TILE - 1) / SCAN_TILE;
    long long* blk_score = (long long*)scratch;
    int* blk_cnt = (int*)((char*)scratch + (size_t)nblk * 8);
    hipLaunchKernelGGL(k_epp_scan_sums, dim3(nblk), dim3(256), 0, stream, N, diff_score, diff_cnt, blk_score, blk_cnt);
    hipLaunchKernelGGL(k_epp_scan_blocks, dim3(1), dim3(256), 0, stream, nblk, blk_score, blk_cnt, diff_cnt != nullptr);
    hipLaunchKernelGGL(k_epp_scan_apply, dim3(nblk), dim3(256), 0, stream, N, diff_score, diff_cnt, blk_score, blk_cnt,
                       inv_scale, score, true_counts, counts, divergence);
    return hipGetLastError();
}

}  // namespace wepp
