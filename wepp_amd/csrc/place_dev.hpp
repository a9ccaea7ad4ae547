// place_dev.hpp -- device-side helpers shared by the placement kernels (route_kernels.hip, sweep_kernels.hip,
// walk_kernels.hip, pass2_kernels.hip, seed_kernels.hip): the packed word fields, the closed form of mapper2_body's
// per-position costs, wave-wide reductions on DPP row shifts, and the per-read result record.
//
// What is computed: for every read, exactly what the two passes of the reference's per-sample loop leave behind
// (src/usher_common.cpp:386-446, each iteration being mapper2_body, src/usher_mapper.cpp:168-506): the minimum
// parsimony score over eligible nodes, the number of eligible nodes attaining it, and the winner under the
// (num_leaves, BFS index) tie-break.  The score of node n for read S is
//     score(n) = base(n) + c_S(parent(n)) + adj_S(n)
// where base(n) is read-independent, c_S is the read-dependent correction of the parent genotype and adj_S(n) is
// non-zero only when n itself mutates a position listed in S.  c_S changes only at "events": entering / leaving
// the subtree of a node that mutates a position of S.  Integer work only: no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_mat.hpp"

namespace wepp {

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

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

}  // namespace

}  // namespace wepp
