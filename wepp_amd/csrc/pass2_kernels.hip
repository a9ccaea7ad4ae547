// pass2_kernels.hip -- what the reference's second pass and its -p mode produce beyond (node, score, count): all
// per-node scores / all optimal nodes (k_scores), imputed mutations of the chosen node (k_imputed), excess mutations
// of (sample, node) pairs (k_excess).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_mat.hpp"
#include "place_dev.hpp"

namespace wepp {

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

}  // namespace wepp
