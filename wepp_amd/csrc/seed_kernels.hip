// seed_kernels.hip -- whole-genome samples: per-chunk signatures rule out nearly all of the tree.
//
// The samples usher_common is fed come from read_vcf (src/mutation_annotated_tree.cpp:2033-2130): one entry per VCF
// row at which the sample differs from the reference or is missing -- tens to hundreds of positions all over the
// genome.  Such a sample fits no genome window, lists too many positions to walk, and its tree-wide bound
// theta = score(root) + |S| admits the whole tree; a tile sweep of the whole-tree stream cost 47 us per sample.
//
// The bound used here.  mapper2_body scores the sample at node n against a genotype E(n): the mutations on the path
// above n plus n's own mutations the sample shares (src/usher_mapper.cpp:191-287), and charges one for every
// non-missing entry whose alleles E(n) does not offer at the entry's position (:293-389; with no mutation on the
// path the entry is compared with its own reference base, :302-305,342).  Call an entry HARD when it is not missing
// and its alleles exclude its reference base: it costs one unless some mutation on the path (or of n) carries a
// compatible allele at its position.  Every mutation E(n) can hold, for a node n of a chunk [a, b) of the DFS order,
// sits on a node of the chunk or on an ancestor of a -- the chunk's SIGNATURE (flatmat.hpp: seed_sig) holds the OR
// of their alleles per position.  With T a set of hard entries and H_T(C) the number of them the signature of chunk
// C could serve,
//     score(n) >= |T| - H_T(C)        for every node n of C
// (the other terms of the score -- the remaining entries, the path's mutations the sample does not list,
// :394-446 -- are non-negative).  The root always competes, and every evaluated chunk yields achieved scores: with
// B = the best score found so far (root score + 1 before anything is found), a chunk with |T| - H_T(C) > B holds no
// node that wins or ties, and is never looked at.  Exact: a chunk is skipped only on that inequality; which chunks
// are tried first only decides how soon B is small.
//
// One workgroup per sample: (1) its words and a position bitmap go to LDS; (2) H_T of every chunk -- for every hard
// entry one coalesced pass over its position's row of nibbles, eight chunks per dword, SWAR counters in registers;
// (3) best first: the chunks with the largest H_T, then level by level (H_T = h, h - 1, ...) while |T| - h <= B, the
// waves sharing a level's chunks; (4) a chunk's nodes are scored exactly from the stream's checkpoint at the chunk
// start: the events of 32 blocks are scanned for the sample's positions (lane = event, eight loads in flight), the hit
// events listed in LDS, then every block takes its static summary (no hit) or is evaluated node by node (lane = node);
// (5) the partials are combined and the result record is written (place_dev.hpp: emit_result).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_mat.hpp"
#include "place_dev.hpp"

namespace wepp {

namespace {
constexpr uint32_t SEED_WAVES = SEED_THREADS / 64;
constexpr uint32_t SEED_SHARED_WORDS = 16 + 3 * SEED_WAVES;   // bound, counters, largest count, per-wave partials
constexpr uint32_t SEED_HIT_CAP = 96;                         // hit events of a group of blocks a wave lists before it evaluates them
// LDS (dwords): position bitmap [bm_words] | the sample's words [ent_cap] | hard entries [SEED_MAX_HARD + 1] |
// per-chunk counters, one byte each, [chunks rounded up to 64] | shared scalars | per wave: hit list [3 * SEED_HIT_CAP]
__host__ __device__ inline uint32_t seed_h_words(uint32_t chunks) { return ((chunks + 63) & ~63u) / 4; }
}  // namespace

// A sample that still faces a level of SEED_HEAVY_LEVEL_MIN chunks or more (it sits far from every node of the tree: its
// best score hardly beats the levels' bounds, thousands of chunks can tie) is handed over with what it has found so far
// (SeedHeavy: a record in global memory): HEAVY = the second pass, SEED_HEAVY_PARTS workgroups per such sample, each
// taking every SEED_HEAVY_PARTS-th slab of 64 chunks of the remaining levels, the best score shared through the
// record; k_seed_heavy_finalize combines their partials.  (One workgroup took 8.8 ms for a sample with 7 013 chunks to
// evaluate: the tail of a 13 ms step of 20 000 samples.)
struct SeedHeavy {
    uint32_t count;                                   // deferred samples (k_seed appends, the finalize kernel clears)
    uint32_t pad[3];
    struct Rec { uint32_t read; int level; int best; int score; uint32_t rank, cnt; } rec[SEED_HEAVY_CAP];   // (score, rank, cnt): what the first pass found; best: the bound the parts share (atomicMin)
    struct Part { int score; uint32_t rank, cnt; } part[SEED_HEAVY_CAP * SEED_HEAVY_PARTS];
};

template <bool HEAVY>
__global__ __launch_bounds__(SEED_THREADS) void k_seed(DevMAT m, DevStream full, const uint32_t* __restrict__ list, uint32_t n_list,
                                                        uint32_t ent_cap, const uint32_t* __restrict__ read_off,
                                                        const uint32_t* __restrict__ read_word, const int32_t* __restrict__ root_score,
                                                        uint32_t* __restrict__ best_bfs_j, int32_t* __restrict__ score_out,
                                                        uint32_t* __restrict__ num_best, uint32_t* __restrict__ flags,
                                                        unsigned long long* __restrict__ work_counter, SeedHeavy* __restrict__ heavy) {
    extern __shared__ uint32_t lds[];
    uint32_t* bitmap = lds;
    uint32_t* S = bitmap + m.bm_words;
    uint32_t* hard = S + ent_cap;
    uint32_t* H32 = hard + (SEED_MAX_HARD + 1);
    const uint8_t* H = reinterpret_cast<const uint8_t*>(H32);
    int* sh = reinterpret_cast<int*>(H32 + seed_h_words(m.seed_chunks));
    uint32_t* hitbuf = reinterpret_cast<uint32_t*>(sh + SEED_SHARED_WORDS);
    // sh[0] bound B, [1] hard entries with a signature row, [2] hard entries in all (= c_S of an empty path),
    // [3] hard entries beyond the tree's last mutated position, [4] largest chunk count, [5] chunks evaluated, [6] the bound a level
    // is entered with, [8 + 3 w ..] partial of wave w
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // HEAVY: workgroup = (deferred sample, part); the grid is sized for the record table, the count is on the device
    const uint32_t hv_i = HEAVY ? blockIdx.x / SEED_HEAVY_PARTS : 0u, hv_p = HEAVY ? blockIdx.x % SEED_HEAVY_PARTS : 0u;
    if (HEAVY && hv_i >= min(heavy->count, SEED_HEAVY_CAP)) return;
    const uint32_t r = HEAVY ? heavy->rec[hv_i].read : list[blockIdx.x];
    const uint32_t so = read_off[r];
    const uint32_t k = min(read_off[r + 1] - so, ent_cap);
    const uint32_t bm_mask = m.bm_words - 1;
    const uint32_t nch = m.seed_chunks, row_words = m.seed_row_words;

    for (uint32_t i = tid; i < m.bm_words; i += SEED_THREADS) bitmap[i] = 0;
    if (tid < SEED_SHARED_WORDS) sh[tid] = 0;
    __syncthreads();
    if (tid == 0) sh[0] = HEAVY ? heavy->rec[hv_i].best : root_score[r] + 1;        // the root always competes: nothing worse can win or tie
    // ---- (1) the sample ----
    for (uint32_t j = tid; j < k; j += SEED_THREADS) {
        const uint32_t w = read_word[so + j];
        const uint32_t p = w_pos(w);
        S[j] = w;
        const bool in_tree = p <= m.max_pos;
        if (in_tree) atomicOr(&bitmap[(p >> 5) & bm_mask], 1u << (p & 31));
        if (!rw_missing(w) && (rw_mut(w) & rw_ref(w)) == 0) {
            atomicAdd(&sh[2], 1);
            if (in_tree) {
                const uint32_t slot = (uint32_t)atomicAdd(&sh[1], 1);
                if (slot < SEED_MAX_HARD) hard[slot] = w;        // (any subset of the hard entries gives a valid bound)
            } else atomicAdd(&sh[3], 1);                         // no mutation there anywhere: it costs one on every node
        }
    }
    __syncthreads();
    const uint32_t nh = min((uint32_t)sh[1], SEED_MAX_HARD);
    const int sT = (int)nh + sh[3];                   // |T|
    const int c0 = sh[2];                             // c_S with no mutation on the path (usher_mapper.cpp:302-305,342)
    auto bit = [&](uint32_t pos) -> bool { return (bitmap[(pos >> 5) & bm_mask] >> (pos & 31)) & 1u; };

    // ---- (2) H_T of every chunk: one pass over the signature row of every hard entry ----
    for (uint32_t d = tid; d < seed_h_words(nch) / 2; d += SEED_THREADS) {
        uint32_t acc4 = 0, lo = 0, hi = 0, since = 0;
        if (d < row_words) {
            auto add = [&](uint32_t x, uint32_t w) {
                uint32_t y = x & (rw_mut(w) * 0x11111111u);
                y |= y >> 1;
                y |= y >> 2;
                acc4 += y & 0x11111111u;
                if (++since == 15) { lo += acc4 & 0x0F0F0F0Fu; hi += (acc4 >> 4) & 0x0F0F0F0Fu; acc4 = 0; since = 0; }
            };
            uint32_t j = 0;
            for (; j + 4 <= nh; j += 4) {            // four rows' loads in flight together
                const uint32_t w0 = hard[j], w1 = hard[j + 1], w2 = hard[j + 2], w3 = hard[j + 3];
                const uint32_t x0 = m.seed_sig[(size_t)w_pos(w0) * row_words + d], x1 = m.seed_sig[(size_t)w_pos(w1) * row_words + d];
                const uint32_t x2 = m.seed_sig[(size_t)w_pos(w2) * row_words + d], x3 = m.seed_sig[(size_t)w_pos(w3) * row_words + d];
                add(x0, w0); add(x1, w1); add(x2, w2); add(x3, w3);
            }
            for (; j < nh; j++) { const uint32_t w0 = hard[j]; add(m.seed_sig[(size_t)w_pos(w0) * row_words + d], w0); }
            lo += acc4 & 0x0F0F0F0Fu;
            hi += (acc4 >> 4) & 0x0F0F0F0Fu;
        }
        // byte i of lo = chunk 8 d + 2 i, of hi = chunk 8 d + 2 i + 1: interleave into chunk order
        H32[2 * d] = (lo & 0xFFu) | ((hi & 0xFFu) << 8) | ((lo & 0xFF00u) << 8) | ((hi & 0xFF00u) << 16);
        H32[2 * d + 1] = ((lo >> 16) & 0xFFu) | (((hi >> 16) & 0xFFu) << 8) | ((lo >> 24) << 16) | ((hi >> 24) << 24);
    }
    __syncthreads();
    // ---- the largest count: the levels below are tried best first ----
    {
        uint32_t hm = 0;
        for (uint32_t c4 = tid; c4 < (nch + 3) / 4; c4 += SEED_THREADS) {
            const uint32_t x = H32[c4];
#pragma unroll
            for (uint32_t q = 0; q < 4; q++)
                if (4 * c4 + q < nch) hm = max(hm, (x >> (8 * q)) & 0xFFu);
        }
        hm = ~wave_min_u32(~hm);
        if (lane == 0) atomicMax(&sh[4], (int)hm);
    }
    __syncthreads();
    const int hmax = sh[4];
    // chunks per level (first pass: a level of SEED_HEAVY_LEVEL_MIN chunks or more is where a sample is handed over)
    uint32_t* lvl = hitbuf;                  // [256] (the waves' hit lists are not in use yet)
    if (!HEAVY && heavy) {
        for (uint32_t i = tid; i < 256; i += SEED_THREADS) lvl[i] = 0;
        __syncthreads();
        for (uint32_t c = tid; c < nch; c += SEED_THREADS) atomicAdd(&lvl[H[c]], 1u);
        __syncthreads();
    }
    // (level -> count, read by every thread before the hit lists overwrite the table)
    int defer_level = -1;
    if (!HEAVY && heavy)
        for (int h = hmax; h >= 0; h--)
            if (lvl[h] >= SEED_HEAVY_LEVEL_MIN) { defer_level = h; break; }
    __syncthreads();

    // ---- exact evaluation of one chunk by one wave ----
    int bs = HEAVY ? heavy->rec[hv_i].best : root_score[r] + 1;
    uint32_t br = 0xFFFFFFFFu, cnt = 0;
    uint32_t wbytes = 0, wchunks = 0;
    uint32_t* hits = hitbuf + wv * (3 * SEED_HIT_CAP);    // this wave's hit list: event index, tree word, sample word
    auto take = [&](int sc, uint32_t rk, uint32_t kk) {
        if (sc < bs) { bs = sc; br = rk; cnt = kk; }
        else if (sc == bs) { cnt += kk; br = min(br, rk); }
    };
    // A block that holds events of the sample is evaluated node by node (lane = node): first the hit events are
    // applied -- `nhb` of them from this wave's list, hits[h0 ..) -- to the lanes' running sums (what the events add to
    // c_S in front of the lane's node, the node's own adjustments), then the nodes are scored.
    struct NodeAcc { int cadd = 0, adj = 0, dcom = 0, net = 0; bool touched = false; };
    auto apply_hits = [&](uint32_t node0, uint32_t h0, uint32_t nhb, NodeAcc& a) {
        for (uint32_t q0 = 0; q0 < nhb; q0 += 64) {
            const uint32_t nq = min(64u, nhb - q0);
            uint32_t mt = 0, wq = 0, sq = 0;
            if (lane < nq) {
                const uint32_t* hrec = hits + 3 * (h0 + q0 + lane);
                mt = (uint32_t)full.ev_meta[hrec[0]];       // the events' node offsets: one gather for all of them
                wq = hrec[1];
                sq = hrec[2];
            }
            for (uint32_t q = 0; q < nq; q++) {
                const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)wq, (int)q);
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)mt, (int)q) & EV_OFF_MASK_DEV;
                const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)sq, (int)q);
                const int delta = enter_delta(wl, sl);
                if (wl & W_EXIT_DEV) {
                    a.cadd += (lane >= o) ? -delta : 0;     // the subtree that carried wl ended just before node o
                    a.net -= delta;
                } else {
                    if (!(wl & W_LEAF_DEV)) {
                        // descendants of node o see the new allele; the root also scores itself with its own
                        // mutations applied (usher_mapper.cpp:266-271)
                        const bool is_root = (node0 + o) == 0;
                        a.cadd += (lane > o || (is_root && lane == o)) ? delta : 0;
                        a.net += delta;
                    }
                    if (lane == o) {
                        a.touched = true;
                        own_adjust(wl, sl, a.adj, a.dcom);
                    }
                }
            }
        }
        wbytes += nhb;
    };
    auto score_nodes = [&](uint32_t node0, uint32_t nn, int cc, const NodeAcc& a) {
        const bool nvalid = lane < nn;
        const int64_t key = nvalid ? full.nkey[node0 + lane] : 0;
        const uint32_t st = nvalid ? full.nstat[node0 + lane] : 0;
        wbytes += nn * 12;
        const int base = (int)(key >> 32);
        const uint32_t rank = (uint32_t)(key & 0xFFFFFFFFll);
        const uint32_t nmut = st & NS_CNT_MASK_DEV, ncom0 = (st >> 14) & NS_CNT_MASK_DEV;
        const bool leaf = st & NS_LEAF_DEV, masked = st & NS_MASKED_DEV, root = st & NS_ROOT_DEV;
        bool elig;
        int score = base + cc + a.cadd;
        if (root) elig = true;
        else if (masked) elig = false;
        else if (a.touched) {
            score += a.adj;
            const int ncom = (int)ncom0 + a.dcom;
            elig = leaf ? (ncom > 0) : (ncom > 0 || ncom == (int)nmut);     // usher_mapper.cpp:455-456
        } else elig = st & NS_ELIG0_DEV;
        elig = elig && nvalid;
        if (__ballot(elig && score <= bs)) {
            const int smin = wave_min_i32(elig ? score : 0x7FFFFFFF);
            const bool at_min = elig && score == smin;
            const uint32_t cntb = (uint32_t)__popcll(__ballot(at_min));
            const uint32_t rmin = wave_min_u32(at_min ? rank : 0xFFFFFFFFu);
            take(smin, rmin, cntb);
        }
    };
    auto eval_chunk = [&](uint32_t c_in) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_in);   // wave-uniform
        const uint32_t b0 = c * m.seed_stride, b1 = min(full.NB, b0 + m.seed_stride);
        int cc = c0;
        {   // the state of a sequential sweep at the chunk start: enter words of the nodes still open there
            const uint32_t cpi = b0 / full.cp_stride;
            const uint32_t e0 = full.cp_off[cpi], e1 = full.cp_off[cpi + 1];
            for (uint32_t e = e0; e < e1; e += 64) {
                int d = 0;
                if (e + lane < e1) {
                    const uint32_t w = full.cp_word[e + lane];
                    if (bit(w_pos(w))) {
                        const uint32_t sw = find_entry(S, 0u, k, w_pos(w));
                        if (sw != NONE) d = enter_delta(w, sw);
                    }
                }
                cc += (int)wave_sum_u32((uint32_t)d);
            }
            wbytes += (e1 - e0) * 4 + 8;
        }
        // groups of up to 32 blocks: their event offsets and summaries are fetched together (lane = block), their
        // events are scanned for positions of the sample eight loads at a time (lane = event) -- the chunk's memory
        // round trips are a handful instead of four per block
        for (uint32_t bg = b0; bg < b1; bg += 32) {
            const uint32_t ng = min(32u, b1 - bg);
            const uint32_t eo = lane <= ng ? full.blk_eoff[bg + lane] : 0u;
            BlkSum ms{};
            if (lane < ng) ms = full.blk_sum[bg + lane];
            const uint32_t E0 = (uint32_t)__builtin_amdgcn_readlane((int)eo, 0);
            const uint32_t E1 = (uint32_t)__builtin_amdgcn_readlane((int)eo, (int)ng);
            wbytes += (E1 - E0) * 4 + ng * 36;
            uint32_t nhit = 0;
            bool overflow = false;
            for (uint32_t e = E0; e < E1 && !overflow; e += 8 * 64) {
                uint32_t w8[8];
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) w8[q] = (e + q * 64 + lane < E1) ? full.ev_word[e + q * 64 + lane] : W_PAD_DEV;
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) {
                    const uint32_t w = w8[q];
                    uint32_t sw = NONE;
                    if (__ballot(w != W_PAD_DEV && bit(w_pos(w)))) {
                        if (w != W_PAD_DEV && bit(w_pos(w))) sw = find_entry(S, 0u, k, w_pos(w));
                    }
                    const unsigned long long hm = __ballot(sw != NONE);
                    if (hm) {
                        const uint32_t nq = (uint32_t)__popcll(hm);
                        if (nhit + nq > SEED_HIT_CAP) { overflow = true; break; }
                        if (sw != NONE) {
                            uint32_t* hrec = hits + 3 * (nhit + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull)));
                            hrec[0] = e + q * 64 + lane;
                            hrec[1] = w;
                            hrec[2] = sw;
                        }
                        nhit += nq;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!overflow) {
                uint32_t hp = 0;
                for (uint32_t j = 0; j < ng; j++) {
                    const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)eo, (int)(j + 1));
                    uint32_t nhb = 0;                       // hits of block j: the list is in event order
                    while (hp + nhb < nhit && hits[3 * (hp + nhb)] < e1) nhb++;
                    const int sbase = __builtin_amdgcn_readlane(ms.base, (int)j);
                    const uint32_t scnt = (uint32_t)__builtin_amdgcn_readlane((int)ms.cnt, (int)j);
                    if (!nhb) {
                        // no event of the sample in the block: every node scores base + c, the static summary is the answer
                        if (scnt && sbase + cc <= bs) take(sbase + cc, (uint32_t)__builtin_amdgcn_readlane((int)ms.rank, (int)j), scnt);
                    } else {
                        const uint32_t node0 = (uint32_t)__builtin_amdgcn_readlane((int)ms.node0, (int)j);
                        NodeAcc acc;
                        apply_hits(node0, hp, nhb, acc);
                        score_nodes(node0, (uint32_t)__builtin_amdgcn_readlane((int)ms.nn, (int)j), cc, acc);
                        cc += acc.net;
                        hp += nhb;
                    }
                }
            } else {
                // more hit events than the list holds (a sample that lists most positions of the tree): block by block,
                // the events of a block applied 64 at a time through the same list
                for (uint32_t j = 0; j < ng; j++) {
                    const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)eo, (int)j);
                    const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)eo, (int)(j + 1));
                    const uint32_t node0 = (uint32_t)__builtin_amdgcn_readlane((int)ms.node0, (int)j);
                    NodeAcc acc;
                    bool any = false;
                    for (uint32_t e = e0; e < e1; e += 64) {
                        const uint32_t w = (e + lane < e1) ? full.ev_word[e + lane] : W_PAD_DEV;
                        uint32_t sw = NONE;
                        if (w != W_PAD_DEV && bit(w_pos(w))) sw = find_entry(S, 0u, k, w_pos(w));
                        const unsigned long long hm = __ballot(sw != NONE);
                        if (!hm) continue;
                        any = true;
                        if (sw != NONE) {
                            uint32_t* hrec = hits + 3 * (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
                            hrec[0] = e + lane;
                            hrec[1] = w;
                            hrec[2] = sw;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        apply_hits(node0, 0u, (uint32_t)__popcll(hm), acc);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    const int sbase = __builtin_amdgcn_readlane(ms.base, (int)j);
                    const uint32_t scnt = (uint32_t)__builtin_amdgcn_readlane((int)ms.cnt, (int)j);
                    if (!any) {
                        if (scnt && sbase + cc <= bs) take(sbase + cc, (uint32_t)__builtin_amdgcn_readlane((int)ms.rank, (int)j), scnt);
                    } else {
                        score_nodes(node0, (uint32_t)__builtin_amdgcn_readlane((int)ms.nn, (int)j), cc, acc);
                        cc += acc.net;
                    }
                }
            }
        }
        wchunks++;
        if (lane == 0) { atomicMin(&sh[0], bs); if (HEAVY) atomicMin(&heavy->rec[hv_i].best, bs); }
    };

    // ---- (3) + (4) best first: the chunks with the largest count, then level by level while the level's bound
    // |T| - h can still reach the best score found so far; the waves share a level's chunks in slabs of 64 ----
    bool deferred = false;
    for (int h = HEAVY ? min(hmax, heavy->rec[hv_i].level) : hmax; h >= 0; h--) {
        __syncthreads();
        if (tid == 0) sh[6] = HEAVY ? min(sh[0], *reinterpret_cast<volatile int*>(&heavy->rec[hv_i].best)) : sh[0];
        __syncthreads();
        if (sT - h > sh[6]) break;                     // (the same value in every wave: the barriers above order it)
        if (!HEAVY && h == defer_level) {
            // this level and the ones below go to the second pass -- if its table has a record left
            if (tid == 0) {
                const uint32_t slot = atomicAdd(&heavy->count, 1u);
                sh[7] = slot < SEED_HEAVY_CAP ? (int)slot + 1 : 0;
            }
            __syncthreads();
            if (sh[7]) { deferred = true; break; }
            defer_level = -1;
        }
        for (uint32_t s0 = (hv_p * SEED_WAVES + wv) * 64; s0 < nch; s0 += 64 * SEED_WAVES * (HEAVY ? SEED_HEAVY_PARTS : 1u)) {
            const uint32_t c = s0 + lane;
            unsigned long long live = __ballot(c < nch && (int)H[min(c, nch - 1)] == h);
            while (live) {
                const int l = __builtin_ctzll(live);
                live &= live - 1;
                // (the bound may have come down since the level was entered)
                if (sT - h <= min(bs, *reinterpret_cast<volatile int*>(&sh[0]))) eval_chunk(s0 + (uint32_t)l);
            }
        }
    }
    __syncthreads();
    // ---- (5) combine the waves' partials, write the result ----
    if (lane == 0) {
        sh[8 + 3 * wv] = bs;
        sh[9 + 3 * wv] = (int)br;
        sh[10 + 3 * wv] = (int)cnt;
        atomicAdd(&sh[5], (int)wchunks);
        if (work_counter) atomicAdd(work_counter + WALK_COUNTERS + ((blockIdx.x * SEED_WAVES + wv) & (WALK_COUNTERS - 1)),
                                    (unsigned long long)wbytes + (wv == 0 ? (unsigned long long)nh * row_words * 4 + (unsigned long long)k * 4 : 0ull));
    }
    __syncthreads();
    if (tid == 0) {
        int fs = 0x7FFFFFFF;
        uint32_t fr = 0xFFFFFFFFu, fc = 0;
        for (uint32_t w2 = 0; w2 < SEED_WAVES; w2++) {
            const int ps = sh[8 + 3 * w2];
            const uint32_t pr = (uint32_t)sh[9 + 3 * w2], pc = (uint32_t)sh[10 + 3 * w2];
            if (!pc) continue;
            if (ps < fs) { fs = ps; fr = pr; fc = pc; }
            else if (ps == fs) { fc += pc; fr = min(fr, pr); }
        }
        if (HEAVY) heavy->part[hv_i * SEED_HEAVY_PARTS + hv_p] = SeedHeavy::Part{fs, fr, fc};
        else if (deferred) heavy->rec[sh[7] - 1] = SeedHeavy::Rec{r, defer_level, fc ? min(fs, root_score[r] + 1) : root_score[r] + 1, fs, fr, fc};
        else emit_result(m, r, read_off, read_word, fs, fr, fc, best_bfs_j, score_out, num_best, flags);
        if (work_counter && HEAVY) atomicAdd(work_counter + 2 * WALK_COUNTERS + 1, (unsigned long long)(uint32_t)sh[5]);
        if (work_counter && !HEAVY) {
            // behind the walks' two counter arrays: seeded samples, chunks evaluated, chunks in all, the most chunks one
            // sample evaluated, and samples by chunks evaluated (<= 1, <= 4, <= 16, <= 64, <= 256, <= 1024, <= 4096, more)
            unsigned long long* sc = work_counter + 2 * WALK_COUNTERS;
            const uint32_t ev = (uint32_t)sh[5];
            atomicAdd(sc, 1ull);
            atomicAdd(sc + 1, (unsigned long long)ev);
            atomicAdd(sc + 2, (unsigned long long)nch);
            atomicMax(sc + 3, (unsigned long long)ev);
            uint32_t bkt = 0;
            while (bkt < 7 && ev > (1u << (2 * bkt))) bkt++;
            atomicAdd(sc + 4 + bkt, 1ull);
        }
    }
}

// the deferred samples' result = what the first pass found + the parts of the second; clears the table for the next call
__global__ void k_seed_heavy_finalize(DevMAT m, SeedHeavy* __restrict__ heavy, const uint32_t* __restrict__ read_off,
                                      const uint32_t* __restrict__ read_word, uint32_t* __restrict__ best_bfs_j,
                                      int32_t* __restrict__ score_out, uint32_t* __restrict__ num_best, uint32_t* __restrict__ flags) {
    const uint32_t n = min(heavy->count, SEED_HEAVY_CAP);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const SeedHeavy::Rec rc = heavy->rec[i];
        int fs = rc.cnt ? rc.score : 0x7FFFFFFF;
        uint32_t fr = rc.cnt ? rc.rank : 0xFFFFFFFFu, fc = rc.cnt;
        for (uint32_t p = 0; p < SEED_HEAVY_PARTS; p++) {
            const SeedHeavy::Part pt = heavy->part[i * SEED_HEAVY_PARTS + p];
            if (!pt.cnt) continue;
            if (pt.score < fs) { fs = pt.score; fr = pt.rank; fc = pt.cnt; }
            else if (pt.score == fs) { fc += pt.cnt; fr = min(fr, pt.rank); }
        }
        emit_result(m, rc.read, read_off, read_word, fs, fr, fc, best_bfs_j, score_out, num_best, flags);
    }
    __syncthreads();
    if (threadIdx.x == 0) heavy->count = 0;
}

hipError_t seed_set_max_lds(uint32_t bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)k_seed<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_seed<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

size_t seed_heavy_bytes() { return sizeof(SeedHeavy); }

uint32_t seed_lds_bytes(const DevMAT& m, uint32_t ent_cap) {
    return (m.bm_words + ent_cap + (SEED_MAX_HARD + 1) + seed_h_words(m.seed_chunks) + SEED_SHARED_WORDS + SEED_WAVES * 3 * SEED_HIT_CAP) * 4;
}

hipError_t launch_seed(const DevMAT& m, const DevStream& full, const uint32_t* list, uint32_t n_list, uint32_t ent_cap,
                       const uint32_t* d_read_off, const uint32_t* d_read_word, const int32_t* root_score,
                       uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best, uint32_t* flags,
                       unsigned long long* work_counter, void* heavy_table, hipStream_t stream) {
    if (n_list == 0) return hipSuccess;
    SeedHeavy* heavy = static_cast<SeedHeavy*>(heavy_table);       // (zeroed at creation, cleared by the finalize kernel; nullptr: one pass)
    hipLaunchKernelGGL(k_seed<false>, dim3(n_list), dim3(SEED_THREADS), seed_lds_bytes(m, ent_cap), stream, m, full, list, n_list, ent_cap,
                       d_read_off, d_read_word, root_score, best_bfs_j, score, num_best, flags, work_counter, heavy);
    if (heavy) {
        // sized for the whole table: the number of deferred samples stays on the device (workgroups beyond it leave at once)
        hipLaunchKernelGGL(k_seed<true>, dim3(SEED_HEAVY_CAP * SEED_HEAVY_PARTS), dim3(SEED_THREADS), seed_lds_bytes(m, ent_cap), stream, m, full,
                           list, n_list, ent_cap, d_read_off, d_read_word, root_score, best_bfs_j, score, num_best, flags, work_counter, heavy);
        hipLaunchKernelGGL(k_seed_heavy_finalize, dim3(1), dim3(256), 0, stream, m, heavy, d_read_off, d_read_word, best_bfs_j, score, num_best, flags);
    }
    return hipGetLastError();
}

}  // namespace wepp
