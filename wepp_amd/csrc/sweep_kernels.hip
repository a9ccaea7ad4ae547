// sweep_kernels.hip -- the event-stream sweeps: one wavefront sweeps a chunk of a stream for a TILE of up to 64
// reads (lane = read); blocks without an event for a read cost that read one summary update, blocks with events are
// re-evaluated node by node (lane = node) for just the reads concerned.  Variants: plain (short
// reads), DENSE (tile-sorted position index in LDS), WIN (per-position read masks, lane = (event, read) pair).
// Plus the finalize kernels that combine the chunk partials of a read.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_mat.hpp"
#include "place_dev.hpp"

namespace wepp {

namespace {

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

}  // namespace

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
        const uint32_t off = (pos >> 3) & bm_mask4;
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
        // wave-uniform; false for most blocks.  One 64-bit OR and one compare: hits of either half, or bit 0
        // from the sign of 128 - (events of the block) (a block holds fewer than 2^31 events)
        const bool any_hit = (hm0 | hm1 | (unsigned long long)((128u - (e1 - e0)) >> 31)) != 0;
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

// all the window plans of one placement call in ONE launch (1.2 kb reads: ~30 windows, each with too few tiles to fill
// the chip on its own): a workgroup = one tile of one window and DENSE_WAVES chunks of the window's stream, the plans in
// the order of the kernel argument (capi.cpp: longest chunks first)
__global__ __launch_bounds__(64 * DENSE_WAVES) void k_sweep_windows(const DevStream* __restrict__ wstreams, WinPlans pl, uint32_t bm_words,
                                                                    uint32_t max_pos, const uint32_t* __restrict__ read_off,
                                                                    const uint32_t* __restrict__ read_word,
                                                                    const int32_t* __restrict__ root_score) {
    uint32_t p = 0;
    while (p + 1 < pl.n && blockIdx.x >= pl.p[p].wg_end) p++;
    const WinPlanDev& q = pl.p[p];
    const uint32_t wg0 = p ? pl.p[p - 1].wg_end : 0;
    const DevStream st = wstreams[q.sid];
    sweep_tile<true, true, true>(st, blockIdx.x - wg0, 0u, bm_words, max_pos, q.ent_cap, q.win_base, read_off, read_word, root_score,
                                 q.list, q.n_list, q.T, q.ntiles, q.bpc, q.part_score, q.part_rank, q.part_cnt);
}

// -----------------------------------------------------------------------------
// finalize: combine the chunks of a read, map the winner back to the
// reference's BFS index and recompute its has_unique flag
// (usher_mapper.cpp:184,199,262,472,492).
// -----------------------------------------------------------------------------
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

__global__ void k_finalize_windows(DevMAT m, WinPlans pl, const uint32_t* __restrict__ read_off,
                                   const uint32_t* __restrict__ read_word, uint32_t* __restrict__ best_bfs_j,
                                   int32_t* __restrict__ score, uint32_t* __restrict__ num_best,
                                   uint32_t* __restrict__ flags) {
    uint32_t p = 0;
    while (p + 1 < pl.n && blockIdx.x >= pl.p[p].fin_end) p++;
    const WinPlanDev& q = pl.p[p];
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
// launchers (called from capi.cpp)
// -----------------------------------------------------------------------------
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

hipError_t launch_sweep_windows(const DevMAT& m, const DevStream* d_wstreams, const WinPlans& pl, const uint32_t* d_read_off,
                                const uint32_t* d_read_word, const int32_t* root_score, uint32_t lds_bytes, hipStream_t stream) {
    if (pl.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sweep_windows, dim3(pl.p[pl.n - 1].wg_end), dim3(64 * DENSE_WAVES), lds_bytes, stream, d_wstreams, pl, m.bm_words,
                       m.max_pos, d_read_off, d_read_word, root_score);
    return hipGetLastError();
}

hipError_t launch_finalize_windows(const DevMAT& m, const WinPlans& pl, const uint32_t* d_read_off, const uint32_t* d_read_word,
                                   uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best, uint32_t* flags, hipStream_t stream) {
    if (pl.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_finalize_windows, dim3(pl.p[pl.n - 1].fin_end), dim3(256), 0, stream, m, pl, d_read_off, d_read_word, best_bfs_j,
                       score, num_best, flags);
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
    e = hipFuncSetAttribute((const void*)k_sweep_windows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_sweep<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace wepp
