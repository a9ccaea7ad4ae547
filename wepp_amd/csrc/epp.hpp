// epp.hpp -- WEPP's own read placement (wepp_filter::cartesian_map,
// src/WEPP/initial_filter.cpp:41-239) on the flat MAT: shared declarations of
// epp_kernels.hip and epp_capi.cpp.  See DESIGN.md section 4.8.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wepp {

constexpr uint32_t EPP_BINS = 50;            // NUM_RANGE_BINS, src/WEPP/config.hpp:13
constexpr uint32_t EPP_MAX_GROUPS = 512;     // genome windows per call (the reference builds 25 range trees)
constexpr uint32_t EPP_SEL_EVENTS = 1024;    // events of the full stream one wave selects from
constexpr uint32_t EPP_NO_LIST = 0xFFFFFFFFu;

// One window = a run of reads that are consecutive in (start, end) order; its event stream is
// the sub-sequence of the MAT's EPP event stream with positions inside [ws, we].
struct EppGroup {
    uint32_t ws, we;          // genome window covered by the group's reads
    uint32_t tile0, ntiles;   // tiles (64 * rpl consecutive sorted reads) of the group
    uint32_t n_events;        // length of the window's event stream
    uint32_t nchunks;         // the stream is swept in nchunks pieces of chunk_events
    uint32_t job0;            // first sweep job: job = job0 + chunk * ntiles + tile
    uint32_t pad;
    uint64_t soff;            // offset of the stream in st_word / st_node
};

struct EppSweepArgs {
    const EppGroup* groups;
    uint32_t G, n_jobs, R, N;
    uint32_t chunk_events, bm_words, tab_rows, bin_size;
    const uint32_t* st_word;
    const uint32_t* st_node;
    const uint32_t* read_off;
    const uint32_t* read_word;
    const int32_t* start;
    const int32_t* end;
    const int32_t* degree;
    const uint32_t* order;    // sorted index -> read
    // per (job, lane): pass 1 writes (min of the relative distance over the chunk's nodes, number of
    // nodes attaining it, net change); k_epp_combine turns net into the distance at the chunk's
    // start and cnt into the EPP-list cursor at the chunk's start
    int32_t* part_min;
    uint32_t* part_cnt;
    int32_t* part_net;
    // per sorted read, written by k_epp_combine
    int32_t* best;
    uint32_t* mult;
    long long* delta_fx;
    // pass 2
    const uint64_t* epp_base; // per read: offset of its EPP list, or ~0 when the list is not kept
    uint32_t* epp_nodes;
    unsigned long long* diff_score;   // [N+1] fixed point
    int* diff_cnt;                    // [(N+1) * EPP_BINS] or null
    double fx_scale;                  // 2^k
};

hipError_t launch_epp_select_count(const uint32_t* ev_word, uint64_t n_events, const EppGroup* groups,
                                   const uint32_t* we_max, uint32_t G, uint32_t nblk, uint32_t* cnt,
                                   hipStream_t stream);
hipError_t launch_epp_select_scan(uint32_t* cnt, uint32_t G, uint32_t nblk, uint32_t* totals, hipStream_t stream);
hipError_t launch_epp_select_scatter(const uint32_t* ev_word, const uint32_t* ev_node, uint64_t n_events,
                                     const EppGroup* groups, const uint32_t* we_max, uint32_t G, uint32_t nblk,
                                     const uint32_t* cnt, uint32_t* st_word, uint32_t* st_node, hipStream_t stream);
// rpl = reads per lane (1 or 4): a tile is 64 * rpl reads
hipError_t launch_epp_sweep(const EppSweepArgs& a, int pass, uint32_t rpl, uint32_t lds_bytes, hipStream_t stream);
hipError_t launch_epp_combine(const EppSweepArgs& a, uint32_t tiles_per_group, uint32_t rpl, hipStream_t stream);
// prefix sums of the difference arrays -> per-haplotype outputs (arena order)
hipError_t launch_epp_finish(uint32_t N, const unsigned long long* diff_score, double inv_scale, double* score,
                             const int* diff_cnt, const int* true_counts, int* counts, double* divergence,
                             void* scratch, hipStream_t stream);
size_t epp_finish_scratch_bytes(uint32_t N);

}  // namespace wepp
