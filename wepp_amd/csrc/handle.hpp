// handle.hpp -- the object behind wepp_mat_t and the small helpers the C-ABI translation
// units (capi.cpp, epp_capi.cpp) share.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/wepp_place.h"
#include "device_mat.hpp"
#include "errors.hpp"
#include "flatmat.hpp"
#include "host_pool.hpp"
#include "tunables.hpp"

using namespace wepp;

// What ONE placement call in flight needs of its own: workspace, routing counters, the side streams its launch
// chains run on.  A handle has two: the sub-batches of wepp_place_batch alternate between them, so that the routing
// of sub-batch k+1 (kernels, a device-to-host copy, the host's poll) overlaps the walks of sub-batch k, each lane's
// work ordered on the lane's own compute stream.  wepp_place_batch_device uses lane 0.
struct PlaceLane {
    // grow-only workspace: read list, routing arrays, partial results
    void* ws = nullptr;
    size_t ws_bytes = 0;
    void* ws2 = nullptr;              // grow-only workspace of the chunked walks (job tables, partials)
    size_t ws2_bytes = 0;
    uint32_t* d_info = nullptr;       // two sets of tier_info (TI_WORDS each, used alternately: k_route clears the other one) followed by blk_counts
    uint32_t info_idx = 0;            // the set the next call uses (zero: cleared at creation or by the previous call's k_route)
    uint32_t* h_info = nullptr;       // pinned copy of tier_info
    // the launch chains of a call are independent: they run concurrently on side streams
    hipStream_t side[MAX_STREAMS] = {};
    hipEvent_t fork_ev = nullptr, join_ev[MAX_STREAMS] = {};
    hipEvent_t route_ev = nullptr;    // behind k_route: the plain walks start from here on their side stream, before the host has the counters
};

struct wepp_mat {
    static constexpr uint32_t kLanes = 2;
    int device = 0;
    DevMAT dev{};
    std::vector<DevStream> streams;
    std::vector<DevWalk> walks;       // position index + range-query structures of every stream (k_walk)
    uint64_t wc_nodes = 0;            // nodes of all window crowns (the arena of slot WC_SLOT)
    uint32_t wc_count = 0;            // window crowns built
    PlaceTunables tun;                // the environment's knobs, read once when the handle was created (tunables.hpp)
    int use_walk = 1;                 // reads with few entries walk their own events (WEPP_WALK=0: sweeps only)
    int use_seeds = 1;                // whole-genome samples go by chunk signatures (WEPP_SEED=0: tile sweeps)
    std::atomic<uint32_t> job_events[2] = {{WALK_JOB_EVENTS}, {WALK_JOB_EVENTS}};   // events per job of the chunked classes in the next call (a hint: the two launch threads of a pipelined call read and write it freely)
    int walk_ok = 1;                  // 0: a stream is too large for the walk's packed interval stack (sweeps only)
    void* d_seed_heavy = nullptr;     // k_seed's table of samples handed to its second pass (seed_kernels.hip), zero between calls
    unsigned long long* d_work = nullptr;   // [WALK_COUNTERS] loop iterations of the walks, [WALK_COUNTERS] bytes the walks / seeds asked memory for,
                                            // [D_WORK_EXTRA] seeded samples, chunks they evaluated, chunks in all, most per sample, histogram -- since the last timing reset
    static constexpr uint32_t D_WORK_EXTRA = 16;
    static constexpr size_t D_WORK_BYTES = (2 * WALK_COUNTERS + D_WORK_EXTRA) * sizeof(unsigned long long);
    std::vector<uint64_t> stream_bytes;
    std::vector<DevStream> wstreams;  // window streams (PLAN_WIN)
    const DevStream* d_wstreams = nullptr;  // ... and the same records on the device (k_sweep_windows names a stream by its index)
    std::vector<uint64_t> wstream_bytes;
    wepp_mat_stats stats{};
    std::vector<uint32_t> bfs2id;
    std::vector<uint32_t> dfs2id;     // caller id of the node with pre-order (arena) index k
    // EPP event stream (flatmat.hpp), resident next to the sweep streams
    const uint32_t* epp_word = nullptr;
    const uint32_t* epp_node = nullptr;
    uint64_t epp_events = 0;
    void* epp_ws = nullptr;           // (unused)
    size_t epp_ws_bytes = 0;
    // device buffers of wepp_epp_map kept between calls (gigabytes at 16 M nodes: a hipMalloc of that size costs
    // tens of milliseconds); a call takes the blocks that fit and hands everything back when it returns.  Freed
    // with the handle (release() has selected the device by then).
    struct DevBlockCache {
        std::vector<std::pair<void*, size_t>> blocks;
        ~DevBlockCache() { for (auto& b : blocks) (void)hipFree(b.first); }
    } epp_cache;
    std::vector<uint32_t> epp_pending;   // EPP lists of the last wepp_epp_map that did not fit the caller's buffer (wepp_epp_fetch_lists)
    std::vector<void*> allocs;
    uint32_t tile_reads = 64;
    int use_crowns = 1;
    // grow-only device copies of the caller's host buffers (wepp_place_batch): reads in, results out
    void* io_in = nullptr;
    void* io_out = nullptr;
    size_t io_in_bytes = 0, io_out_bytes = 0;
    void* pin = nullptr;              // pinned staging of the same (pageable caller buffers go through it)
    size_t pin_bytes = 0;
    std::unique_ptr<HostPool> pool;   // host workers of wepp_place_batch (started by the first large batch)
    // wepp_place_batch as a pipeline: sub-batch k's reads go up on pipe_h2d while the kernels of k-1 run on
    // pipe_compute and the results of k-2 come down on pipe_d2h; one event per sub-batch and stage
    static constexpr uint32_t kPipeMax = 8;
    uint32_t pipe_sub_batches = 0;    // 0: chosen per call (one device call per 2 M reads, at most 8)
    hipStream_t pipe_h2d[kLanes] = {}, pipe_d2h[kLanes] = {}, pipe_compute[kLanes] = {};     // (a set per lane: the two launch threads of a call do not share a stream)
    hipEvent_t pipe_up[kPipeMax] = {}, pipe_done[kPipeMax] = {}, pipe_out[4 * kPipeMax] = {};   // (pipe_out: one per sub-batch and result array)
    void* pin_out = nullptr;          // pinned staging of the results (the reads' staging is `pin`)
    size_t pin_out_bytes = 0;
    // plan id of every read of the most recent placement call (k_route writes it; wepp_mat_last_tiers / _plans read it);
    // grow-only, outside the workspace so that the sub-batches of one host call add up to the whole call's picture
    uint32_t* d_wsid_of = nullptr;    // window crown (index into DevMAT::wc_info) of the reads routed to slot WC_SLOT, same sizing
    uint8_t* d_plan_of = nullptr;
    size_t plan_of_bytes = 0;
    PlaceLane lane[2];
    static constexpr uint32_t kRing = 64;
    hipEvent_t ev0[kRing] = {}, ev1[kRing] = {};
    std::atomic<uint32_t> ww_by_jobs{0};  // the previous call held too many reads with 17 - 256 events for a wave each: this call cuts their walks into jobs (a hint, like job_events)
    std::atomic<uint64_t> n_timed{0}; // placement calls since the last timing reset (a call claims its slot of the event ring when it starts)
    std::mutex stat_mu;               // the counters below: the sub-batches of a pipelined wepp_place_batch finish on two host threads
    uint64_t plan_reads_in = 0;       // reads of the sub-batches of the current call that have been planned
    uint64_t last_passes = 0, last_bytes = 0;
    uint64_t acc_passes = 0, acc_bytes = 0;   // the same summed over the calls since the last timing reset
    uint64_t last_walk_reads = 0;     // reads of the most recent call that walked their own events (k_walk)
    uint32_t last_n_reads = 0;        // reads of the most recent placement call (wepp_mat_last_tiers)
};

namespace wepp {

inline int hip_fail(hipError_t e, const char* what) {
    return set_error(WEPP_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                      \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return hip_fail(_e, #expr);  \
    } while (0)

template <typename T>
inline int upload(wepp_mat* h, const std::vector<T>& v, const T** out) {
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    h->allocs.push_back(p);
    h->stats.device_bytes += bytes;
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return WEPP_OK;
}

}  // namespace wepp
