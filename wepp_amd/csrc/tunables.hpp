// tunables.hpp -- every environment knob of the placement path, read ONCE when a handle is created (wepp_mat::tun)
// or once per flatten (FlattenOptions), clamped to what the kernels take.  None of them changes a result: they are
// A/B aids, profiling aids and test hooks (DESIGN.md "Diagnostic switches").  The hot call path reads struct fields,
// never the environment.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "device_mat.hpp"
#include "flatten_options.hpp"

namespace wepp {


struct PlaceTunables {
    // walks
    bool walk = true;                    // WEPP_WALK=0: every read is placed by sweeps
    uint32_t walk_eager_nodes = WALK_EAGER_MAX_NODES;   // WEPP_WALK_EAGER_NODES
    uint32_t walk_max_events = WALK_MAX_EVENTS;         // WEPP_WALK_MAX_EVENTS: events from which a walk is cut into jobs
    uint32_t job_events = 0;             // WEPP_WALK_JOB_EVENTS (0: follow the handle's traffic)
    uint32_t stack8 = WALK8_ROWS;        // WEPP_WALK_STACK8  (<= WALK8_STACK: the rows a walk workgroup can get)
    uint32_t stack16 = WALK16_ROWS;      // WEPP_WALK_STACK16 (<= WALK16_STACK)
    bool ww_fixed = false;               // WEPP_WW_BLOCK_MAX_SMALL / _BIG given: reads per routing block listed for k_walk_wave (default: all or none, by the previous call's counts)
    uint32_t ww_block_max_small = 0xFFFFFFFFu, ww_block_max_big = 0xFFFFFFFFu;
    bool sort_reads = true;              // WEPP_SORT_READS=0: keep the caller's order on the whole-tree stream
    bool walk_sort = true;               // WEPP_WALK_SORT=0: keep the caller's order in the chunked walk classes
    // sweeps
    bool win_eager = true;               // WEPP_WIN_EAGER=0: window streams prune with the block minimum
    uint32_t target_waves = 4096;        // WEPP_TARGET_WAVES
    uint32_t target_waves_dense = 16384; // WEPP_TARGET_WAVES_DENSE
    uint64_t chunk_bytes = SWEEP_CHUNK_BYTES;   // WEPP_CHUNK_BYTES
    bool sweep_unfused = false;          // WEPP_SWEEP_UNFUSED=1: one sweep launch per plan, back to back
    bool blind16 = false;                // WEPP_BLIND16=1: the plain walks of 9 - 16 entries launched blind behind k_route too (measured: they start when the walks of 1 - 8 entries end either way)
    bool windows_unfused = false;        // WEPP_WINDOWS_UNFUSED=1: one launch per window plan, chunks sized per plan (round 3's form; results identical)
    // seeds: whole-genome samples
    bool seed = true;                    // WEPP_SEED=0: whole-genome samples take the tile sweeps
    bool seed_heavy = true;              // WEPP_SEED_HEAVY=0: no second pass (a sample's workgroup evaluates every chunk it must itself)
    uint32_t seed_min_hard = SEED_MIN_HARD;          // WEPP_SEED_MIN_HARD
    uint32_t seed_min_nodes = SEED_MIN_STREAM_NODES; // WEPP_SEED_MIN_NODES
    // host pipeline of wepp_place_batch
    uint32_t pipe_sub_batches = 0;       // WEPP_PIPE_SUBBATCHES
    uint32_t pipe_min_chunks = 4;        // WEPP_PIPE_MIN_CHUNKS: copy chunks of a host-buffer batch (rounded up to a multiple of its sub-batches)
    uint32_t host_threads = 0;           // WEPP_HOST_THREADS (0: cores this process may use / handles alive)
    // diagnostics
    bool debug_plans = false, debug_timing = false, walk_debug = false;

    static PlaceTunables from_env() {
        PlaceTunables t;
        t.walk = env::flag("WEPP_WALK", true);
        t.walk_eager_nodes = (uint32_t)env::u64("WEPP_WALK_EAGER_NODES", WALK_EAGER_MAX_NODES, 0, 0xFFFFFFFFu);
        t.walk_max_events = (uint32_t)env::u64("WEPP_WALK_MAX_EVENTS", WALK_MAX_EVENTS, 0, 0xFFFFu);
        t.job_events = (uint32_t)env::u64("WEPP_WALK_JOB_EVENTS", 0, 1, 0xFFFFu);
        // a knob above the rows the walk kernels allocate would let a lane write past its wave's LDS region
        t.stack8 = (uint32_t)env::u64("WEPP_WALK_STACK8", WALK8_ROWS, 0, WALK8_STACK);
        t.stack16 = (uint32_t)env::u64("WEPP_WALK_STACK16", WALK16_ROWS, 0, WALK16_STACK);
        t.ww_fixed = env::is_set("WEPP_WW_BLOCK_MAX_SMALL") || env::is_set("WEPP_WW_BLOCK_MAX_BIG");
        t.ww_block_max_small = (uint32_t)env::u64("WEPP_WW_BLOCK_MAX_SMALL", 0xFFFFFFFFu, 0, 0xFFFFFFFFu);
        t.ww_block_max_big = (uint32_t)env::u64("WEPP_WW_BLOCK_MAX_BIG", 0xFFFFFFFFu, 0, 0xFFFFFFFFu);
        t.sort_reads = env::flag("WEPP_SORT_READS", true);
        t.walk_sort = t.sort_reads && env::flag("WEPP_WALK_SORT", true);
        t.win_eager = env::flag("WEPP_WIN_EAGER", true);
        t.target_waves = (uint32_t)env::u64("WEPP_TARGET_WAVES", 4096, 1, 1u << 24);
        t.target_waves_dense = (uint32_t)env::u64("WEPP_TARGET_WAVES_DENSE", 16384, 1, 1u << 24);
        t.chunk_bytes = env::u64("WEPP_CHUNK_BYTES", SWEEP_CHUNK_BYTES, 4096, 1ull << 40);
        t.sweep_unfused = env::is_set("WEPP_SWEEP_UNFUSED") && env::flag("WEPP_SWEEP_UNFUSED", false);
        t.blind16 = env::flag("WEPP_BLIND16", false);
        t.windows_unfused = env::is_set("WEPP_WINDOWS_UNFUSED") && env::flag("WEPP_WINDOWS_UNFUSED", false);
        t.seed = env::flag("WEPP_SEED", true);
        t.seed_heavy = env::flag("WEPP_SEED_HEAVY", true);
        t.seed_min_hard = (uint32_t)env::u64("WEPP_SEED_MIN_HARD", SEED_MIN_HARD, 0, 0xFFFFu);
        t.seed_min_nodes = (uint32_t)env::u64("WEPP_SEED_MIN_NODES", SEED_MIN_STREAM_NODES, 0, 0xFFFFFFFFu);
        t.pipe_sub_batches = (uint32_t)env::u64("WEPP_PIPE_SUBBATCHES", 0, 1, 8);
        t.pipe_min_chunks = (uint32_t)env::u64("WEPP_PIPE_MIN_CHUNKS", 4, 1, 64);
        t.host_threads = (uint32_t)env::u64("WEPP_HOST_THREADS", 0, 1, 256);
        t.debug_plans = env::is_set("WEPP_DEBUG_PLANS");
        t.debug_timing = env::is_set("WEPP_DEBUG_TIMING");
        t.walk_debug = env::is_set("WEPP_WALK_DEBUG");
        return t;
    }
};

}  // namespace wepp
