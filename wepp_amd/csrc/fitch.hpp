// fitch.hpp -- device views and launchers of the per-site Fitch-Sankoff kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wepp {

struct FitchTree {
    uint32_t N, max_depth;
    uint32_t C;                     // chunks of consecutive DFS nodes, one wave each
    const uint32_t* meta;           // [N] DFS pre-order: depth | leaf << 31
    const uint32_t* chunk_start;    // [C + 1]
    const uint32_t* chunk_depth;    // [C + 1] nodes open when chunk c starts (c = C: after the last node)
    const uint32_t* chunk_min;      // [C] smallest depth of a node of the chunk
    const uint32_t* chunk_open;     // [(C + 1) * (max_depth + 1)] the open node of every level at a chunk start
};

struct FitchSites {
    uint32_t n_sites;
    const uint8_t* ref_idx;     // [n_sites] 0..3
    const uint32_t* var_off;    // [n_sites + 1]
    const uint32_t* var_dfs;    // tree samples of a row, ascending DFS index
    const uint8_t* var_nuc;     // allele masks
};

// level-synchronous form: the tree in BFS order
struct FitchLevels {
    uint32_t N, n_levels;
    const uint32_t* coff;           // [N + 1] children of node i = BFS indices [coff[i], coff[i + 1]) (BFS keeps siblings together)
    const uint32_t* parent;         // [N] BFS index of the parent (root: 0)
};
#ifndef WEPP_FITCH_LEVEL_CHUNK
#define WEPP_FITCH_LEVEL_CHUNK 256
#endif
constexpr uint32_t FITCH_LEVEL_CHUNK = WEPP_FITCH_LEVEL_CHUNK;   // consecutive nodes of a level per wave (measured: 256 beats 512, 2048, 8192 -- many short waves fill the chip level by level)

#ifndef WEPP_FITCH_LEVEL_WAVES
#define WEPP_FITCH_LEVEL_WAVES 4
#endif
constexpr uint32_t FITCH_ROWS_PER_LANE = 4;   // level kernels: a batch is 64 * 4 VCF rows, one dword of bytes per lane and node
constexpr uint32_t FITCH_LEVEL_WAVES = WEPP_FITCH_LEVEL_WAVES;   // independent waves per workgroup of the level kernels

constexpr uint32_t FITCH_MAX_DEPTH = 140;   // (depth + 1) KiB of LDS per wave

// forward = chunk-parallel pass + stitch of the nodes that span chunk boundaries;
// inh_part / out_part: [nbatches][C][max_depth + 1][64] int4 scratch each
hipError_t launch_fitch_forward(const FitchTree& t, const FitchSites& s, uint32_t batch0, uint32_t nbatches,
                                uint8_t* tables, int4* inh_part, int4* out_part, hipStream_t stream);
// set form (fitch_kernels.hip): valid when every observed allele set is non-empty and no node has
// more than FITCH_SETS_MAX_CHILDREN children; scratch is uint2 per (chunk, level, row); tables then
// hold "not optimal" masks (launch_fitch_backward with masks = true)
constexpr uint32_t FITCH_SETS_MAX_CHILDREN = 32767;
hipError_t launch_fitch_forward_sets(const FitchTree& t, const FitchSites& s, uint32_t batch0, uint32_t nbatches,
                                     uint8_t* tables, uint2* inh_part, uint2* out_part, hipStream_t stream);
// level-synchronous form (the default when the set form is valid): `bytes` = [nbatches][N][256] in BFS
// order, batches of 256 rows (batch0 / nbatches count those); FitchSites::var_dfs then holds BFS indices; emitted node indices are BFS indices
hipError_t launch_fitch_levels(const FitchLevels& t, const uint32_t* h_level_off, const FitchSites& s, uint32_t batch0,
                               uint32_t nbatches, uint8_t* bytes, unsigned long long* out_count, uint64_t capacity,
                               uint2* out, hipStream_t stream);
hipError_t launch_fitch_backward(const FitchTree& t, const FitchSites& s, uint32_t batch0, uint32_t nbatches,
                                 const uint8_t* tables, bool masks, unsigned long long* out_count, uint64_t capacity,
                                 uint2* out, hipStream_t stream);

// order of the emitted mutations: by row, then by BFS index of the node (the order mapper_body emits them,
// :115); k_fitch_sort_keys builds (row << 28 | node) from the queue entries, sort_reads.hip sorts
hipError_t launch_fitch_sort_keys(const uint2* out, uint64_t n, unsigned long long* keys, uint32_t* vals, hipStream_t stream);
hipError_t launch_fitch_decode(const unsigned long long* keys, const uint32_t* vals, const uint32_t* bfs2id, uint64_t n,
                               uint32_t* out_site, uint32_t* out_node, uint8_t* out_par, uint8_t* out_mut,
                               hipStream_t stream);
// the rows of a call prepared on the device (sort_reads.hip): node ids -> keys, sorted per row, duplicates dropped;
// the result is in (keys_b, nuc_b) or, after a second sort, in (keys_a, nuc_a).  hipErrorInvalidValue = a node id
// out of range.  Synchronises the stream once (two flags come back).
hipError_t fitch_rows_temp_bytes(uint64_t nv, uint32_t n_sites, size_t* bytes);
hipError_t launch_fitch_prepare(const uint32_t* d_var_node, const uint8_t* d_var_nuc, const uint32_t* d_var_off,
                                uint32_t n_sites, uint64_t nv, uint32_t N, const uint32_t* d_id2key, uint32_t* keys_a,
                                uint8_t* nuc_a, uint32_t* keys_b, uint8_t* nuc_b, uint32_t* d_flags, void* temp,
                                size_t temp_bytes, bool* result_in_b, hipStream_t stream);
hipError_t sort_u64_u32_temp_bytes(uint64_t n, uint32_t end_bit, size_t* bytes);
hipError_t launch_sort_u64_u32(const unsigned long long* keys_in, unsigned long long* keys_out, const uint32_t* vals_in,
                               uint32_t* vals_out, uint64_t n, uint32_t end_bit, void* temp, size_t temp_bytes,
                               hipStream_t stream);

}  // namespace wepp
