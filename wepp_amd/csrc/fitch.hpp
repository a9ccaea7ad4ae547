// fitch.hpp -- device views and launchers of the per-site Fitch-Sankoff kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wepp {

struct FitchTree {
    uint32_t N, max_depth;
    const uint32_t* meta;       // [N] DFS pre-order: depth | leaf << 31
};

struct FitchSites {
    uint32_t n_sites;
    const uint8_t* ref_idx;     // [n_sites] 0..3
    const uint32_t* var_off;    // [n_sites + 1]
    const uint32_t* var_dfs;    // tree samples of a row, ascending DFS index
    const uint8_t* var_nuc;     // allele masks
};

constexpr uint32_t FITCH_MAX_DEPTH = 140;   // (depth + 1) KiB of LDS per wave

hipError_t launch_fitch_forward(const FitchTree& t, const FitchSites& s, uint32_t batch0, uint32_t nbatches,
                                uint8_t* tables, hipStream_t stream);
hipError_t launch_fitch_backward(const FitchTree& t, const FitchSites& s, uint32_t batch0, uint32_t nbatches,
                                 const uint8_t* tables, unsigned long long* out_count, uint64_t capacity, uint2* out,
                                 hipStream_t stream);

}  // namespace wepp
