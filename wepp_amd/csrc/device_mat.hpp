// device_mat.hpp -- the flat MAT as the kernels see it (device pointers) and
// the launcher prototypes shared between place_kernels.hip and capi.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flatmat.hpp"

namespace wepp {

// device-side mirrors of the flatmat.hpp constants (plain ints for kernels)
constexpr int32_t SCORE_INF_DEV = SCORE_INF;
constexpr uint32_t NS_CNT_MASK_DEV = NS_CNT_MASK;
constexpr uint32_t NS_LEAF_DEV = NS_LEAF, NS_MASKED_DEV = NS_MASKED, NS_ELIG0_DEV = NS_ELIG0, NS_ROOT_DEV = NS_ROOT;
constexpr uint32_t EV_OFF_MASK_DEV = EV_OFF_MASK;
constexpr uint32_t W_EXIT_DEV = W_EXIT, W_LEAF_DEV = W_LEAF, W_PAD_DEV = W_PAD;
constexpr uint32_t WEPP_FLAG_HAS_UNIQUE_DEV = WEPP_FLAG_HAS_UNIQUE;

struct DevMAT {
    uint32_t N, NB, cp_stride, bm_words, max_pos;
    const uint32_t* node_woff;
    const uint32_t* words;
    const int64_t* nkey;
    const uint32_t* nstat;
    const uint32_t* rank2dfs;
    const uint32_t* dfs2bfs;
    const uint32_t* blk_node0;
    const uint32_t* blk_eoff;
    const BlkSum* blk_sum;
    const uint32_t* ev_word;
    const uint8_t* ev_meta;
    const uint32_t* cp_off;
    const uint32_t* cp_word;
};

hipError_t launch_tile_max_entries(const uint32_t* d_read_off, uint32_t n_reads, uint32_t T,
                                   uint32_t* d_out_max, hipStream_t stream);
hipError_t launch_sweep(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word,
                        uint32_t n_reads, uint32_t T, uint32_t ntiles, uint32_t nchunks,
                        uint32_t blocks_per_chunk, bool s_in_lds, uint32_t lds_bytes, int32_t* part_score,
                        uint32_t* part_rank, uint32_t* part_cnt, hipStream_t stream);
hipError_t launch_finalize(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word,
                           uint32_t n_reads, uint32_t nchunks, const int32_t* part_score,
                           const uint32_t* part_rank, const uint32_t* part_cnt, uint32_t* best_bfs_j,
                           int32_t* score, uint32_t* num_best, uint32_t* flags, hipStream_t stream);
hipError_t sweep_set_max_lds(uint32_t bytes);

}  // namespace wepp
