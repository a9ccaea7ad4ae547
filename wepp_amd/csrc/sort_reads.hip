// sort_reads.hip -- orders the reads that sweep the whole-tree stream by first listed position
// (keys from k_first_pos, route_kernels.hip) with rocPRIM's device radix sort, a library
// primitive.  Only used when that stream holds enough reads for a sweep to cost more than the sort.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "device_mat.hpp"
#include "fitch.hpp"

namespace wepp {

hipError_t sort_reads_temp_bytes(uint32_t n, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, WALK_SORT_KEY_BITS, nullptr);
}

hipError_t launch_sort_reads(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                             uint32_t n, void* temp, size_t temp_bytes, hipStream_t stream, uint32_t key_bits) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, key_bits,
                                     stream);
}

// exclusive prefix sum of 32-bit counts (the jobs of the reads whose walk is cut into chunks)
hipError_t scan_u32_temp_bytes(uint32_t n, size_t* bytes) {
    *bytes = 0;
    return rocprim::exclusive_scan(nullptr, *bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, n,
                                   rocprim::plus<uint32_t>(), nullptr);
}

hipError_t launch_exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, void* temp, size_t temp_bytes,
                                     hipStream_t stream) {
    return rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), stream);
}

// ---- Fitch-Sankoff rows prepared on the device ---------------------------------------------
// A VCF row names tree samples by node id in file order; the kernels want them by BFS (or DFS) index,
// ascending, a node named twice keeping its later entry (usher_mapper.cpp:57-62).  Map -> segmented sort
// (one segment per row, stable) -> drop the earlier of equal neighbours (their slots go to the end of the row
// with key ~0 in a second sort, only when a row had any).
namespace {
__global__ void k_fitch_map(const uint32_t* __restrict__ var_node, uint64_t nv, uint32_t N,
                            const uint32_t* __restrict__ id2key, uint32_t* __restrict__ keys, uint32_t* __restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const uint32_t id = var_node[i];
    if (id >= N) { flags[0] = 1; keys[i] = 0; return; }
    keys[i] = id2key[id];
}
__global__ void k_fitch_dedupe(const uint32_t* __restrict__ var_off, uint32_t* __restrict__ keys, uint8_t* __restrict__ nuc,
                               uint32_t* __restrict__ flags) {
    const uint32_t row = blockIdx.x;
    const uint32_t lo = var_off[row], hi = var_off[row + 1];
    // (neighbours are read before anything of the row is overwritten: two passes with a block barrier)
    for (uint32_t base = lo; base < hi; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const bool drop = i + 1 < hi && keys[i + 1] == keys[i];
        __syncthreads();
        if (drop) { keys[i] = 0xFFFFFFFFu; nuc[i] = 0; flags[1] = 1; }
        __syncthreads();
    }
}
}  // namespace

hipError_t fitch_rows_temp_bytes(uint64_t nv, uint32_t n_sites, size_t* bytes) {
    *bytes = 0;
    return rocprim::segmented_radix_sort_pairs(nullptr, *bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                               (const uint8_t*)nullptr, (uint8_t*)nullptr, (unsigned int)nv, n_sites,
                                               (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0, 32, nullptr);
}

hipError_t launch_fitch_prepare(const uint32_t* d_var_node, const uint8_t* d_var_nuc, const uint32_t* d_var_off,
                                uint32_t n_sites, uint64_t nv, uint32_t N, const uint32_t* d_id2key, uint32_t* keys_a,
                                uint8_t* nuc_a, uint32_t* keys_b, uint8_t* nuc_b, uint32_t* d_flags, void* temp,
                                size_t temp_bytes, bool* result_in_b, hipStream_t stream) {
    *result_in_b = true;
    if (nv == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_flags, 0, 8, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fitch_map, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, d_var_node, nv, N, d_id2key, keys_a,
                       d_flags);
    e = rocprim::segmented_radix_sort_pairs(temp, temp_bytes, (const uint32_t*)keys_a, keys_b, d_var_nuc, nuc_b, (unsigned int)nv,
                                            n_sites, d_var_off, d_var_off + 1, 0, 32, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fitch_dedupe, dim3(n_sites), dim3(256), 0, stream, d_var_off, keys_b, nuc_b, d_flags);
    uint32_t h_flags[2] = {0, 0};
    e = hipMemcpyAsync(h_flags, d_flags, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    if (h_flags[0]) return hipErrorInvalidValue;          // a node id out of range
    if (h_flags[1]) {
        // some row named a node twice: the dropped slots (key ~0) move behind the row's kept entries
        e = rocprim::segmented_radix_sort_pairs(temp, temp_bytes, (const uint32_t*)keys_b, keys_a, (const uint8_t*)nuc_b, nuc_a,
                                                (unsigned int)nv, n_sites, d_var_off, d_var_off + 1, 0, 32, stream);
        *result_in_b = false;
    }
    return e;
}

// 64-bit keys / 32-bit values (the mutations emitted by the Fitch-Sankoff down pass: key = row | node)
hipError_t sort_u64_u32_temp_bytes(uint64_t n, uint32_t end_bit, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                     (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, end_bit, nullptr);
}

hipError_t launch_sort_u64_u32(const unsigned long long* keys_in, unsigned long long* keys_out, const uint32_t* vals_in,
                               uint32_t* vals_out, uint64_t n, uint32_t end_bit, void* temp, size_t temp_bytes,
                               hipStream_t stream) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, stream);
}

}  // namespace wepp
