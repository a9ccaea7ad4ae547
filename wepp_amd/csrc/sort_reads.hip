// sort_reads.hip -- orders the reads that sweep the whole-tree stream by first listed position
// (keys from k_first_pos, place_kernels.hip) with rocPRIM's device radix sort, a library
// primitive.  Only used when that stream holds enough reads for a sweep to cost more than the sort.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "device_mat.hpp"
#include "fitch.hpp"

namespace wepp {

hipError_t sort_reads_temp_bytes(uint32_t n, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, SORT_KEY_BITS, nullptr);
}

hipError_t launch_sort_reads(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                             uint32_t n, void* temp, size_t temp_bytes, hipStream_t stream) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, SORT_KEY_BITS,
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
