// staged_copy.hpp -- device -> pageable host copies of large outputs.
// hipMemcpy into pageable memory stages through a pinned buffer on ONE host thread (~10 GB/s); here the DMA
// fills two pinned buffers alternately while several host threads move the previous chunk out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace wepp {

constexpr size_t STAGE_CHUNK = 32ull << 20;   // bytes per pinned buffer
constexpr size_t STAGE_MIN = 64ull << 20;     // smaller copies go the plain way
constexpr unsigned STAGE_THREADS = 6;         // host threads moving a chunk out of the pinned buffer

// synchronous with respect to `stream`: returns when `dst` holds the data
inline hipError_t d2h_staged(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (bytes < STAGE_MIN) {
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
        return e != hipSuccess ? e : hipStreamSynchronize(stream);
    }
    // staging state per device (events belong to a device): one staged copy at a time per device and process
    struct Stage { std::mutex mu; char* pin[2] = {nullptr, nullptr}; hipEvent_t ev[2]; };
    constexpr int MAX_DEV = 64;
    static Stage stages[MAX_DEV];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEV) {
        e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
        return e != hipSuccess ? e : hipStreamSynchronize(stream);
    }
    Stage& st = stages[dev];
    std::lock_guard<std::mutex> lock(st.mu);
    char** pin = st.pin;
    hipEvent_t* ev = st.ev;
    if (!pin[0]) {
        char *a = nullptr, *b = nullptr;
        if ((e = hipHostMalloc((void**)&a, STAGE_CHUNK, hipHostMallocPortable)) != hipSuccess) return e;
        if ((e = hipHostMalloc((void**)&b, STAGE_CHUNK, hipHostMallocPortable)) != hipSuccess) { (void)hipHostFree(a); return e; }
        if ((e = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming)) != hipSuccess) {
            (void)hipHostFree(a); (void)hipHostFree(b);
            return e;
        }
        pin[0] = a; pin[1] = b;
    }
    const size_t nch = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
    auto issue = [&](size_t c) -> hipError_t {
        const size_t off = c * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
        hipError_t x = hipMemcpyAsync(pin[c & 1], (const char*)src + off, len, hipMemcpyDeviceToHost, stream);
        return x != hipSuccess ? x : hipEventRecord(ev[c & 1], stream);
    };
    if ((e = issue(0)) != hipSuccess) return e;
    const unsigned nthr = std::max(1u, std::min(STAGE_THREADS, std::thread::hardware_concurrency()));
    for (size_t c = 0; c < nch; c++) {
        // chunk c + 1 goes into the other buffer, whose previous content (chunk c - 1) was moved out in the last round
        if (c + 1 < nch && (e = issue(c + 1)) != hipSuccess) return e;
        if ((e = hipEventSynchronize(ev[c & 1])) != hipSuccess) return e;
        const size_t off = c * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
        char* d = (char*)dst + off;
        const char* s = pin[c & 1];
        std::vector<std::thread> pool;
        const size_t per = ((len + nthr - 1) / nthr + 4095) & ~(size_t)4095;
        for (unsigned t = 1; t < nthr; t++) {
            const size_t a = std::min(len, (size_t)t * per), b = std::min(len, (size_t)(t + 1) * per);
            if (b > a) pool.emplace_back([=]() { std::memcpy(d + a, s + a, b - a); });
        }
        std::memcpy(d, s, std::min(len, per));
        for (auto& th : pool) th.join();
    }
    return hipSuccess;
}

}  // namespace wepp
