// host_pool.hpp -- a handle's host worker threads: they check and stage the reads of wepp_place_batch and move its
// results out.  Starting and joining eight std::threads costs more than the work they share on a 1 M-read batch;
// these sleep on a condition variable between calls.  run() is called by one thread at a time (a handle is not
// re-entrant); the caller works along.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#ifdef __linux__
#include <sched.h>
#endif
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace wepp {

// Host threads this process may really use: the affinity mask (taskset, a container's cpuset) capped by the cgroup
// CPU quota (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1).  std::thread::hardware_concurrency() sees
// neither: eight ranks on a node each sizing a pool from it start 8 x 15 workers on the cores of one container.
inline uint32_t usable_host_threads() {
    uint32_t n = std::max(1u, std::thread::hardware_concurrency());
#ifdef __linux__
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::max(1, CPU_COUNT(&set));
    auto quota = [&](const char* path_quota, const char* path_period) {
        FILE* f = std::fopen(path_quota, "r");
        if (!f) return;
        char a[64] = {0}, b[64] = {0};
        long long q = -1, p = -1;
        if (path_period == nullptr) {                    // cgroup v2: "max 100000" or "<quota> <period>"
            if (std::fscanf(f, "%63s %63s", a, b) == 2 && std::strcmp(a, "max") != 0) { q = std::atoll(a); p = std::atoll(b); }
        } else if (std::fscanf(f, "%lld", &q) == 1) {
            FILE* g = std::fopen(path_period, "r");
            if (g) { if (std::fscanf(g, "%lld", &p) != 1) p = -1; std::fclose(g); }
        }
        std::fclose(f);
        if (q > 0 && p > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (q + p - 1) / p));
    };
    quota("/sys/fs/cgroup/cpu.max", nullptr);
    quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
#endif
    return n;
}

class HostPool {
  public:
    explicit HostPool(uint32_t n_workers) {
        for (uint32_t i = 0; i < n_workers; i++) threads_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    HostPool(const HostPool&) = delete;
    HostPool& operator=(const HostPool&) = delete;
    uint32_t workers() const { return (uint32_t)threads_.size(); }

    // fn(i) for i in [0, n), shared between the workers and the caller; returns when all are done.  An exception
    // thrown by fn (on any thread) is kept -- the first one --, the remaining tasks still run to the end (they may
    // reference the caller's locals, which must outlive them), and run() rethrows it on the calling thread.
    void run(uint32_t n, const std::function<void(uint32_t)>& fn) {
        if (n == 0) return;
        start(n, fn);
        finish();
    }
    // The same in two halves: start() hands the tasks to the workers and returns; the caller does something else (the
    // GPU launches of a pipelined wepp_place_batch) and then calls finish(), which works along on what is left and
    // waits for the rest.  Tasks are taken in ascending order of i.  `fn` must stay alive until finish() returns.
    void start(uint32_t n, const std::function<void(uint32_t)>& fn) {
        {
            // (a worker that woke up late for the previous round may still be looking at its counters)
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [this] { return active_ == 0; });
            fn_ = &fn;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            done_ = 0;
            error_ = nullptr;
            gen_++;
        }
        cv_.notify_all();
    }
    void finish() {
        const uint32_t mine = work();
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> lk(m_);
            done_ += mine;
            cv_done_.wait(lk, [this] { return done_ == n_ && active_ == 0; });
            fn_ = nullptr;
            err = error_;
            error_ = nullptr;
        }
        if (err) std::rethrow_exception(err);
    }

  private:
    uint32_t work() {
        uint32_t mine = 0;
        for (;;) {
            const uint32_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_) break;
            try {
                (*fn_)(i);
            } catch (...) {
                std::lock_guard<std::mutex> lk(m_);
                if (!error_) error_ = std::current_exception();
            }
            mine++;
        }
        return mine;
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                active_++;          // n_, fn_ and next_ stay put while a worker is active
            }
            const uint32_t mine = work();
            {
                std::lock_guard<std::mutex> lk(m_);
                done_ += mine;
                active_--;
                if (active_ == 0) cv_done_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    const std::function<void(uint32_t)>* fn_ = nullptr;
    uint32_t n_ = 0, done_ = 0, active_ = 0;
    std::atomic<uint32_t> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
    std::exception_ptr error_;    // first exception of the current round (guarded by m_)
};

}  // namespace wepp
