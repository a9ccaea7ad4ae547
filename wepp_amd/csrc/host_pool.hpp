// host_pool.hpp -- a handle's host worker threads: they check and stage the reads of wepp_place_batch and move its
// results out.  Starting and joining eight std::threads costs more than the work they share on a 1 M-read batch;
// these sleep on a condition variable between calls.  run() is called by one thread at a time (a handle is not
// re-entrant); the caller works along.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace wepp {

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
