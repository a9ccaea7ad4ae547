// Stress test of wepp_amd/csrc/host_pool.hpp (the host workers of wepp_place_batch): rounds of 1..33 tasks, every
// task exactly once per round, no task of an earlier round after the next one has started.  Built and run by
// tests/test_host_pool.py, once plain and once under ThreadSanitizer.
#include "host_pool.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 20000;
    wepp::HostPool pool(7);
    std::vector<std::atomic<int>> hit(64);
    std::atomic<int> round_seen{0};
    long total = 0;
    for (int round = 0; round < rounds; round++) {
        const uint32_t n = 1 + round % 33;
        for (auto& h : hit) h = 0;
        round_seen = round;
        pool.run(n, [&](uint32_t i) {
            if (round_seen.load() != round) std::printf("task of another round\n");
            hit[i]++;
        });
        for (uint32_t i = 0; i < 64; i++)
            if (hit[i] != (i < n ? 1 : 0)) { std::printf("BAD round %d task %u ran %d times\n", round, i, (int)hit[i]); return 1; }
        total += n;
        // every 97th round: some tasks throw.  All tasks of the round still run (they reference this frame), the
        // first exception comes out of run() on this thread, and the pool works on afterwards.
        if (round % 97 == 0) {
            for (auto& h : hit) h = 0;
            bool caught = false;
            try {
                pool.run(n, [&](uint32_t i) {
                    hit[i]++;
                    if (i % 3 == 0) throw std::runtime_error("task failed");
                });
            } catch (const std::runtime_error&) { caught = true; }
            if (!caught) { std::printf("BAD round %d: the exception was lost\n", round); return 1; }
            for (uint32_t i = 0; i < 64; i++)
                if (hit[i] != (i < n ? 1 : 0)) { std::printf("BAD throwing round %d task %u ran %d times\n", round, i, (int)hit[i]); return 1; }
        }
    }
    std::printf("ok %ld tasks\n", total);
    return 0;
}
