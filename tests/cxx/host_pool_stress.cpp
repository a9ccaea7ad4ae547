// Stress test of wepp_amd/csrc/host_pool.hpp (the host workers of wepp_place_batch): rounds of 1..33 tasks, every
// task exactly once per round, no task of an earlier round after the next one has started.  Built and run by
// tests/test_host_pool.py, once plain and once under ThreadSanitizer.
#include "host_pool.hpp"

#include <atomic>
#include <cstdio>
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
    }
    std::printf("ok %ld tasks\n", total);
    return 0;
}
