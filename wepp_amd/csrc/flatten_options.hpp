// flatten_options.hpp -- the environment helpers and what a flatten reads from the environment (flatmat.cpp), once
// per flatten_tree call.  HIP-free: flatmat.cpp is also built with g++ under sanitizers (tests/test_flatten_sanitized.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "flatmat.hpp"

namespace wepp {

namespace env {
inline bool is_set(const char* name) { return std::getenv(name) != nullptr; }
inline bool flag(const char* name, bool dflt) {            // "0" switches off, anything else on
    const char* v = std::getenv(name);
    return v ? v[0] != '0' : dflt;
}
inline uint64_t u64(const char* name, uint64_t dflt, uint64_t lo, uint64_t hi) {   // negative / malformed values fall to `lo`
    const char* v = std::getenv(name);
    if (!v) return dflt;
    const long long x = std::atoll(v);
    return x < (long long)lo ? lo : std::min<uint64_t>((uint64_t)x, hi);
}
}  // namespace env

// what a flatten reads from the environment (flatmat.cpp), once per flatten_tree call
struct FlattenOptions {
    uint32_t ix_pre_min_nodes = IX_PRE_MIN_NODES;   // WEPP_IX_PRE_MIN_NODES: streams from this size carry the per-entry pre-test bytes
    bool win_whole_tree = false;                    // WEPP_WIN_WHOLE_TREE=1: every window stream from the whole tree (pseudo-nodes)
    uint32_t seed_chunk_blocks = SEED_CHUNK_BLOCKS; // WEPP_SEED_CHUNK_BLOCKS: blocks of the whole-tree stream per seed chunk
    static FlattenOptions from_env() {
        FlattenOptions o;
        o.ix_pre_min_nodes = (uint32_t)env::u64("WEPP_IX_PRE_MIN_NODES", IX_PRE_MIN_NODES, 0, 0xFFFFFFFFu);
        o.win_whole_tree = env::is_set("WEPP_WIN_WHOLE_TREE") && env::flag("WEPP_WIN_WHOLE_TREE", false);
        o.seed_chunk_blocks = (uint32_t)env::u64("WEPP_SEED_CHUNK_BLOCKS", SEED_CHUNK_BLOCKS, 1, 1024);
        return o;
    }
};

}  // namespace wepp
