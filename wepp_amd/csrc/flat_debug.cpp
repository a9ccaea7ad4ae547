// flat_debug.cpp -- host-only view of the flattened MAT for the CPU tests.
#include <cstring>
#include <new>
#include <string>

#include "../../include/wepp_place.h"
#include "errors.hpp"
#include "flatmat.hpp"

struct wepp_flat {
    wepp::FlatMAT f;
};

extern "C" int wepp_flat_create(const wepp_tree_desc* tree, wepp_flat_t** out) {
    if (!tree || !out) return wepp::set_error(WEPP_EINVAL, "null argument");
    *out = nullptr;
    wepp_flat* h = new (std::nothrow) wepp_flat();
    if (!h) return wepp::set_error(WEPP_ENOMEM, "out of host memory");
    std::string err;
    int rc;
    try {
        rc = wepp::flatten_tree(*tree, h->f, err);
    } catch (const std::bad_alloc&) {
        delete h;
        return wepp::set_error(WEPP_ENOMEM, "out of host memory while flattening the tree");
    }
    if (rc != WEPP_OK) { delete h; return wepp::set_error(rc, err); }
    *out = h;
    return WEPP_OK;
}

extern "C" int wepp_flat_get(const wepp_flat_t* flat, const char* name, const void** data, uint64_t* count,
                             uint32_t* elem_bytes) {
    if (!flat || !name || !data || !count || !elem_bytes) return wepp::set_error(WEPP_EINVAL, "null argument");
    const wepp::FlatMAT& f = flat->f;
#define FIELD(n)                                                        \
    if (std::strcmp(name, #n) == 0) {                                   \
        *data = f.n.data();                                             \
        *count = f.n.size();                                            \
        *elem_bytes = (uint32_t)sizeof(f.n[0]);                         \
        return WEPP_OK;                                                 \
    }
    FIELD(node_woff) FIELD(words) FIELD(nkey) FIELD(nstat) FIELD(rank2dfs) FIELD(dfs2bfs) FIELD(bfs2id)
    FIELD(dfs2id) FIELD(parent_dfs) FIELD(dfs_end) FIELD(num_leaves) FIELD(blk_node0) FIELD(blk_eoff)
    FIELD(blk_sum) FIELD(ev_word) FIELD(ev_meta) FIELD(cp_off) FIELD(cp_word)
#undef FIELD
    return wepp::set_error(WEPP_EINVAL, std::string("unknown flat field: ") + name);
}

extern "C" int wepp_flat_scalars(const wepp_flat_t* flat, wepp_mat_stats* stats, uint32_t* cp_stride) {
    if (!flat) return wepp::set_error(WEPP_EINVAL, "null argument");
    const wepp::FlatMAT& f = flat->f;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->n_nodes = f.N;
        stats->n_mutations = f.M;
        stats->n_masked = f.n_masked;
        stats->n_events = f.E;
        stats->n_blocks = f.NB;
        stats->n_leaves = f.n_leaves;
        stats->max_depth = f.max_depth;
        stats->max_position = f.max_pos;
        stats->stream_bytes = 4ull * f.E + (uint64_t)f.NB * (sizeof(wepp::BlkSum) + 4);
    }
    if (cp_stride) *cp_stride = f.cp_stride;
    return WEPP_OK;
}

extern "C" int wepp_flat_destroy(wepp_flat_t* flat) {
    delete flat;
    return WEPP_OK;
}
