// flat_debug.cpp -- host-only view of the flattened MAT for the CPU tests.
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/wepp_place.h"
#include "errors.hpp"
#include "flatmat.hpp"

extern "C" uint64_t wepp_debug_flatten_count(void) { return wepp::flatten_count(); }

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
#define FIELD(obj, n)                                                   \
    if (std::strcmp(name, #n) == 0) {                                   \
        *data = obj.n.data();                                           \
        *count = obj.n.size();                                          \
        *elem_bytes = (uint32_t)sizeof(obj.n[0]);                       \
        return WEPP_OK;                                                 \
    }
    // window crowns: "wc_tau" / "wc_nodes" = [windows x WC_MAX] tau (INT32_MAX: none) and node count of every crown
    if (std::strcmp(name, "wc_tau") == 0 || std::strcmp(name, "wc_nodes") == 0) {
        wepp_flat* self = const_cast<wepp_flat*>(flat);
        if (self->wc_tau.empty()) {
            self->wc_tau.assign(f.wcrowns.size() * wepp::WC_MAX, 0x7FFFFFFF);
            self->wc_nodes.assign(f.wcrowns.size() * wepp::WC_MAX, 0);
            for (size_t w = 0; w < f.wcrowns.size(); w++)
                for (size_t i = 0; i < f.wcrowns[w].size(); i++) {
                    self->wc_tau[w * wepp::WC_MAX + i] = f.wcrowns[w][i].tau;
                    self->wc_nodes[w * wepp::WC_MAX + i] = f.wcrowns[w][i].n;
                }
        }
        const bool tau = name[3] == 't';
        *data = tau ? (const void*)self->wc_tau.data() : (const void*)self->wc_nodes.data();
        *count = self->wc_tau.size();
        *elem_bytes = 4;
        return WEPP_OK;
    }
    size_t si = f.streams.size() - 1;
    if (name[0] >= '0' && name[0] <= '9' && std::strchr(name, ':')) {
        si = (size_t)std::strtoul(name, nullptr, 10);
        name = std::strchr(name, ':') + 1;
        if (si >= f.streams.size()) return wepp::set_error(WEPP_EINVAL, "stream index out of range");
    } else {
        FIELD(f, node_woff) FIELD(f, words) FIELD(f, rank2dfs) FIELD(f, dfs2bfs) FIELD(f, bfs2id) FIELD(f, dfs2id)
        FIELD(f, parent_dfs) FIELD(f, dfs_end) FIELD(f, num_leaves) FIELD(f, epp_word) FIELD(f, epp_node) FIELD(f, maxnest) FIELD(f, rank2bfs) FIELD(f, seed_sig)
    }
    // "w<i>:" selects window stream i
    const bool win = name[0] == 'w' && name[1] >= '0' && name[1] <= '9' && std::strchr(name, ':');
    if (win) {
        si = (size_t)std::strtoul(name + 1, nullptr, 10);
        name = std::strchr(name, ':') + 1;
        if (si >= f.wstreams.size()) return wepp::set_error(WEPP_EINVAL, "window stream index out of range");
    }
    // "c<w>.<i>:" selects crown i of genome window w (flatmat.hpp: wcrowns)
    const bool wcr = name[0] == 'c' && name[1] >= '0' && name[1] <= '9' && std::strchr(name, '.') && std::strchr(name, ':');
    size_t ci = 0;
    if (wcr) {
        si = (size_t)std::strtoul(name + 1, nullptr, 10);
        ci = (size_t)std::strtoul(std::strchr(name, '.') + 1, nullptr, 10);
        name = std::strchr(name, ':') + 1;
        if (si >= f.wcrowns.size() || ci >= f.wcrowns[si].size()) return wepp::set_error(WEPP_EINVAL, "window crown index out of range");
    }
    const wepp::Stream& st = wcr ? f.wcrowns[si][ci] : win ? f.wstreams[si] : f.streams[si];
    FIELD(st, ncnt)
    FIELD(st, nkey) FIELD(st, nstat) FIELD(st, blk_node0) FIELD(st, blk_eoff) FIELD(st, blk_sum) FIELD(st, ev_word)
    FIELD(st, ev_meta) FIELD(st, ev_lb) FIELD(st, cp_off) FIELD(st, cp_word)
    FIELD(st, ix_head) FIELD(st, ix_ent) FIELD(st, ix_nest) FIELD(st, ix_pre) FIELD(st, nrec) FIELD(st, rq_pre) FIELD(st, rq_suf) FIELD(st, rq_dst) FIELD(st, sp)
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
        stats->n_events = f.full().E;
        stats->n_blocks = f.full().NB;
        stats->n_leaves = f.n_leaves;
        stats->max_depth = f.max_depth;
        stats->max_position = f.max_pos;
        stats->stream_bytes = f.full().stream_bytes();
        stats->n_streams = (uint32_t)f.streams.size();
        for (const auto& wc : f.wcrowns)
            for (const wepp::Stream& st : wc) { stats->n_window_crowns++; stats->window_crown_nodes += st.n; }
        for (const wepp::Stream& st : f.wstreams) {
            stats->n_window_streams++;
            stats->n_window_streams_crown += st.ncnt.empty() ? 1u : 0u;
            stats->window_stream_nodes += st.n;
        }
        for (size_t i = 0; i < f.streams.size(); i++) {
            stats->stream_tau[i] = f.streams[i].tau;
            stats->stream_nodes[i] = f.streams[i].n;
            stats->stream_bytes_of[i] = f.streams[i].stream_bytes();
        }
        stats->window_size = wepp::WIN_SIZE;
        stats->window_stride = wepp::WIN_STRIDE;
        stats->window_uncovered_positions = f.max_pos + 1 > wepp::MAX_WINDOWS * wepp::WIN_STRIDE ? f.max_pos + 1 - wepp::MAX_WINDOWS * wepp::WIN_STRIDE : 0;
        stats->seed_chunks = f.seed_chunks;
        stats->seed_chunk_blocks = f.seed_stride;
        stats->seed_sig_bytes = (uint64_t)f.seed_sig.size() * 4;
    }
    if (cp_stride) *cp_stride = f.full().cp_stride;
    return WEPP_OK;
}

extern "C" int wepp_flat_destroy(wepp_flat_t* flat) {
    delete flat;
    return WEPP_OK;
}
