// capi.cpp -- extern "C" shim: handle lifetime, HBM residency, launches.
//
// Replaces the per-sample body of usher_common (src/usher_common.cpp:339-446):
// the BFS expansion, the N empty vectors per sample, both tbb::parallel_for
// passes over mapper2_body and the locked argmin become one batch call.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "handle.hpp"
#include "host_pool.hpp"

namespace {

// handles alive in this process: a handle's host workers are its share of the cores the process may use
std::atomic<uint32_t> g_handles_alive{0};

void release(wepp_mat* h) {
    if (!h) return;
    g_handles_alive.fetch_sub(1, std::memory_order_relaxed);
    (void)hipSetDevice(h->device);
    for (void* p : h->allocs) (void)hipFree(p);
    for (PlaceLane& L : h->lane) {
        if (L.ws) (void)hipFree(L.ws);
        if (L.ws2) (void)hipFree(L.ws2);
        if (L.d_info) (void)hipFree(L.d_info);
        if (L.h_info) (void)hipHostFree(L.h_info);
        for (uint32_t i = 0; i < MAX_STREAMS; i++) {
            if (L.side[i]) (void)hipStreamDestroy(L.side[i]);
            if (L.join_ev[i]) (void)hipEventDestroy(L.join_ev[i]);
        }
        if (L.fork_ev) (void)hipEventDestroy(L.fork_ev);
        if (L.route_ev) (void)hipEventDestroy(L.route_ev);
    }
    if (h->io_in) (void)hipFree(h->io_in);
    if (h->io_out) (void)hipFree(h->io_out);
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->epp_ws) (void)hipFree(h->epp_ws);
    if (h->d_work) (void)hipFree(h->d_work);
    if (h->d_seed_heavy) (void)hipFree(h->d_seed_heavy);
    for (uint32_t i = 0; i < wepp_mat::kRing; i++) {
        if (h->ev0[i]) (void)hipEventDestroy(h->ev0[i]);
        if (h->ev1[i]) (void)hipEventDestroy(h->ev1[i]);
    }
    for (uint32_t i = 0; i < wepp_mat::kPipeMax; i++) {
        if (h->pipe_up[i]) (void)hipEventDestroy(h->pipe_up[i]);
        if (h->pipe_done[i]) (void)hipEventDestroy(h->pipe_done[i]);
    }
    for (uint32_t i = 0; i < 4 * wepp_mat::kPipeMax; i++)
        if (h->pipe_out[i]) (void)hipEventDestroy(h->pipe_out[i]);
    for (hipStream_t st : h->pipe_h2d) if (st) (void)hipStreamDestroy(st);
    for (hipStream_t st : h->pipe_d2h) if (st) (void)hipStreamDestroy(st);
    for (hipStream_t st : h->pipe_compute)
        if (st) (void)hipStreamDestroy(st);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->d_plan_of) (void)hipFree(h->d_plan_of);
    if (h->d_wsid_of) (void)hipFree(h->d_wsid_of);
    delete h;
}

}  // namespace

namespace {
// the device half of wepp_mat_create: the flat image `f` (flatmat.hpp) copied into HBM of `device`, plus the handle's
// streams, events and counters.  The image is only read: one image serves any number of devices / handles.
int upload_flat(const FlatMAT& f, int device, wepp_mat_t** out) {
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return set_error(WEPP_EDEVICE, "no HIP device available (the placement engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return set_error(WEPP_EINVAL, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    wepp_mat* h = new (std::nothrow) wepp_mat();
    if (!h) return set_error(WEPP_ENOMEM, "out of host memory");
    g_handles_alive.fetch_add(1, std::memory_order_relaxed);
    h->device = device;
    h->bfs2id = f.bfs2id;
    h->dfs2id = f.dfs2id;
    h->stats.n_nodes = f.N;
    h->stats.n_mutations = f.M;
    h->stats.n_masked = f.n_masked;
    h->stats.n_events = f.full().E;
    h->stats.n_blocks = f.full().NB;
    h->stats.n_leaves = f.n_leaves;
    h->stats.max_depth = f.max_depth;
    h->stats.max_position = f.max_pos;
    h->stats.stream_bytes = f.full().stream_bytes();
    h->stats.n_streams = (uint32_t)f.streams.size();

    DevMAT& d = h->dev;
    d.N = f.N;
    d.max_pos = f.max_pos;
    d.bm_words = 1;
    while (d.bm_words < (f.max_pos >> 5) + 1) d.bm_words <<= 1;   // power of two (k_sweep masks the index)
    d.n_streams = (uint32_t)f.streams.size();
    d.root_base = f.root_base;
    h->tun = PlaceTunables::from_env();
    d.walk_eager_nodes = h->tun.walk_eager_nodes;
    int rc;
#define UP(dst, vec) if ((rc = upload(h, vec, &dst)) != WEPP_OK) { release(h); return rc; }
    // (inside up_stream a failure returns the code; the caller releases the handle)
    UP(d.node_woff, f.node_woff) UP(d.words, f.words) UP(d.nstat, f.nstat) UP(d.rank2dfs, f.rank2dfs)
    UP(d.dfs2bfs, f.dfs2bfs) UP(d.rank2bfs, f.rank2bfs)
    {
        std::vector<uint32_t> bfs2dfs(f.N);
        for (uint32_t k = 0; k < f.N; k++) bfs2dfs[f.dfs2bfs[k]] = k;
        UP(d.bfs2dfs, bfs2dfs)
    }
    UP(d.parent_dfs, f.parent_dfs)
    UP(d.maxnest, f.maxnest)
    UP(h->epp_word, f.epp_word) UP(h->epp_node, f.epp_node)
    h->epp_events = f.epp_word.size();
    auto up_stream = [&](const Stream& st, uint32_t tier, bool eager, DevStream& ds) -> int {
        ds = DevStream{};
        ds.n = st.n;
        ds.NB = st.NB;
        ds.cp_stride = st.cp_stride;
        ds.ncp = (uint32_t)st.cp_off.size() - 1;
        ds.eager = eager ? 1u : 0u;
        ds.tier = tier;
        ds.e_pad = (uint32_t)st.E;
        UP(ds.nkey, st.nkey) UP(ds.nstat, st.nstat) UP(ds.blk_node0, st.blk_node0) UP(ds.blk_eoff, st.blk_eoff)
        UP(ds.ev_meta, st.ev_meta) UP(ds.cp_off, st.cp_off) UP(ds.cp_word, st.cp_word)
        if (!st.ncnt.empty()) UP(ds.ncnt, st.ncnt)
        // the sweep loads two events per lane from every block start and the summaries of the next
        // three blocks without looking at the stream's end: padding behind the last event / block
        std::vector<uint32_t> evw(st.ev_word);
        evw.resize(evw.size() + EV_TAIL_PAD, W_PAD);
        std::vector<uint8_t> evl(st.ev_lb);
        evl.resize(evl.size() + EV_TAIL_PAD, 255);
        std::vector<BlkSum> sums(st.blk_sum);
        sums.resize(sums.size() + SUM_TAIL_PAD, sums.empty() ? BlkSum{} : sums.back());
        UP(ds.ev_word, evw) UP(ds.ev_lb, evl) UP(ds.blk_sum, sums)
        return WEPP_OK;
    };
    for (size_t i = 0; i < f.wstreams.size(); i++) {
        DevStream ds;
        if ((rc = up_stream(f.wstreams[i], (uint32_t)(f.streams.size() - 1), h->tun.win_eager, ds)) != WEPP_OK) return rc;
        h->wstreams.push_back(ds);
        h->wstream_bytes.push_back(f.wstreams[i].stream_bytes());
        if (i < MAX_WINDOWS) d.win_n[i] = f.wstreams[i].ncnt.empty() ? f.wstreams[i].n : 0xFFFFFFFFu;   // (pseudo-nodes: the whole tree)
        h->stats.n_window_streams++;
        h->stats.n_window_streams_crown += f.wstreams[i].ncnt.empty() ? 1u : 0u;
        h->stats.window_stream_nodes += f.wstreams[i].n;
    }
    d.n_windows = (uint32_t)f.wstreams.size();
    if (!h->wstreams.empty()) UP(h->d_wstreams, h->wstreams)
    for (size_t i = 0; i < f.streams.size(); i++) {
        const Stream& st = f.streams[i];
        DevStream ds;
        if ((rc = up_stream(st, (uint32_t)i, i + 1 < f.streams.size(), ds)) != WEPP_OK) return rc;
        h->streams.push_back(ds);
        h->stream_bytes.push_back(st.stream_bytes());
        h->walks.push_back(DevWalk{});      // (a view into the walk arena, filled in below)
        d.tau[i] = st.tau;
        h->stats.stream_tau[i] = st.tau;
        h->stats.stream_nodes[i] = st.n;
        h->stats.stream_bytes_of[i] = st.stream_bytes();
    }
    // ---- the WALK ARENA: the walk structures (position index, range-query tables) of every stream -- the window crowns
    // first, then the tree-wide streams -- concatenated array by array into ONE allocation per array (device_mat.hpp:
    // WcInfo = the offsets of a stream's slices; entry indices are absolute in the arena).  Which stream a read walks is a
    // per-READ value (k_route: wsid), so one launch of k_walk serves the reads of all streams, sized without a look at
    // the routing counters. ----
    {
        const size_t n_wc = f.wcrowns.size() * WC_MAX;
        std::vector<WcInfo> info(n_wc + f.streams.size());
        std::vector<IxHead> a_head;
        std::vector<IxEnt> a_ent;
        std::vector<uint8_t> a_nest, a_sp;
        std::vector<NodeRec> a_nrec;
        std::vector<SegNode> a_pre, a_suf, a_dst;
        size_t tot_n = 0, tot_ent = 0, tot_head = 0, tot_sp = 0, tot_dst = 0;
        for (const auto& wc : f.wcrowns)
            for (const Stream& st : wc) {
                tot_n += st.n; tot_ent += st.ix_ent.size(); tot_head += st.ix_head.size(); tot_sp += st.sp.size(); tot_dst += st.rq_dst.size();
            }
        const size_t wc_n = tot_n, wc_ent = tot_ent, wc_head = tot_head, wc_sp = tot_sp, wc_dst = tot_dst;   // (the window crowns' share)
        for (const Stream& st : f.streams) {
            tot_n += st.n; tot_ent += st.ix_ent.size(); tot_head += st.ix_head.size(); tot_sp += st.sp.size(); tot_dst += st.rq_dst.size();
        }
        if (tot_ent >= 0xFFFFFFF0ull || tot_n >= 0xFFFFFFF0ull || tot_head >= 0xFFFFFFF0ull || tot_dst >= 0xFFFFFFF0ull) {
            release(h);
            return set_error(WEPP_ELIMIT, "the walk structures exceed 2^32 index entries");
        }
        a_head.reserve(wc_head); a_ent.reserve(wc_ent); a_nest.reserve(wc_head); a_sp.reserve(wc_sp);
        a_nrec.reserve(wc_n); a_pre.reserve(wc_n); a_suf.reserve(wc_n); a_dst.reserve(wc_dst);
        for (size_t w = 0; w < f.wcrowns.size(); w++)
            for (size_t i = 0; i < f.wcrowns[w].size(); i++) {
                const Stream& st = f.wcrowns[w][i];
                WcInfo& wi = info[w * WC_MAX + i];
                const uint32_t ent_off = (uint32_t)a_ent.size();
                wi.n = st.n;
                wi.rq_blocks = st.rq_blocks;
                wi.last_ent = ent_off + (uint32_t)st.ix_ent.size() - 1;
                wi.has_pre = st.ix_pre.empty() ? 0u : st.ix_pre[0];
                wi.node_off = (uint32_t)a_nrec.size();
                wi.head_off = (uint32_t)a_head.size();
                wi.nest_off = (uint32_t)a_nest.size();
                wi.dst_off = (uint32_t)a_dst.size();
                wi.sp_off = a_sp.size();
                wi.tau = st.tau;
                wi.whole = st.whole;
                wi.whole_bfs = f.rank2bfs[st.whole.rank < f.N ? st.whole.rank : 0u];
                for (IxHead hd : st.ix_head) { hd.off += ent_off; a_head.push_back(hd); }
                for (IxEnt e : st.ix_ent) { if (e.up != IX_NONE) e.up += ent_off; a_ent.push_back(e); }
                a_nest.insert(a_nest.end(), st.ix_nest.begin(), st.ix_nest.end());
                a_nrec.insert(a_nrec.end(), st.nrec.begin(), st.nrec.end());
                a_pre.insert(a_pre.end(), st.rq_pre.begin(), st.rq_pre.end());
                a_suf.insert(a_suf.end(), st.rq_suf.begin(), st.rq_suf.end());
                a_dst.insert(a_dst.end(), st.rq_dst.begin(), st.rq_dst.end());
                a_sp.insert(a_sp.end(), st.sp.begin(), st.sp.end());
                h->wc_nodes += st.n;
                h->wc_count++;
            }
        // one device allocation per array; the window crowns' (rebased) concatenation goes to its front, every tree-wide
        // stream's arrays are copied from the image into their slices, the entry indices of a slice rebased by a kernel
        DevWalk arena{};                           // (n = 0: never a stream of its own)
        auto dev_array = [&](auto*& out, size_t count, const auto& front) -> int {
            using T = std::remove_cv_t<std::remove_reference_t<decltype(front[0])>>;
            void* p = nullptr;
            const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
            hipError_t e2 = hipMalloc(&p, bytes);
            if (e2 != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e2));
            h->allocs.push_back(p);
            h->stats.device_bytes += bytes;
            if (!front.empty()) HIP_TRY(hipMemcpy(p, front.data(), front.size() * sizeof(T), hipMemcpyHostToDevice));
            out = (T*)p;
            return WEPP_OK;
        };
        IxHead* g_head; IxEnt* g_ent; uint8_t *g_nest, *g_sp; NodeRec* g_nrec; SegNode *g_pre, *g_suf, *g_dst;
#define ARR(ptr, count, front) if ((rc = dev_array(ptr, count, front)) != WEPP_OK) { release(h); return rc; }
        ARR(g_head, tot_head, a_head) ARR(g_ent, tot_ent, a_ent) ARR(g_nest, tot_head, a_nest) ARR(g_sp, tot_sp, a_sp)
        ARR(g_nrec, tot_n, a_nrec) ARR(g_pre, tot_n, a_pre) ARR(g_suf, tot_n, a_suf) ARR(g_dst, tot_dst, a_dst)
#undef ARR
        arena.ix_head = g_head; arena.ix_ent = g_ent; arena.ix_nest = g_nest; arena.sp = g_sp;
        arena.nrec = g_nrec; arena.rq_pre = g_pre; arena.rq_suf = g_suf; arena.rq_dst = g_dst;
        {
            size_t o_n = wc_n, o_ent = wc_ent, o_head = wc_head, o_sp = wc_sp, o_dst = wc_dst;
            for (size_t i = 0; i < f.streams.size(); i++) {
                const Stream& st = f.streams[i];
                WcInfo& wi = info[n_wc + i];
                wi.n = st.n;
                wi.rq_blocks = st.rq_blocks;
                wi.last_ent = (uint32_t)(o_ent + st.ix_ent.size() - 1);
                wi.has_pre = st.ix_pre.empty() ? 0u : st.ix_pre[0];
                wi.node_off = (uint32_t)o_n; wi.head_off = (uint32_t)o_head; wi.nest_off = (uint32_t)o_head; wi.dst_off = (uint32_t)o_dst;
                wi.sp_off = o_sp;
                wi.tau = st.tau;
                wi.whole = st.whole;
                wi.whole_bfs = f.rank2bfs[st.whole.rank < f.N ? st.whole.rank : 0u];
#define CP(dst, off, vec) if (!vec.empty()) HIP_TRY(hipMemcpy(dst + off, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice));
                CP(g_head, o_head, st.ix_head) CP(g_ent, o_ent, st.ix_ent) CP(g_nest, o_head, st.ix_nest) CP(g_sp, o_sp, st.sp)
                CP(g_nrec, o_n, st.nrec) CP(g_pre, o_n, st.rq_pre) CP(g_suf, o_n, st.rq_suf) CP(g_dst, o_dst, st.rq_dst)
#undef CP
                if (o_ent) HIP_TRY(launch_rebase_index(g_head + o_head, (uint32_t)st.ix_head.size(), g_ent + o_ent, (uint32_t)st.ix_ent.size(), (uint32_t)o_ent, nullptr));
                // the stream's own view (plan-wise users: k_route, the chunked walks): entry indices are absolute, so the
                // entries are addressed from the arena's base; positions and nodes are the stream's own
                DevWalk& dw = h->walks[i];
                dw.n = st.n; dw.rq_blocks = st.rq_blocks; dw.last_ent = wi.last_ent; dw.has_pre = wi.has_pre; dw.whole = st.whole;
                dw.ix_head = g_head + o_head; dw.ix_ent = g_ent; dw.ix_nest = g_nest + o_head; dw.nrec = g_nrec + o_n;
                dw.rq_pre = g_pre + o_n; dw.rq_suf = g_suf + o_n; dw.rq_dst = g_dst + o_dst; dw.sp = g_sp + o_sp;
                o_n += st.n; o_ent += st.ix_ent.size(); o_head += st.ix_head.size(); o_sp += st.sp.size(); o_dst += st.rq_dst.size();
            }
            HIP_TRY(hipDeviceSynchronize());
        }
        d.tw_base = (uint32_t)n_wc;
        // ... and their sweep streams, for the reads that cannot walk (k_sweep_arena): the same kind of arena; every
        // stream keeps its own tail padding (the sweep loads past a block's events and prefetches summaries)
        {
            std::vector<int64_t> s_nkey;
            std::vector<uint32_t> s_nstat, s_node0, s_eoff, s_evw, s_cpo, s_cpw;
            std::vector<uint8_t> s_meta, s_lb;
            std::vector<BlkSum> s_sum;
            struct Off { size_t nkey, node0, eoff, sum, ev, cpo, cpw; };
            std::vector<Off> offs;
            for (const auto& wc : f.wcrowns)
                for (const Stream& st : wc) {
                    offs.push_back(Off{s_nkey.size(), s_node0.size(), s_eoff.size(), s_sum.size(), s_evw.size(), s_cpo.size(), s_cpw.size()});
                    s_nkey.insert(s_nkey.end(), st.nkey.begin(), st.nkey.end());
                    s_nstat.insert(s_nstat.end(), st.nstat.begin(), st.nstat.end());
                    s_node0.insert(s_node0.end(), st.blk_node0.begin(), st.blk_node0.end());
                    s_eoff.insert(s_eoff.end(), st.blk_eoff.begin(), st.blk_eoff.end());
                    s_sum.insert(s_sum.end(), st.blk_sum.begin(), st.blk_sum.end());
                    s_sum.resize(s_sum.size() + SUM_TAIL_PAD, st.blk_sum.empty() ? BlkSum{} : st.blk_sum.back());
                    s_evw.insert(s_evw.end(), st.ev_word.begin(), st.ev_word.end());
                    s_evw.resize(s_evw.size() + EV_TAIL_PAD, W_PAD);
                    s_meta.insert(s_meta.end(), st.ev_meta.begin(), st.ev_meta.end());
                    s_meta.resize(s_meta.size() + EV_TAIL_PAD, 0);
                    s_lb.insert(s_lb.end(), st.ev_lb.begin(), st.ev_lb.end());
                    s_lb.resize(s_lb.size() + EV_TAIL_PAD, 255);
                    s_cpo.insert(s_cpo.end(), st.cp_off.begin(), st.cp_off.end());
                    s_cpw.insert(s_cpw.end(), st.cp_word.begin(), st.cp_word.end());
                }
            const int64_t* d_nkey; const uint32_t *d_nstat, *d_node0, *d_eoff, *d_evw, *d_cpo, *d_cpw;
            const uint8_t *d_meta, *d_lb; const BlkSum* d_sum;
            UP(d_nkey, s_nkey) UP(d_nstat, s_nstat) UP(d_node0, s_node0) UP(d_eoff, s_eoff) UP(d_sum, s_sum) UP(d_evw, s_evw)
            UP(d_meta, s_meta) UP(d_lb, s_lb) UP(d_cpo, s_cpo) UP(d_cpw, s_cpw)
            std::vector<DevStream> ws(f.wcrowns.size() * WC_MAX, DevStream{});
            size_t k = 0;
            for (size_t w = 0; w < f.wcrowns.size(); w++)
                for (size_t i = 0; i < f.wcrowns[w].size(); i++, k++) {
                    const Stream& st = f.wcrowns[w][i];
                    DevStream& ds = ws[w * WC_MAX + i];
                    ds.n = st.n; ds.NB = st.NB; ds.cp_stride = st.cp_stride; ds.ncp = (uint32_t)st.cp_off.size() - 1;
                    ds.eager = 1u; ds.tier = WC_SLOT; ds.e_pad = (uint32_t)st.E;
                    ds.nkey = d_nkey + offs[k].nkey; ds.nstat = d_nstat + offs[k].nkey;
                    ds.blk_node0 = d_node0 + offs[k].node0; ds.blk_eoff = d_eoff + offs[k].eoff; ds.blk_sum = d_sum + offs[k].sum;
                    ds.ev_word = d_evw + offs[k].ev; ds.ev_meta = d_meta + offs[k].ev; ds.ev_lb = d_lb + offs[k].ev;
                    ds.cp_off = d_cpo + offs[k].cpo; ds.cp_word = d_cpw + offs[k].cpw;
                }
            UP(d.wc_streams, ws)
        }
        UP(d.wc_info, info)
        d.wc_windows = (uint32_t)f.wcrowns.size();
        h->stats.n_window_crowns = h->wc_count;
        h->stats.window_crown_nodes = h->wc_nodes;
        h->walks.resize(MAX_STREAMS, DevWalk{});
        h->walks[WC_SLOT] = arena;
    }
    UP(d.walks, h->walks)
    // seed signatures (flatmat.hpp): whole-genome samples
    d.seed_sig = nullptr;
    d.seed_stride = d.seed_chunks = d.seed_row_words = 0;
    if (!f.seed_sig.empty()) {
        UP(d.seed_sig, f.seed_sig)
        d.seed_stride = f.seed_stride;
        d.seed_chunks = f.seed_chunks;
        d.seed_row_words = f.seed_row_words;
    }
    h->use_seeds = h->tun.seed ? 1 : 0;
    h->stats.seed_chunks = d.seed_chunks;
    h->stats.seed_chunk_blocks = d.seed_stride;
    h->stats.seed_sig_bytes = (uint64_t)f.seed_sig.size() * 4;
    h->stats.window_size = WIN_SIZE;
    h->stats.window_stride = WIN_STRIDE;
    h->stats.window_uncovered_positions = f.max_pos + 1 > MAX_WINDOWS * WIN_STRIDE ? f.max_pos + 1 - MAX_WINDOWS * WIN_STRIDE : 0;
#undef UP
    {
        const bool walk_on = h->tun.walk;
        // k_walk keeps an open interval as (subtree end << WALK_DELTA_BITS) | delta in one dword: a stream of
        // 2^(32 - WALK_DELTA_BITS) nodes or more is left to the sweeps (device_mat.hpp)
        h->walk_ok = 1;
        for (const Stream& st : f.streams)
            if ((uint64_t)st.n >= (1ull << (32 - WALK_DELTA_BITS))) h->walk_ok = 0;
        h->use_walk = (walk_on && h->walk_ok) ? 1 : 0;
    }
    // (two arrays of WALK_COUNTERS slots: loop iterations of the walks' waves, bytes their lanes asked memory for)
    e = hipMalloc((void**)&h->d_work, wepp_mat::D_WORK_BYTES);
    if (e == hipSuccess) e = hipMemset(h->d_work, 0, wepp_mat::D_WORK_BYTES);
    if (e != hipSuccess) { release(h); return hip_fail(e, "handle setup"); }
    for (PlaceLane& L : h->lane) {
        if (e == hipSuccess) e = hipMalloc((void**)&L.d_info, (2 * TI_WORDS + ROUTE_BLOCKS * MAX_PLANS) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(L.d_info, 0, 2 * TI_WORDS * sizeof(uint32_t));
        if (e == hipSuccess) e = hipHostMalloc((void**)&L.h_info, TI_WORDS * sizeof(uint32_t), hipHostMallocDefault);
        for (uint32_t i = 0; i < MAX_STREAMS && e == hipSuccess; i++) {
            e = hipStreamCreateWithFlags(&L.side[i], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&L.join_ev[i], hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&L.fork_ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&L.route_ev, hipEventDisableTiming);
    }
    for (uint32_t i = 0; i < wepp_mat::kRing && e == hipSuccess; i++) {
        e = hipEventCreate(&h->ev0[i]);
        if (e == hipSuccess) e = hipEventCreate(&h->ev1[i]);
    }
    if (e == hipSuccess) e = sweep_set_max_lds(160 * 1024);
    if (e == hipSuccess) e = seed_set_max_lds(160 * 1024);
    if (e == hipSuccess && h->dev.seed_chunks) {
        e = hipMalloc(&h->d_seed_heavy, seed_heavy_bytes());
        if (e == hipSuccess) e = hipMemset(h->d_seed_heavy, 0, seed_heavy_bytes());
    }
    if (e != hipSuccess) { release(h); return hip_fail(e, "handle setup"); }
    *out = h;
    return WEPP_OK;
}
}  // namespace

extern "C" int wepp_mat_create(const wepp_tree_desc* tree, int device, wepp_mat_t** out) {
    if (!tree || !out) return set_error(WEPP_EINVAL, "null argument");
    *out = nullptr;
    FlatMAT f;
    std::string err;
    try {
        int rc = flatten_tree(*tree, f, err);
        if (rc != WEPP_OK) return set_error(rc, err);
    } catch (const std::bad_alloc&) {
        return set_error(WEPP_ENOMEM, "out of host memory while flattening the tree");
    }
    return upload_flat(f, device, out);
}

// One flatten per host, one upload per device: the multi-GPU host loop flattens once (wepp_flat_create) and every
// device thread uploads the same image.
extern "C" int wepp_mat_upload(const wepp_flat_t* flat, int device, wepp_mat_t** out) {
    if (!flat || !out) return set_error(WEPP_EINVAL, "null argument");
    if (flat->f.streams.empty()) return set_error(WEPP_EINVAL, "the flat image holds no sweep streams");
    return upload_flat(flat->f, device, out);
}

extern "C" int wepp_mat_destroy(wepp_mat_t* mat) {
    release(mat);
    return WEPP_OK;
}

extern "C" int wepp_mat_get_stats(const wepp_mat_t* mat, wepp_mat_stats* out) {
    if (!mat || !out) return set_error(WEPP_EINVAL, "null argument");
    *out = mat->stats;
    return WEPP_OK;
}

extern "C" int wepp_mat_bfs_order(const wepp_mat_t* mat, uint32_t* bfs_ids) {
    if (!mat || !bfs_ids) return set_error(WEPP_EINVAL, "null argument");
    std::memcpy(bfs_ids, mat->bfs2id.data(), mat->bfs2id.size() * sizeof(uint32_t));
    return WEPP_OK;
}

extern "C" int wepp_mat_set_tile_reads(wepp_mat_t* mat, uint32_t reads_per_tile) {
    if (!mat) return set_error(WEPP_EINVAL, "null argument");
    if (reads_per_tile < 1 || reads_per_tile > 64) return set_error(WEPP_EINVAL, "reads_per_tile must be 1..64");
    mat->tile_reads = reads_per_tile;
    return WEPP_OK;
}

extern "C" int wepp_mat_set_use_crowns(wepp_mat_t* mat, int enable) {
    if (!mat) return set_error(WEPP_EINVAL, "null argument");
    mat->use_crowns = enable ? 1 : 0;
    return WEPP_OK;
}

extern "C" int wepp_mat_set_use_seeds(wepp_mat_t* mat, int enable) {
    if (!mat) return set_error(WEPP_EINVAL, "null argument");
    mat->use_seeds = enable ? 1 : 0;
    return WEPP_OK;
}

extern "C" int wepp_mat_set_pipeline(wepp_mat_t* mat, uint32_t sub_batches) {
    if (!mat) return set_error(WEPP_EINVAL, "null argument");
    if (sub_batches > wepp_mat::kPipeMax) return set_error(WEPP_EINVAL, "at most 8 sub-batches");
    mat->pipe_sub_batches = sub_batches;
    return WEPP_OK;
}

extern "C" int wepp_mat_set_use_walk(wepp_mat_t* mat, int enable) {
    if (!mat) return set_error(WEPP_EINVAL, "null argument");
    if (enable && !mat->walk_ok)
        return set_error(WEPP_ELIMIT, "per-read walks need streams of fewer than 2^25 nodes; this tree is placed by sweeps");
    mat->use_walk = enable ? 1 : 0;
    return WEPP_OK;
}

namespace {
// The placement of one batch whose reads are on the device.  plan_base / plan_total: the batch is reads
// [plan_base, plan_base + n_reads) of a call of plan_total reads (a sub-batch of wepp_place_batch's pipeline; a
// direct wepp_place_batch_device call is the whole: base 0, total n_reads) -- where its plan ids go in d_plan_of.
// the plan id and walk-slice id of every read of a call (wepp_mat_last_plans / _crowns read them back): grow-only
int ensure_plan_buffers(wepp_mat_t* mat, uint32_t plan_total) {
    if (plan_total <= mat->plan_of_bytes) return WEPP_OK;
    if (mat->d_plan_of) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(mat->d_plan_of); mat->d_plan_of = nullptr; mat->plan_of_bytes = 0; }
    if (mat->d_wsid_of) { (void)hipFree(mat->d_wsid_of); mat->d_wsid_of = nullptr; }
    const size_t need = (size_t)plan_total + plan_total / 4 + 256;
    hipError_t e = hipMalloc((void**)&mat->d_plan_of, need);
    if (e == hipSuccess) e = hipMalloc((void**)&mat->d_wsid_of, need * 4);
    if (e != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc plan ids: ") + hipGetErrorString(e));
    mat->plan_of_bytes = need;
    return WEPP_OK;
}

// Two sub-batches of one wepp_place_batch may be inside this function at once, on two host threads and two lanes (each
// lane has its own workspace, counters, side streams and events): what they share on the handle is read-only here but
// for the event ring (a slot claimed atomically), the job-size hint (atomic) and the call statistics (stat_mu).
int place_device(wepp_mat_t* mat, const uint32_t* d_read_off, const uint32_t* d_read_word, uint32_t n_reads,
                 uint32_t* d_best_bfs_j, int32_t* d_score, uint32_t* d_num_best, uint32_t* d_flags, hipStream_t stream,
                 uint32_t plan_base, uint32_t plan_total, uint32_t lane_idx = 0) {
    PlaceLane& L = mat->lane[lane_idx];
    {
        // (a pipelined wepp_place_batch has sized them for the whole call before its sub-batches start, on one thread)
        const int prc = ensure_plan_buffers(mat, plan_total);
        if (prc != WEPP_OK) return prc;
    }
    const uint32_t T = mat->tile_reads;
    const uint32_t ns = mat->dev.n_streams;

    // ---- workspace: [tier_of R bytes][list R][root_score R][partials: sum over tiers of nchunks*count*12] ----
    // The partials are sized for the worst case once the per-tier counts are known.
    bool ws_moved = false;
    auto grow = [&](size_t need) -> int {
        if (need <= L.ws_bytes) return WEPP_OK;
        // (the plain walks of this call may already run on their side stream, out of the lists in this workspace)
        if (L.ws) {
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipStreamSynchronize(L.side[MAX_STREAMS - 1]));
            for (uint32_t i : {MAX_STREAMS - 6, MAX_STREAMS - 7, MAX_STREAMS - 3, MAX_STREAMS - 2}) HIP_TRY(hipStreamSynchronize(L.side[i]));
            (void)hipFree(L.ws);
            L.ws = nullptr;
            L.ws_bytes = 0;
        }
        // (half as much again: the partials of a batch of long reads vary by a third with the tile size its longest
        // read allows, and a regrowth costs a hipFree + hipMalloc of ~100 MB -- 16 ms -- plus a second routing pass)
        need += need / 2;
        hipError_t e = hipMalloc(&L.ws, need);
        if (e != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc workspace: ") + hipGetErrorString(e));
        L.ws_bytes = need;
        ws_moved = true;
        return WEPP_OK;
    };
    const size_t tier_bytes = 0;       // (the plan ids live in mat->d_plan_of)
    const size_t list_bytes = (((size_t)n_reads * 4) + 255) & ~(size_t)255;
    const PlaceTunables& tun = mat->tun;
    const bool sort_reads = tun.sort_reads;   // (WEPP_SORT_READS=0, an A/B aid: keep the caller's read order on the whole-tree stream too)
    size_t sort_temp = 0;
    if (sort_reads) HIP_TRY(sort_reads_temp_bytes(n_reads, &sort_temp));
    sort_temp = (sort_temp + 255) & ~(size_t)255;
    // fixed part of the workspace: list | root_score | slot in block | jobs | sort keys in/out | sorted lists |
    // sort temp of the whole-tree plan and of the four walk classes (their sorts run on different side streams)
    constexpr uint32_t N_SORTS = 5;
    // (+ the chunked walk classes sized blind: first job of a read, and per class its reads, its job table, three partials per job)
    constexpr size_t blind_bytes = 2 * ((size_t)BLIND_CHUNKED_READS * 4 + (size_t)BLIND_JOB_CAP * 16);
    const size_t fixed_bytes = tier_bytes + 8 * list_bytes + blind_bytes + (sort_reads ? 3 * list_bytes + N_SORTS * sort_temp : 0);
    {
        // before routing only the fixed regions are needed; reserve a typical partial size too
        int rc = grow(fixed_bytes + (size_t)n_reads * 12 * 2);
        if (rc != WEPP_OK) return rc;
    }
    uint8_t* tier_of = nullptr;
    uint32_t *list = nullptr, *slot_in_blk = nullptr, *key_in = nullptr, *key_out = nullptr, *val_in = nullptr;
    uint32_t* job_n = nullptr;       // jobs of every read whose walk is cut into chunks (k_route)
    uint32_t* wlist[2] = {nullptr, nullptr};   // the plain walk classes' reads that have events in their stream (k_route appends them)
    uint32_t* wwlist = nullptr;                // reads with many events, a wave each (wave_kernels.hip)
    uint32_t *b_first = nullptr, *b_clist[2] = {nullptr, nullptr}, *b_jobs[2] = {nullptr, nullptr};   // the chunked classes sized blind (k_route's tables)
    int32_t* b_ps[2] = {nullptr, nullptr};
    uint32_t *b_pr[2] = {nullptr, nullptr}, *b_pc[2] = {nullptr, nullptr};
    uint32_t* wsid = nullptr;        // the window crown (index into DevMAT::wc_info) of every read routed to slot WC_SLOT
    int32_t* root_score = nullptr;
    void* sort_tmp = nullptr;
    auto carve = [&]() {
        char* p = (char*)L.ws;
        tier_of = mat->d_plan_of + plan_base;
        list = (uint32_t*)p; p += list_bytes;
        root_score = (int32_t*)p; p += list_bytes;
        slot_in_blk = (uint32_t*)p; p += list_bytes;
        job_n = (uint32_t*)p; p += list_bytes;
        wlist[0] = (uint32_t*)p; p += list_bytes;
        wlist[1] = (uint32_t*)p; p += list_bytes;
        b_first = (uint32_t*)p; p += list_bytes;
        wwlist = (uint32_t*)p; p += list_bytes;
        for (uint32_t cc = 0; cc < 2; cc++) {
            b_clist[cc] = (uint32_t*)p; p += (size_t)BLIND_CHUNKED_READS * 4;
            b_jobs[cc] = (uint32_t*)p; p += (size_t)BLIND_JOB_CAP * 4;
            b_ps[cc] = (int32_t*)p; p += (size_t)BLIND_JOB_CAP * 4;
            b_pr[cc] = (uint32_t*)p; p += (size_t)BLIND_JOB_CAP * 4;
            b_pc[cc] = (uint32_t*)p; p += (size_t)BLIND_JOB_CAP * 4;
        }
        wsid = mat->d_wsid_of + plan_base;
        if (sort_reads) {
            key_in = (uint32_t*)p; p += list_bytes;
            key_out = (uint32_t*)p; p += list_bytes;
            val_in = (uint32_t*)p; p += list_bytes;   // the sorted lists (same offsets as in `list`)
            sort_tmp = p;                             // N_SORTS regions of sort_temp bytes
        }
    };
    carve();
    uint32_t* tier_info = L.d_info + L.info_idx * TI_WORDS;
    uint32_t* blk_counts = L.d_info + 2 * TI_WORDS;

    // events per job of the two chunked classes: WEPP_WALK_JOB_EVENTS fixes them, else they follow the handle's
    // traffic (device_mat.hpp: WALK_TARGET_JOBS)
    const uint32_t job_events = tun.job_events ? (tun.job_events | (tun.job_events << 16)) : (mat->job_events[0] | (mat->job_events[1] << 16));
    const uint32_t walk_max_events = tun.walk_max_events, stack8 = tun.stack8, stack16 = tun.stack16;
    // whole-genome samples are seeded (seed_kernels.hip) when work skipping is on and the tree carries signatures
    const uint32_t seed_min_hard = (mat->use_seeds && mat->use_crowns && mat->dev.seed_chunks) ? tun.seed_min_hard : 0xFFFFFFFFu;
    // ---- route the reads to streams ------------------------------------------------------
    const uint32_t ev_slot = (uint32_t)(mat->n_timed.fetch_add(1, std::memory_order_relaxed) % wepp_mat::kRing);
    constexpr uint32_t BLIND16_STREAM = MAX_STREAMS - 6;     // (side stream of the second plain walk class)
    constexpr uint32_t PLAN_STREAM = MAX_STREAMS - 7;
    // Streams of a call.  The caller's stream runs k_route and, right behind it (no cross-stream wait: ~25 us), the
    // chunked walks of the first class -- the longest chain of the common case; the plain walks and the second chunked
    // class start from the routing kernel's event on side streams; everything the HOST has to size -- k_scatter, the
    // counters' copy, the sweeps, the planned walks -- lives on the plan stream `ps`, off the walks' path.  All of them
    // join the caller's stream at the end.
    const bool walking = mat->use_walk && walk_max_events;
    hipStream_t ps = walking ? L.side[PLAN_STREAM] : stream;
    // one walk class out of k_route's tables: the plain classes from their read lists, the chunked ones from their job tables
    auto blind_class = [&](uint32_t cls, hipStream_t q, bool fork_q) -> hipError_t {
        hipError_t e = hipSuccess;
        if (fork_q) e = hipStreamWaitEvent(q, L.route_ev, 0);
        if (e != hipSuccess) return e;
        if (cls == PLAN_WALK8 || cls == PLAN_WALK16) {
            e = launch_walk_blind(mat->dev, cls, cls == PLAN_WALK8 ? stack8 : stack16, n_reads, wlist[cls], tier_info + TI_WCUR + cls, d_read_off, d_read_word,
                                  root_score, d_best_bfs_j, d_score, d_num_best, d_flags, mat->d_work, wsid, q);
        } else {
            const uint32_t cc = cls - PLAN_WALKC8;
            WalkJobs jb{};
            jb.job_first = b_first;
            jb.skip = tier_info + TI_JOVER + cc;
            jb.job_n = job_n;
            jb.part_score = b_ps[cc];
            jb.part_rank = b_pr[cc];
            jb.part_cnt = b_pc[cc];
            e = launch_walk_jobs_blind(mat->dev, cls, cc ? stack16 : stack8, jb, b_jobs[cc], tier_info + TI_JCUR + cc, b_clist[cc], tier_info + TI_CCUR + cc,
                                       d_read_off, d_read_word, root_score, d_best_bfs_j, d_score, d_num_best, d_flags, mat->d_work, wsid, q);
        }
        return e;
    };
    const bool ww_jobs = mat->ww_by_jobs.load(std::memory_order_relaxed) != 0;     // (the reads with 17 - 256 events: a wave each, or jobs -- by the handle's previous call)
    // (... and then reads of up to 16 events walk plainly, as before the waves: 100 000 reads with 7 - 16 events are
    // nothing to a walk launch -- its time is its longest walk -- and 0.3 ms as jobs)
    const uint32_t walk_limit = (ww_jobs && !tun.ww_fixed) ? std::max(walk_max_events, WALK_MAX_EVENTS_BY_JOBS) : walk_max_events;
    auto route = [&]() -> int {
        // the counters alternate between two sets: this call's set is zero (cleared at creation or by the
        // previous k_route), and this k_route clears the other one for the next call
        tier_info = L.d_info + L.info_idx * TI_WORDS;
        uint32_t* tier_info_next = L.d_info + (L.info_idx ^ 1u) * TI_WORDS;
        // k_route places the reads that have no event in their stream itself and lists the other plain walkers; their
        // walks are launched from those lists at once, sized for the worst case, on a side stream -- the routing
        // counters' trip to the host, the planning of the rarer classes and k_scatter are off their path
        RouteDirect direct{};
        if (walking) direct = RouteDirect{{wlist[0], wlist[1]}, {b_clist[0], b_clist[1]}, {b_jobs[0], b_jobs[1]}, b_first, wwlist,
                                          d_best_bfs_j, d_score, d_num_best, d_flags, mat->d_work, walk_max_events,
                                          tun.ww_fixed ? tun.ww_block_max_small : ww_jobs ? 0u : 0xFFFFFFFFu, tun.ww_fixed ? tun.ww_block_max_big : ww_jobs ? 0u : 0xFFFFFFFFu};
        HIP_TRY(launch_route(mat->dev, d_read_off, d_read_word, n_reads, mat->use_crowns, walking ? walk_limit : 0u, job_events, stack8, stack16, seed_min_hard, tun.seed_min_nodes, job_n, tier_of, root_score, blk_counts,
                             tier_info, slot_in_blk, tier_info_next, wsid, direct, stream));
        L.info_idx ^= 1u;
        HIP_TRY(hipEventRecord(mat->ev0[ev_slot], stream));        // (the timed span of a call: everything behind the routing kernel)
        if (walking) {
            HIP_TRY(hipEventRecord(L.route_ev, stream));
            // Launched BLIND, sized for the worst case, before the routing counters are back: the two classes of reads with
            // at most WALK8_K entries -- nearly every read of a sequencing run.  The plain walks go on the caller's stream, right
            // behind k_route (no cross-stream wait: ~25 us); on a side stream the reads with many events, a wave each, and
            // the chunked class (reads with even more events, cut into jobs: k_route has entered them into a job table; walk +
            // combination leave at once when the class outgrew the table, TI_JOVER: the host's planned launch takes it).  The classes of 9 - 16 entries are launched from the
            // same tables once the counters say they hold reads (blind_class below): two launches sized for a million
            // reads that find a handful cost ~20 us of workgroup dispatch each, and eight API calls per device call.
            HIP_TRY(blind_class(PLAN_WALK8, stream, false));
            {   // the reads with many events, a wave per 64 of them (wave_kernels.hip), then what is left of the chunked class
                hipStream_t q = L.side[MAX_STREAMS - 1];
                HIP_TRY(hipStreamWaitEvent(q, L.route_ev, 0));
                HIP_TRY(launch_walk_wave(mat->dev, wwlist, tier_info + TI_WWCUR, n_reads, d_read_off, d_read_word, root_score, d_best_bfs_j, d_score, d_num_best,
                                         d_flags, mat->d_work, wsid, q));
                HIP_TRY(hipEventRecord(L.join_ev[MAX_STREAMS - 1], q));
                // (the job class on a stream of its own: nearly always two launches that find nothing, ~10 us that
                // used to sit behind the wave kernel on the call's longest chain)
                hipStream_t q2 = L.side[MAX_STREAMS - 2];
                HIP_TRY(hipStreamWaitEvent(q2, L.route_ev, 0));
                HIP_TRY(blind_class(PLAN_WALKC8, q2, false));
                HIP_TRY(hipEventRecord(L.join_ev[MAX_STREAMS - 2], q2));
            }
            if (tun.blind16) {
                // ... and the plain walks of 9 - 16 entries on a stream of their own: launched once the counters have
                // reached the host they started when the walks of 1 - 8 entries ended, ~15 us of a 170 us step
                HIP_TRY(blind_class(PLAN_WALK16, L.side[BLIND16_STREAM], true));
                HIP_TRY(hipEventRecord(L.join_ev[BLIND16_STREAM], L.side[BLIND16_STREAM]));
            }
            HIP_TRY(hipStreamWaitEvent(ps, L.route_ev, 0));
        }
        HIP_TRY(launch_scatter(tier_of, slot_in_blk, n_reads, blk_counts, tier_info, list, walking, ps));
        return WEPP_OK;
    };

    {
        int rc = route();
        if (rc != WEPP_OK) return rc;
    }
    HIP_TRY(hipMemcpyAsync(L.h_info, tier_info, TI_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, ps));
    // the host sizes the launches from the counters, and the GPU idles until it has: poll for them (a
    // blocking wait adds its wake-up, ~15 us per call, to that idle time) ...
    {
        // ... spinning for the first 200 us (the routing kernel of a million reads takes ~45, behind the reads' H2D), then yielding the core
        // between queries -- eight ranks spinning on the cores of one container starve their own staging workers --,
        // and after ~2 ms (a long queue in front of this call) waiting blocking
        hipError_t q = hipErrorNotReady;
        const auto t_poll = std::chrono::steady_clock::now();
        for (;;) {
            q = hipStreamQuery(ps);
            if (q != hipErrorNotReady) break;
            const auto waited = std::chrono::steady_clock::now() - t_poll;
            if (waited > std::chrono::milliseconds(2)) break;
            if (waited > std::chrono::microseconds(200)) std::this_thread::yield();
        }
        (void)hipGetLastError();   // "not ready" is not an error: keep it out of the launchers' hipGetLastError()
        if (q != hipSuccess) HIP_TRY(hipStreamSynchronize(ps));
    }
    const uint32_t* info = L.h_info;
    for (uint32_t cc = 0; cc < 2; cc++) {
        const uint64_t ev = (uint64_t)info[TI_EVENTS + cc] << 6;
        const uint32_t je = (uint32_t)std::min<uint64_t>(WALK_JOB_EVENTS_MAX, std::max<uint64_t>(WALK_JOB_EVENTS, ev / WALK_TARGET_JOBS));
        mat->job_events[cc] = je <= WALK_JOB_EVENTS ? je : ((je + 15u) & ~15u);       // (what the NEXT call's k_route cuts this class's walks into)
    }

    // the reads with 17 - 256 events of the NEXT call: a wave per 64 events each while they are few (scaled to the call's size)
    {
        const uint64_t scale = std::max<uint64_t>(1, ((uint64_t)n_reads + (1u << 20) - 1) >> 20);
        mat->ww_by_jobs.store((info[TI_WWCAND] > WW_CALL_MAX_SMALL * scale || info[TI_WWCAND + 1] > WW_CALL_MAX_BIG * scale) ? 1u : 0u, std::memory_order_relaxed);
    }
    const bool blind_walks = mat->use_walk && walk_max_events;       // (launched behind k_route, joined below)
    bool late16[2] = {false, false};
    if (blind_walks) {
        // the classes of 9 - 16 entries, from k_route's tables like the others, when they hold reads
        if (tun.blind16) late16[0] = true;
        else if (info[TI_WCUR + 1]) { HIP_TRY(blind_class(PLAN_WALK16, L.side[BLIND16_STREAM], true)); HIP_TRY(hipEventRecord(L.join_ev[BLIND16_STREAM], L.side[BLIND16_STREAM])); late16[0] = true; }
        if (info[TI_CCUR + 1] && !info[TI_JOVER + 1]) { HIP_TRY(blind_class(PLAN_WALKC16, L.side[MAX_STREAMS - 3], true)); HIP_TRY(hipEventRecord(L.join_ev[MAX_STREAMS - 3], L.side[MAX_STREAMS - 3])); late16[1] = true; }
    }
    // ---- plan the launches ---------------------------------------------------------
    struct Plan { uint32_t t, count, off, T, ntiles, nchunks, bpc, ent_cap, key_cap, lds_bytes; bool s_in_lds, dense, window, win_table; size_t part_off; const uint32_t* lst; const DevStream* st; uint64_t sbytes; };
    Plan plans[MAX_PLANS];
    uint32_t np = 0;
    size_t part_total = 0;
    const uint32_t bm_bytes = mat->dev.bm_words * 4;
    if (bm_bytes > 128 * 1024) return set_error(WEPP_ELIMIT, "position bitmap does not fit in LDS");
    // walk plans (device_mat.hpp): the reads of one (class, stream) that walk their own events
    WalkPlans walkc[2]{};
    uint32_t walkc_off[2] = {0, 0};   // list offsets of the chunked walk plans
    uint64_t walk_reads = (uint64_t)info[TI_WCUR] + info[TI_WCUR + 1] + info[TI_RESOLVED] + info[TI_W16WAVE], n_jobs[2] = {0, 0};     // (the plain walk classes: placed by k_route itself or by the blind walks behind it)
    uint32_t walkc_reads[2] = {0, 0};
    uint32_t arena_n = 0, arena_off = 0, arena_maxk = 1;
    uint32_t seed_n = 0, seed_off = 0, seed_maxk = 1;
    size_t arena_part = 0;
    // the window plans share ONE launch (k_sweep_windows): their chunks are sized for the tiles of all of them together
    // (sized per plan, each window asked for enough waves to fill the chip on its own: 1.2 kb reads, 30 windows -- 9
    // workgroups per tile, each rebuilding the tile's table to sweep two blocks of the window's stream)
    const bool fuse_windows = !tun.sweep_unfused && !tun.windows_unfused && mat->d_wstreams;
    auto tile_reads = [&](uint32_t maxk) { uint32_t Tp = T; while (Tp > 1 && (uint64_t)Tp * maxk > MAX_TILE_ENTRIES) Tp >>= 1; return Tp; };
    auto takes_table = [&](uint32_t maxk) { return maxk <= MAX_TILE_ENTRIES && maxk >= DENSE_MIN_READ_WORDS && mat->dev.max_pos <= DENSE_MAX_POS; };
    uint64_t win_tiles = 0;
    if (fuse_windows)
        for (uint32_t id = 0; id < MAX_PLANS; id++) {
            const uint32_t count = info[TI_COUNT + id], maxk = std::max<uint32_t>(1, info[TI_MAXK + id]);
            if (count && plan_class(id) == PLAN_WIN && takes_table(maxk)) win_tiles += (count + tile_reads(maxk) - 1) / tile_reads(maxk);
        }
    for (uint32_t id = 0; id < MAX_PLANS; id++) {
        const uint32_t count = info[TI_COUNT + id];
        if (!count) continue;
        const uint32_t t = plan_index(id), cls = plan_class(id);
        if (cls == PLAN_SEED) {
            // whole-genome samples: a workgroup each, final results written by the kernel (seed_kernels.hip)
            if (!mat->dev.seed_chunks) return set_error(WEPP_EDEVICE, "routing produced an invalid plan id");
            seed_n = count;
            seed_off = info[TI_OFF + id];
            seed_maxk = std::max<uint32_t>(1, info[TI_MAXK + id]);
            continue;
        }
        {
            // (slot WC_SLOT of a walk class = the window crowns: one plan, a crown per read)
            if (cls > PLAN_WIN || (cls == PLAN_WIN ? t >= mat->wstreams.size() : (t >= ns && !(t == WC_SLOT && mat->dev.wc_windows))))
                return set_error(WEPP_EDEVICE, "routing produced an invalid plan id");
        }
        if (cls == PLAN_WALKC8 || cls == PLAN_WALKC16) {
            // the reads with many events: their walks are cut into jobs (below); the plans of a chunked class
            // are the streams, the jobs of stream t numbered behind those of the streams before it
            const uint32_t cc = cls - PLAN_WALKC8;
            if (blind_walks && !info[TI_JOVER + cc]) { walk_reads += count; continue; }     // (launched blind behind k_route)
            WalkPlans& wc = walkc[cc];
            WalkPlanDev& d = wc.p[wc.n];
            d.tier = t;
            d.n_list = info[TI_JOBS + cc * MAX_STREAMS + t];
            d.job0 = (uint32_t)n_jobs[cc];
            walkc_off[cc] = info[TI_OFF + plan_id(cls, 0)];
            d.list = nullptr;
            d.wave_end = (wc.n ? wc.p[wc.n - 1].wave_end : 0u) + walk_plan_waves(d.n_list);
            wc.n++;
            n_jobs[cc] += d.n_list;
            walkc_reads[cc] += count;
            walk_reads += count;
            continue;
        }
        if (cls == PLAN_WALK8 || cls == PLAN_WALK16) continue;     // (never counted: k_route places or lists them itself, see walk_reads below)
        if (cls == PLAN_SWEEP && t == WC_SLOT) {
            // the reads that sweep their window crown, one wave each (k_sweep_arena); one partial per read
            arena_n = count;
            arena_off = info[TI_OFF + id];
            arena_maxk = std::max<uint32_t>(1, info[TI_MAXK + id]);
            arena_part = part_total;
            part_total += (size_t)count * 12 * ARENA_CHUNKS;
            if ((uint64_t)count * ARENA_CHUNKS >= (1ull << 31)) return set_error(WEPP_ELIMIT, "too many window-crown sweeps in one call; split the batch");
            continue;
        }
        Plan& p = plans[np++];
        p.window = cls == PLAN_WIN;
        p.t = p.window ? ns - 1 : t;                 // (a window stream is the whole tree for its reads)
        p.st = p.window ? &mat->wstreams[t] : &mat->streams[t];
        p.sbytes = p.window ? mat->wstream_bytes[t] : mat->stream_bytes[t];
        p.count = count;
        p.off = info[TI_OFF + id];
        // reads per tile: at most MAX_TILE_ENTRIES read words per tile (long reads share a
        // sweep between fewer reads), a power of two, never more than the knob
        const uint32_t maxk = std::max<uint32_t>(1, info[TI_MAXK + id]);
        uint32_t Tp = T;
        while (Tp > 1 && (uint64_t)Tp * maxk > MAX_TILE_ENTRIES) Tp >>= 1;
        p.T = Tp;
        p.ntiles = (count + Tp - 1) / Tp;
        // a read longer than MAX_TILE_ENTRIES words is swept alone with its words left in global memory
        p.s_in_lds = maxk <= MAX_TILE_ENTRIES;
        p.dense = p.s_in_lds && maxk >= DENSE_MIN_READ_WORDS && mat->dev.max_pos <= DENSE_MAX_POS;
        const uint64_t need = std::min<uint64_t>((uint64_t)std::min(Tp, count) * maxk, MAX_TILE_ENTRIES);
        const uint32_t cap = (uint32_t)((need + 63) & ~63ull);
        uint32_t kcap = 64;
        while (kcap < need) kcap <<= 1;                        // the bitonic network wants a power of two
        p.ent_cap = p.s_in_lds ? cap : 0;
        p.key_cap = p.dense ? kcap : 0;
        // long reads inside a window: per-position read masks instead of sorted keys (the argument then
        // carries the window's first position)
        p.win_table = p.window && p.dense;
        if (p.win_table) p.key_cap = t * WIN_STRIDE;
        p.lds_bytes = p.s_in_lds ? sweep_lds_bytes(mat->dev.bm_words, cap, kcap, p.dense, p.win_table) : bm_bytes;
        // chunks: enough single-wave workgroups to fill 256 CUs, cut at checkpoints
        const DevStream& st = *p.st;
        const uint32_t target_waves = tun.target_waves;   // 4096: 16 resident single-wave workgroups per CU x 256 CUs; 2048 / 8192 / 16384 measured slower (WEPP_TARGET_WAVES: tuning aid)
        // (the 8-wave workgroups of the dense / window variant run for milliseconds: four times as many of them
        // balance the chip better -- 1.2 kb reads 132 -> 120 ms per 200 K at 16384, the same at 32768)
        const uint32_t target_waves_dense = tun.target_waves_dense;
        uint32_t nchunks = std::max<uint32_t>(1, ((p.dense ? target_waves_dense : target_waves) + p.ntiles - 1) / p.ntiles);
        if (p.win_table && fuse_windows) nchunks = std::max<uint32_t>(1, (uint32_t)((target_waves_dense + win_tiles - 1) / win_tiles));
        // ... and chunks no longer than what stays in an XCD's L2 while the tiles sweep it: the waves of
        // a launch are ordered chunk-major (all tiles of chunk 0, then of chunk 1, ...), so the ~4 K
        // resident waves walk the same ~1.5 MB of the stream together instead of drifting apart over
        // 118 MB (whole-tree sweep of 1 M reads: 432 ms with one chunk per tile, 235 ms with 80)
        const uint64_t chunk_bytes = tun.chunk_bytes;
        nchunks = std::max<uint32_t>(nchunks, (uint32_t)((p.sbytes + chunk_bytes - 1) / chunk_bytes));
        nchunks = std::min<uint32_t>(nchunks, (uint32_t)std::max<uint64_t>(1, SWEEP_MAX_PARTIAL_BYTES / ((uint64_t)count * 12)));
        // ... but a chunk's wave pays a set-up (bitmap, read words, checkpoint) worth several blocks: no chunk
        // shorter than 8 blocks, however few tiles the plan has (a plan of a few hundred reads used to be cut into
        // 4096 waves of one or two blocks each)
        nchunks = std::min<uint32_t>(nchunks, std::max<uint32_t>(1, st.NB / 8));
        if (p.dense) nchunks = std::max<uint32_t>(nchunks, DENSE_WAVES_PER_WG);   // one chunk per wave of the workgroup
        nchunks = std::min(nchunks, st.ncp);
        const uint32_t cps_per_chunk = (st.ncp + nchunks - 1) / nchunks;
        p.bpc = cps_per_chunk * st.cp_stride;
        p.nchunks = (st.NB + p.bpc - 1) / p.bpc;
        if (p.dense) p.nchunks = (p.nchunks + DENSE_WAVES_PER_WG - 1) / DENSE_WAVES_PER_WG * DENSE_WAVES_PER_WG;
        p.part_off = part_total;
        part_total += (size_t)p.nchunks * count * 12;
    }
    {
        ws_moved = false;
        int rc = grow(fixed_bytes + part_total);
        if (rc != WEPP_OK) return rc;
        if (ws_moved) {
            // the workspace moved: redo the (cheap) routing into the new buffer
            carve();
            rc = route();
            if (rc != WEPP_OK) return rc;
        }
    }
    char* part_base = (char*)L.ws + fixed_bytes;
    // (the workspace may have moved above: every pointer into it is taken from here on)
    for (uint32_t cc = 0; cc < 2; cc++)
        for (uint32_t k = 0; k < walkc[cc].n; k++) walkc[cc].p[k].list = list + walkc_off[cc];
    const bool debug_plans = tun.debug_plans;
    for (uint32_t i = 0; i < np; i++) {
        Plan& p = plans[i];
        p.lst = list + p.off;
        if (debug_plans)
            fprintf(stderr, "[plan] %s t=%u count=%u off=%u T=%u ntiles=%u nchunks=%u bpc=%u NB=%u ncp=%u dense=%d lds=%u ent_cap=%u key_cap=%u part_off=%zu\n",
                    p.window ? "window" : "sweep", p.t, p.count, p.off, p.T, p.ntiles, p.nchunks, p.bpc, p.st->NB, p.st->ncp, (int)p.dense,
                    p.lds_bytes, p.ent_cap, p.key_cap, p.part_off);
        // the reads that sweep the whole tree go by first listed position (sort_reads.hip)
        if (sort_reads && !p.window && p.t + 1 == ns && p.count >= SORT_MIN_READS) {
            HIP_TRY(launch_first_pos(list + p.off, p.count, d_read_off, d_read_word, key_in + p.off, ps));
            HIP_TRY(launch_sort_reads(key_in + p.off, key_out + p.off, list + p.off, val_in + p.off, p.count, sort_tmp, sort_temp, ps));
            p.lst = val_in + p.off;
        }
    }

    // ---- sweeps (timed as a group) + finalizes ------------------------------------------
    // The plain (short-read) plans are fused into ONE launch, longest chunks first; dense
    // and out-of-LDS plans get their own launch on a side stream forked from / joined into
    // `stream`.  Every plan's finalize follows its sweep.
    const uint32_t slot = ev_slot;
    uint64_t passes = 0, bytes = 0;
    auto parts = [&](const Plan& p, int32_t*& ps, uint32_t*& pr, uint32_t*& pc) {
        ps = (int32_t*)(part_base + p.part_off);
        pr = (uint32_t*)(ps + (size_t)p.nchunks * p.count);
        pc = pr + (size_t)p.nchunks * p.count;
    };
    uint32_t order[MAX_PLANS], n_plain = 0, n_other = 0, others[MAX_PLANS];
    // WEPP_SWEEP_UNFUSED=1 (profiling aid): one launch per plan, back to back on `stream`, so that a
    // kernel trace shows the time of every stream's sweep; results are identical
    const bool unfused = tun.sweep_unfused;
    uint32_t wins[MAX_WINDOWS], n_wins = 0;
    for (uint32_t i = 0; i < np; i++) {
        if (plans[i].s_in_lds && !plans[i].dense && !unfused && n_plain < MAX_STREAMS) order[n_plain++] = i;
        else if (plans[i].win_table && fuse_windows && n_wins < MAX_WINDOWS) wins[n_wins++] = i;
        else others[n_other++] = i;
        passes += plans[i].ntiles;                                   // every tile sweeps its stream once
        bytes += (uint64_t)plans[i].ntiles * plans[i].sbytes;
    }
    std::sort(order, order + n_plain, [&](uint32_t a, uint32_t b) { return plans[a].bpc > plans[b].bpc; });
    const bool walks = walkc[0].n || walkc[1].n;
    if (n_jobs[0] + n_jobs[1] >= (1ull << 31)) return set_error(WEPP_ELIMIT, "too many walk jobs in one call; split the batch");
    for (uint32_t cls = 0; cls < 2; cls++)
        passes += (info[TI_WCUR + cls] + 63) / 64;      // a walk "pass" = one wave of 64 reads
    passes += (info[TI_RESOLVED] + 63) / 64 + info[TI_WWCUR] + info[TI_WWCUR + 1];
    for (uint32_t cc = 0; cc < 2; cc++)
        if (blind_walks && !info[TI_JOVER + cc]) {
            passes += (info[TI_JCUR + cc] + 63) / 64;
            bytes += (uint64_t)info[TI_CCUR + cc] * 40 + (uint64_t)info[TI_JCUR + cc] * 16;      // (job table, partials, the combination's reads and results)
        }
    for (uint32_t cc = 0; cc < 2; cc++)
        if (walkc[cc].n) passes += walkc[cc].p[walkc[cc].n - 1].wave_end;
    const uint32_t n_walk_chains = (walkc[0].n || walkc[1].n) ? 1u : 0u;
    const uint32_t n_chains = n_other + n_walk_chains + (arena_n ? 1u : 0u) + (seed_n ? 1u : 0u) + (n_wins ? 1u : 0u);     // launch chains beside the fused plain sweeps
    const bool fork = !unfused && (n_chains > 0) && (n_plain > 0 || n_chains > 1);
    if (fork) HIP_TRY(hipEventRecord(L.fork_ev, ps));
    // the side streams join the caller's stream only after everything has been launched: a join in between
    // would make the launches behind it wait for the side stream's kernels
    uint32_t joins[2 * MAX_STREAMS], n_joins = 0;     // (a stream may be listed twice: waiting twice for its event is harmless)
    // side streams of the sweeps that are launched on their own (dense / window / out-of-LDS plans); the last
    // three belong to the walks
    constexpr uint32_t OTHER_SIDE_STREAMS = MAX_STREAMS - 3;
    // A walk class's reads go by (stream, first listed position) when there are enough of them (device_mat.hpp:
    // WALK_SORT_MIN_READS; WEPP_WALK_SORT=0: A/B aid): sorted on the class's own stream, right before its launch.
    // `region` = which of the sort temp regions (1..4; 0 is the whole-tree plan's).
    const bool walk_sort = tun.walk_sort;
    auto sort_class = [&](uint32_t cls, uint32_t region, hipStream_t q, bool& sorted) -> int {
        sorted = false;
        if (!walk_sort) return WEPP_OK;
        const uint32_t off = info[TI_OFF + plan_id(cls, 0)];
        const uint32_t cnt = info[TI_OFF + plan_id(cls, 0) + MAX_STREAMS] - off;      // (plan ids of a class are consecutive)
        if (cnt < WALK_SORT_MIN_READS) return WEPP_OK;
        HIP_TRY(launch_walk_keys(list + off, cnt, tier_of, d_read_off, d_read_word, key_in + off, q));
        HIP_TRY(launch_sort_reads(key_in + off, key_out + off, list + off, val_in + off, cnt, (char*)sort_tmp + region * sort_temp,
                                  sort_temp, q, WALK_SORT_KEY_BITS));
        sorted = true;
        return WEPP_OK;
    };
    if (walks) {
        // the walks write the final per-read results themselves; the plain ones, the chunked ones and the
        // sweeps run side by side (the walks wait on memory most of the time)
        hipStream_t q = ps;
        if (walkc[0].n || walkc[1].n) {
            // chunked walks, per class: jobs per read in list order -> exclusive scan -> the walk (a partial per
            // job; a job finds its read by bisection in the scanned offsets) -> one combination per read.
            // Buffers: a second grow-only workspace holding both classes.
            auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
            size_t scan_temp[2] = {0, 0}, need = 0, base[2] = {0, 0};
            for (uint32_t cc = 0; cc < 2; cc++) {
                if (!walkc[cc].n) continue;
                HIP_TRY(scan_u32_temp_bytes(walkc_reads[cc], &scan_temp[cc]));
                base[cc] = need;
                need += 2 * pad((size_t)walkc_reads[cc] * 4) + pad(scan_temp[cc]) + 3 * pad((size_t)n_jobs[cc] * 4);
            }
            if (need > L.ws2_bytes) {
                if (L.ws2) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(L.ws2); L.ws2 = nullptr; L.ws2_bytes = 0; }
                hipError_t e2 = hipMalloc(&L.ws2, need + need / 4);
                if (e2 != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc walk workspace: ") + hipGetErrorString(e2));
                L.ws2_bytes = need + need / 4;
            }
            for (uint32_t cc = 0; cc < 2; cc++) {
                if (!walkc[cc].n) continue;
                // each class on a side stream of its own: a chain of short, latency-bound launches
                if (fork) {
                    q = L.side[MAX_STREAMS - 2 - cc];
                    HIP_TRY(hipStreamWaitEvent(q, L.fork_ev, 0));
                }
                const uint32_t R3 = walkc_reads[cc], J = (uint32_t)n_jobs[cc];
                const size_t b_cnt = pad((size_t)R3 * 4), b_tmp = pad(scan_temp[cc]), b_job = pad((size_t)J * 4);
                char* w2 = (char*)L.ws2 + base[cc];
                uint32_t* jcnt = (uint32_t*)w2; w2 += b_cnt;
                uint32_t* joff = (uint32_t*)w2; w2 += b_cnt;
                void* jtmp = w2; w2 += b_tmp;
                WalkJobs jb{};
                jb.n_list = R3;
                jb.job_off = joff;
                jb.job_n = job_n;
                jb.part_score = (int32_t*)w2; w2 += b_job;
                jb.part_rank = (uint32_t*)w2; w2 += b_job;
                jb.part_cnt = (uint32_t*)w2;
                {
                    bool sorted = false;
                    int rc = sort_class(PLAN_WALKC8 + cc, 3 + cc, q, sorted);
                    if (rc != WEPP_OK) return rc;
                    if (sorted)
                        for (uint32_t k = 0; k < walkc[cc].n; k++) walkc[cc].p[k].list = val_in + walkc_off[cc];
                }
                const uint32_t* list3 = walkc[cc].p[0].list;
                // gather (list, count in, count out), scan (in, out), combination (list, offset, count, three partial
                // arrays, four results): what the chain's small kernels move
                bytes += (uint64_t)R3 * (12 + 8 + 12 + 16) + (uint64_t)J * 12;
                HIP_TRY(launch_gather_jobs(list3, R3, job_n, jcnt, q));
                HIP_TRY(launch_exclusive_scan_u32(jcnt, joff, R3, jtmp, scan_temp[cc], q));
                HIP_TRY(launch_walk_jobs(mat->dev, walkc[cc], PLAN_WALKC8 + cc, info[TI_OPEN + 2 + cc], jb, d_read_off, d_read_word, root_score, mat->d_work, wsid, q));
                HIP_TRY(launch_finalize_jobs(mat->dev, list3, R3, jb, d_read_off, d_read_word, d_best_bfs_j, d_score,
                                             d_num_best, d_flags, q));
                if (fork) {
                    HIP_TRY(hipEventRecord(L.join_ev[MAX_STREAMS - 2 - cc], q));
                    joins[n_joins++] = MAX_STREAMS - 2 - cc;
                }
            }
        }
    }
    for (uint32_t k = 0; k < n_other; k++) {
        const Plan& p = plans[others[k]];
        hipStream_t q = fork ? L.side[k % OTHER_SIDE_STREAMS] : ps;
        if (fork) HIP_TRY(hipStreamWaitEvent(q, L.fork_ev, 0));
        int32_t* ps; uint32_t *pr, *pc;
        parts(p, ps, pr, pc);
        HIP_TRY(launch_sweep(mat->dev, *p.st, d_read_off, d_read_word, root_score, p.lst, p.count, p.T,
                             p.ntiles,
                             p.nchunks, p.bpc, p.s_in_lds, p.dense, p.win_table, p.ent_cap, p.key_cap, p.lds_bytes, ps, pr, pc, q));
        HIP_TRY(launch_finalize(mat->dev, d_read_off, d_read_word, p.lst, p.count, p.nchunks, ps, pr, pc,
                                d_best_bfs_j, d_score, d_num_best, d_flags, q));
        if (fork && (k + OTHER_SIDE_STREAMS >= n_other)) {
            // the last launch on every side stream joins the caller's stream
            HIP_TRY(hipEventRecord(L.join_ev[k % OTHER_SIDE_STREAMS], q));
            joins[n_joins++] = k % OTHER_SIDE_STREAMS;
        }
    }
    if (n_wins) {
        // longest chunks first: the workgroups of the largest windows' streams start first
        std::sort(wins, wins + n_wins, [&](uint32_t a, uint32_t b) { return plans[a].bpc > plans[b].bpc; });
        WinPlans pl{};
        uint32_t wg = 0, fin = 0, lds_max = 0;
        for (uint32_t k = 0; k < n_wins; k++) {
            const Plan& p = plans[wins[k]];
            WinPlanDev& d = pl.p[k];
            d.sid = (uint32_t)(p.st - mat->wstreams.data());
            d.list = p.lst;
            d.n_list = p.count; d.T = p.T; d.ntiles = p.ntiles; d.bpc = p.bpc; d.ent_cap = p.ent_cap; d.win_base = p.key_cap; d.nchunks = p.nchunks;
            parts(p, d.part_score, d.part_rank, d.part_cnt);
            wg += p.ntiles * (p.nchunks / DENSE_WAVES_PER_WG);
            d.wg_end = wg;
            fin += finalize_blocks(p.count, p.nchunks);
            d.fin_end = fin;
            lds_max = std::max(lds_max, p.lds_bytes);
        }
        pl.n = n_wins;
        hipStream_t q = fork ? L.side[OTHER_SIDE_STREAMS - 3] : ps;
        if (fork) HIP_TRY(hipStreamWaitEvent(q, L.fork_ev, 0));
        HIP_TRY(launch_sweep_windows(mat->dev, mat->d_wstreams, pl, d_read_off, d_read_word, root_score, lds_max, q));
        HIP_TRY(launch_finalize_windows(mat->dev, pl, d_read_off, d_read_word, d_best_bfs_j, d_score, d_num_best, d_flags, q));
        if (fork) {
            HIP_TRY(hipEventRecord(L.join_ev[OTHER_SIDE_STREAMS - 3], q));
            bool listed = false;
            for (uint32_t i = 0; i < n_joins; i++) listed = listed || joins[i] == OTHER_SIDE_STREAMS - 3;
            if (!listed) joins[n_joins++] = OTHER_SIDE_STREAMS - 3;
        }
    }
    if (arena_n) {
        if (arena_maxk > MAX_TILE_ENTRIES) return set_error(WEPP_ELIMIT, "a read inside one genome window lists more than 8192 positions");
        const uint32_t cap = (arena_maxk + 63) & ~63u;
        hipStream_t q = fork ? L.side[OTHER_SIDE_STREAMS - 1] : ps;      // (the last of the sweeps' side streams)
        if (fork) HIP_TRY(hipStreamWaitEvent(q, L.fork_ev, 0));
        int32_t* ps = (int32_t*)(part_base + arena_part);
        uint32_t *pr = (uint32_t*)(ps + (size_t)arena_n * ARENA_CHUNKS), *pc = pr + (size_t)arena_n * ARENA_CHUNKS;
        HIP_TRY(launch_sweep_arena(mat->dev, mat->dev.wc_streams, wsid, d_read_off, d_read_word, root_score, list + arena_off, arena_n, cap,
                                   sweep_lds_bytes(mat->dev.bm_words, cap, 0, false), ps, pr, pc, q));
        HIP_TRY(launch_finalize(mat->dev, d_read_off, d_read_word, list + arena_off, arena_n, ARENA_CHUNKS, ps, pr, pc, d_best_bfs_j, d_score,
                                d_num_best, d_flags, q));
        if (fork) {
            HIP_TRY(hipEventRecord(L.join_ev[OTHER_SIDE_STREAMS - 1], q));
            bool listed = false;
            for (uint32_t i = 0; i < n_joins; i++) listed = listed || joins[i] == OTHER_SIDE_STREAMS - 1;
            if (!listed) joins[n_joins++] = OTHER_SIDE_STREAMS - 1;
        }
        passes += arena_n;
    }
    if (seed_n) {
        if (seed_maxk > SEED_MAX_ENTRIES) return set_error(WEPP_EDEVICE, "routing seeded a sample with too many entries");
        const uint32_t cap = (seed_maxk + 63) & ~63u;
        hipStream_t q = fork ? L.side[OTHER_SIDE_STREAMS - 2] : ps;
        if (fork) HIP_TRY(hipStreamWaitEvent(q, L.fork_ev, 0));
        if (debug_plans) fprintf(stderr, "[plan] seed count=%u maxk=%u chunks=%u lds=%u\n", seed_n, seed_maxk, mat->dev.seed_chunks, seed_lds_bytes(mat->dev, cap));
        HIP_TRY(launch_seed(mat->dev, mat->streams.back(), list + seed_off, seed_n, cap, d_read_off, d_read_word, root_score, d_best_bfs_j, d_score,
                            d_num_best, d_flags, mat->d_work, tun.seed_heavy ? mat->d_seed_heavy : nullptr, q));
        if (fork) {
            HIP_TRY(hipEventRecord(L.join_ev[OTHER_SIDE_STREAMS - 2], q));
            bool listed = false;
            for (uint32_t i = 0; i < n_joins; i++) listed = listed || joins[i] == OTHER_SIDE_STREAMS - 2;
            if (!listed) joins[n_joins++] = OTHER_SIDE_STREAMS - 2;
        }
        passes += seed_n;
    }
    if (n_plain) {
        SweepPlans pl{};
        uint32_t wg = 0, fin = 0, lds_max = 0;
        for (uint32_t k = 0; k < n_plain; k++) {
            const Plan& p = plans[order[k]];
            SweepPlanDev& d = pl.p[k];
            d.st = *p.st;
            d.list = p.lst;
            d.n_list = p.count;
            d.T = p.T;
            d.ntiles = p.ntiles;
            d.bpc = p.bpc;
            d.ent_cap = p.ent_cap;
            d.nchunks = p.nchunks;
            parts(p, d.part_score, d.part_rank, d.part_cnt);
            wg += p.ntiles * p.nchunks;
            d.wg_end = wg;
            fin += finalize_blocks(p.count, p.nchunks);
            d.fin_end = fin;
            lds_max = std::max(lds_max, p.lds_bytes);
        }
        pl.n = n_plain;
        HIP_TRY(launch_sweep_multi(mat->dev, pl, d_read_off, d_read_word, root_score, lds_max, ps));
        HIP_TRY(launch_finalize_multi(mat->dev, pl, d_read_off, d_read_word, d_best_bfs_j, d_score, d_num_best, d_flags,
                                      ps));
    }
    for (uint32_t i = 0; i < n_joins; i++) HIP_TRY(hipStreamWaitEvent(ps, L.join_ev[joins[i]], 0));
    if (blind_walks) {
        // the plan stream and the blind walks' side streams join the caller's stream
        HIP_TRY(hipEventRecord(L.join_ev[PLAN_STREAM], ps));
        HIP_TRY(hipStreamWaitEvent(stream, L.join_ev[PLAN_STREAM], 0));
        HIP_TRY(hipStreamWaitEvent(stream, L.join_ev[MAX_STREAMS - 1], 0));
        HIP_TRY(hipStreamWaitEvent(stream, L.join_ev[MAX_STREAMS - 2], 0));
        if (late16[0]) HIP_TRY(hipStreamWaitEvent(stream, L.join_ev[BLIND16_STREAM], 0));
        if (late16[1]) HIP_TRY(hipStreamWaitEvent(stream, L.join_ev[MAX_STREAMS - 3], 0));
    }
    HIP_TRY(hipEventRecord(mat->ev1[slot], stream));
    {
        std::lock_guard<std::mutex> lk(mat->stat_mu);
        mat->last_passes = passes;
        mat->last_bytes = bytes;
        mat->acc_passes += passes;
        mat->acc_bytes += bytes;
        if (plan_total == n_reads) {            // a call of its own
            mat->last_n_reads = n_reads;
            mat->last_walk_reads = walk_reads;
        } else {                                // a sub-batch (wepp_place_batch reset the sums): complete once every one is in, in any order
            mat->plan_reads_in += n_reads;
            mat->last_walk_reads += walk_reads;
            mat->last_n_reads = mat->plan_reads_in == plan_total ? plan_total : 0;
        }
    }
    return WEPP_OK;
}
}  // namespace

extern "C" int wepp_place_batch_device(wepp_mat_t* mat, const uint32_t* d_read_off, const uint32_t* d_read_word,
                                       uint32_t n_reads, uint64_t n_read_words, uint32_t* d_best_bfs_j,
                                       int32_t* d_score, uint32_t* d_num_best, uint32_t* d_flags,
                                       void* hip_stream) {
    if (!mat || !d_read_off) return set_error(WEPP_EINVAL, "null argument");
    if (n_read_words && !d_read_word) return set_error(WEPP_EINVAL, "null read_word");
    if (n_reads == 0) return WEPP_OK;
    HIP_TRY(hipSetDevice(mat->device));
    return place_device(mat, d_read_off, d_read_word, n_reads, d_best_bfs_j, d_score, d_num_best, d_flags,
                        (hipStream_t)hip_stream, 0, n_reads);
}

extern "C" int wepp_mat_timing_reset(wepp_mat_t* mat) {
    if (!mat) return set_error(WEPP_EINVAL, "null argument");
    mat->n_timed = 0;
    mat->acc_passes = mat->acc_bytes = 0;
    HIP_TRY(hipSetDevice(mat->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(mat->d_work, 0, wepp_mat::D_WORK_BYTES));
    return WEPP_OK;
}

extern "C" int wepp_mat_last_timing(wepp_mat_t* mat, float* mean_sweep_ms, uint32_t* n_calls, uint64_t* passes,
                                    uint64_t* algorithmic_bytes) {
    if (!mat) return set_error(WEPP_EINVAL, "null argument");
    if (mat->n_timed == 0) return set_error(WEPP_EINVAL, "no placement has been launched on this handle since the last reset");
    HIP_TRY(hipSetDevice(mat->device));
    const uint32_t n = (uint32_t)std::min<uint64_t>(mat->n_timed, wepp_mat::kRing);
    double sum = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t slot = (uint32_t)((mat->n_timed - 1 - i) % wepp_mat::kRing);
        HIP_TRY(hipEventSynchronize(mat->ev1[slot]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, mat->ev0[slot], mat->ev1[slot]));
        sum += ms;
    }
    // what the walks asked memory for: counted by their lanes (k_walk: a 32-byte index entry per node event, the
    // sparse-table bytes and range-query aggregates actually requested, list heads, read words, start states of
    // the jobs, results); averaged over the calls since the reset
    unsigned long long wb = 0;
    {
        std::vector<unsigned long long> slots(WALK_COUNTERS);
        HIP_TRY(hipMemcpy(slots.data(), mat->d_work + WALK_COUNTERS, WALK_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (unsigned long long v : slots) wb += v;
    }
    if (mean_sweep_ms) *mean_sweep_ms = (float)(sum / n);
    if (n_calls) *n_calls = n;
    if (passes) *passes = mat->acc_passes / std::max<uint64_t>(1, mat->n_timed);
    if (algorithmic_bytes) *algorithmic_bytes = (mat->acc_bytes + wb) / std::max<uint64_t>(1, mat->n_timed);
    return WEPP_OK;
}

namespace {
// pinned (hipHostMalloc / hipHostRegister) host memory can be the source or the target of a DMA as it stands
bool is_pinned_host(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
}  // namespace

// Host buffers in, host buffers out.  A large batch runs as a PIPELINE of sub-batches (contiguous ranges of the
// reads): the handle's host workers check the read words of sub-batch k+1 and move them into pinned staging while
// sub-batch k's words go up on one copy stream, the kernels of k-1 run on the compute stream, the results of k-2
// come down on a second copy stream and the workers move those of k-3 out to the caller's arrays.  Buffers the
// caller has pinned are used as they are (no staging pass).  The results are the ones of the unsplit call: a
// read's placement does not depend on its batch (tests/test_gpu_parity.py::test_pipeline_equals_unsplit_call).
extern "C" int wepp_place_batch(wepp_mat_t* mat, const uint32_t* read_off, const uint32_t* read_word,
                                uint32_t n_reads, uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best,
                                uint32_t* flags, int32_t* per_node_scores) {
    if (!mat || !read_off) return set_error(WEPP_EINVAL, "null argument");
    if (per_node_scores && (uint64_t)n_reads * mat->dev.N > (1ull << 31))
        return set_error(WEPP_ELIMIT, "per_node_scores: n_reads * n_nodes exceeds 2^31 values; split the batch");
    if (n_reads == 0) return WEPP_OK;
    if (read_off[0] != 0) return set_error(WEPP_EINVAL, "read_off[0] must be 0");
    const uint64_t nw = read_off[n_reads];
    if (nw && !read_word) return set_error(WEPP_EINVAL, "null read_word");
    HIP_TRY(hipSetDevice(mat->device));
    // WEPP_DEBUG_TIMING=1: wall time of the call's phases to stderr
    const bool dbg_time = mat->tun.debug_timing;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };

    hipError_t e = hipSuccess;
    if (!mat->pipe_h2d[0]) {
        for (hipStream_t& st : mat->pipe_h2d)
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (hipStream_t& st : mat->pipe_compute)
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (hipStream_t& st : mat->pipe_d2h)
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (uint32_t i = 0; i < wepp_mat::kPipeMax && e == hipSuccess; i++) {
            e = hipEventCreateWithFlags(&mat->pipe_up[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&mat->pipe_done[i], hipEventDisableTiming);
        }
        for (uint32_t i = 0; i < 4 * wepp_mat::kPipeMax && e == hipSuccess; i++)
            e = hipEventCreateWithFlags(&mat->pipe_out[i], hipEventDisableTiming);
        if (e != hipSuccess) return hip_fail(e, "pipeline streams / events");
    }
    // ---- buffers: grow-only, on the handle (no hipMalloc / hipFree per call) ----
    const size_t off_bytes = (((size_t)(n_reads + 1) * 4) + 255) & ~(size_t)255;
    const size_t in_need = off_bytes + std::max<size_t>(nw * 4, 16), out_need = (size_t)n_reads * 16;
    const bool in_pinned = is_pinned_host(read_off) && (nw == 0 || is_pinned_host(read_word));
    void* dst[4] = {best_bfs_j, score, num_best, flags};
    bool out_pinned[4];
    bool any_staged_out = false;
    for (int i = 0; i < 4; i++) { out_pinned[i] = dst[i] && is_pinned_host(dst[i]); any_staged_out = any_staged_out || (dst[i] && !out_pinned[i]); }
    {
        auto grow_dev = [&](void*& p, size_t& have, size_t need) -> hipError_t {
            if (need <= have) return hipSuccess;
            if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); p = nullptr; have = 0; }
            hipError_t r = hipMalloc(&p, need + need / 4);
            if (r == hipSuccess) have = need + need / 4;
            return r;
        };
        auto grow_pin = [&](void*& p, size_t& have, size_t need) -> hipError_t {
            if (need <= have) return hipSuccess;
            if (p) { (void)hipDeviceSynchronize(); (void)hipHostFree(p); p = nullptr; have = 0; }
            hipError_t r = hipHostMalloc(&p, need + need / 4, hipHostMallocDefault);
            if (r == hipSuccess) have = need + need / 4;
            return r;
        };
        e = grow_dev(mat->io_in, mat->io_in_bytes, in_need);
        if (e == hipSuccess) e = grow_dev(mat->io_out, mat->io_out_bytes, out_need);
        if (e == hipSuccess && !in_pinned) e = grow_pin(mat->pin, mat->pin_bytes, in_need);
        if (e == hipSuccess && any_staged_out) e = grow_pin(mat->pin_out, mat->pin_out_bytes, out_need);
        if (e != hipSuccess) return set_error(WEPP_ENOMEM, std::string("device / pinned buffers of the batch: ") + hipGetErrorString(e));
    }
    uint32_t* const d_off = (uint32_t*)mat->io_in;
    uint32_t* const d_word = (uint32_t*)((char*)mat->io_in + off_bytes);
    uint32_t* const d_out = (uint32_t*)mat->io_out;
    uint32_t* const pin_off = in_pinned ? nullptr : (uint32_t*)mat->pin;
    uint32_t* const pin_word = in_pinned ? nullptr : (uint32_t*)((char*)mat->pin + off_bytes);
    uint32_t* const po = (uint32_t*)mat->pin_out;

    // ---- sub-batches and tasks ----
    const bool big = n_reads >= (1u << 16);
    if (big && !mat->pool) {
        // this handle's share of the host threads the process may use (affinity mask and cgroup quota, host_pool.hpp),
        // one of them being the calling thread: the handles alive now split them (8 device threads of one C++ host
        // process, or one handle per rank when every rank is a process with its own cpuset); WEPP_HOST_THREADS fixes it
        const uint32_t share = mat->tun.host_threads ? mat->tun.host_threads
                                                     : usable_host_threads() / std::max(1u, g_handles_alive.load(std::memory_order_relaxed));
        mat->pool.reset(new HostPool(std::min<uint32_t>(15, share > 1 ? share - 1 : 1)));
    }
    const uint32_t pipe_env = mat->tun.pipe_sub_batches;
    const uint32_t pipe_knob = mat->pipe_sub_batches ? mat->pipe_sub_batches : pipe_env;
    // S sub-batches = device calls.  A device call is a chain of ~20 short launches with a host round trip in the
    // middle (routing counters): ~0.3 ms whatever its size up to a million short reads, most of it the HOST's share,
    // which two calls cannot overlap -- so a batch is only cut into several calls when each keeps >= 2 M reads
    // (measured, profiles/r3_experiments/pcie_pipeline.txt: 1 M reads in 2 / 4 / 8 calls 1.41 / 1.98 / 3.07 ms against
    // 1.28 in one).  C copy chunks >= S: the reads are checked, staged and sent up chunk by chunk, whatever S.
    // (round 4: with two launch threads a call of half a million reads or more goes as two halves, 1.08 -> 1.00 ms per
    // 1 M reads; three or four are slower again: the host's HIP calls per device call are what runs out)
    const bool can_pair = big && mat->pool && mat->pool->workers() >= 2;
    uint32_t S = !big || per_node_scores ? 1u : pipe_knob ? pipe_knob : std::max<uint32_t>((can_pair && n_reads >= (1u << 19)) ? 2u : 1u, n_reads >> 21);
    S = std::min<uint32_t>(S, wepp_mat::kPipeMax);
    const uint32_t min_chunks = mat->tun.pipe_min_chunks;
    const uint32_t C = big ? S * ((min_chunks + S - 1) / S) : 1;               // copy chunks (a multiple of S, >= 4 unless WEPP_PIPE_MIN_CHUNKS says otherwise)
    const uint32_t CPS = C / S;                                                // chunks per sub-batch
    const uint32_t threads = big ? mat->pool->workers() + 1 : 1;
    const uint32_t PS = big ? std::max(1u, (4 * threads + C - 1) / C) : 1;     // staging parts per chunk
    const uint32_t PO = big ? 4 : 1;                                           // copy-out parts per (sub-batch, array)
    // (several sub-batches, every result array wanted and none in memory the caller pinned: one copy per sub-batch)
    const bool merged_out = S > 1 && dst[0] && dst[1] && dst[2] && dst[3] && !out_pinned[0] && !out_pinned[1] && !out_pinned[2] && !out_pinned[3];
    auto chunk_lo = [&](uint32_t c) { return (uint32_t)((uint64_t)n_reads * c / C); };
    auto sub_lo = [&](uint32_t k) { return chunk_lo(k * CPS); };
    const uint32_t n_stage = C * PS, n_copy = any_staged_out ? S * 4 * PO : 0;

    // preconditions of the reference's merge (usher_mapper.cpp:205-243): sorted, unique positions.  Checked by the
    // staging tasks, each moving its reads into the pinned staging buffer as it goes (one pass over the input); the
    // first offending read (lowest index) is reported.
    // exact, read by read: only run on a range the fast pass below found something in
    auto check_exact = [&](uint32_t lo, uint32_t hi, uint32_t& bad, int& what) {
        for (uint32_t r = lo; r < hi; r++) {
            if (read_off[r + 1] < read_off[r] || read_off[r + 1] > nw) { bad = r; what = 0; return; }
            for (uint32_t k = read_off[r] + 1; k < read_off[r + 1]; k++)
                if ((read_word[k] & 0xFFFFFu) <= (read_word[k - 1] & 0xFFFFFu)) { bad = r; what = 1; return; }
            for (uint32_t k = read_off[r]; k < read_off[r + 1]; k++)
                if (((read_word[k] >> 24) & 15u) == 0 || ((read_word[k] >> 20) & 15u) == 0) { bad = r; what = 2; return; }
        }
    };
    // fast pass over a range of reads: branch-free loops the compiler vectorises.  Offsets: monotone and inside
    // the word array.  Words: no zero mask; positions ascend from one word to the next except where a read
    // starts -- the descents are counted over all words and over the read starts, and must be the same number.
    auto check = [&](uint32_t lo, uint32_t hi, uint32_t& bad, int& what) {
        uint32_t off_bad = (read_off[lo] > nw) ? 1u : 0u;
        for (uint32_t r = lo; r < hi; r++) {
            const uint32_t a = read_off[r], b = read_off[r + 1];
            off_bad |= (uint32_t)(b < a) | (uint32_t)(b > nw);
        }
        if (off_bad) { check_exact(lo, hi, bad, what); return; }
        const uint32_t a0 = read_off[lo], b0 = read_off[hi];
        uint32_t zero = 0, descents = 0, at_starts = 0;
        if (pin_word) {
            for (uint32_t k = a0; k < b0; k++) {
                const uint32_t w = read_word[k];
                zero |= (uint32_t)(((w >> 24) & 15u) == 0) | (uint32_t)(((w >> 20) & 15u) == 0);
                pin_word[k] = w;
            }
        } else {
            for (uint32_t k = a0; k < b0; k++) {
                const uint32_t w = read_word[k];
                zero |= (uint32_t)(((w >> 24) & 15u) == 0) | (uint32_t)(((w >> 20) & 15u) == 0);
            }
        }
        for (uint32_t k = a0 + 1; k < b0; k++) descents += (uint32_t)((read_word[k] & 0xFFFFFu) <= (read_word[k - 1] & 0xFFFFFu));
        for (uint32_t r = lo + 1; r < hi; r++) {
            const uint32_t s0 = read_off[r];
            if (read_off[r + 1] > s0 && s0 > a0) at_starts += (uint32_t)((read_word[s0] & 0xFFFFFu) <= (read_word[s0 - 1] & 0xFFFFFu));
        }
        if (zero || descents != at_starts) check_exact(lo, hi, bad, what);
    };
    std::vector<uint32_t> bad(n_stage, 0xFFFFFFFFu);
    std::vector<int> what(n_stage, 0);
    std::vector<std::atomic<uint32_t>> staged(C);          // staging parts of a chunk that are done
    std::vector<std::atomic<int>> launched(S);             // 0: not yet; 1: its D2H is enqueued (pipe_out[k] recorded); -1: never will be
    for (uint32_t c = 0; c < C; c++) staged[c].store(0);
    for (uint32_t k = 0; k < S; k++) launched[k].store(0);
    std::vector<hipError_t> copy_err(std::max<uint32_t>(n_copy, 1), hipSuccess);
    auto stage_task = [&](uint32_t t) {
        const uint32_t ck = t / PS, part = t % PS;
        const uint32_t lo = chunk_lo(ck), hi = chunk_lo(ck + 1);
        const uint32_t a = lo + (uint32_t)((uint64_t)(hi - lo) * part / PS), b = lo + (uint32_t)((uint64_t)(hi - lo) * (part + 1) / PS);
        // The offsets this task stages: its reads' [a, b), and -- the LAST part of a chunk -- the chunk's end offset too:
        // a chunk's H2D copy takes the offsets [first read, last read + 1] and is enqueued as soon as the chunk's OWN
        // tasks are done.  (Round 3 left the end offset to the first task of the NEXT chunk, which the launch thread
        // does not wait for: with few host workers the copy went up before it was staged -- a stale or uninitialised
        // offset for the read at every chunk boundary, wrong entries for that read or a fault in k_route.)  The first
        // part of a later chunk leaves its first offset to the chunk before: every element has one writer.
        if (pin_off) {
            const uint32_t ca = (part == 0 && ck > 0) ? a + 1 : a, cb = (part + 1 == PS) ? b + 1 : b;
            if (cb > ca) std::memcpy(pin_off + ca, read_off + ca, (size_t)(cb - ca) * 4);
        }
        if (b > a) check(a, b, bad[t], what[t]);
        staged[ck].fetch_add(1, std::memory_order_release);
    };
    auto copy_task = [&](uint32_t t) {          // t in [0, n_copy): (sub-batch, array, part)
        const uint32_t k = t / (4 * PO), i = (t / PO) % 4, part = t % PO;
        if (!dst[i] || out_pinned[i]) return;
        int st;
        while ((st = launched[k].load(std::memory_order_acquire)) == 0) std::this_thread::yield();
        if (st < 0) return;
        (void)hipSetDevice(mat->device);                              // (a worker thread starts on device 0)
        copy_err[t] = hipEventSynchronize(mat->pipe_out[k * 4 + (merged_out ? 0u : i)]);  // this array of the sub-batch has landed (the next one is on the bus)
        if (copy_err[t] != hipSuccess) return;
        const size_t lo = sub_lo(k), hi = sub_lo(k + 1);
        const size_t a = lo + (hi - lo) * part / PO, b = lo + (hi - lo) * (part + 1) / PO;
        std::memcpy((uint32_t*)dst[i] + a, po + (size_t)i * n_reads + a, (b - a) * 4);
    };
    double t_launched = 0;

    // ---- the launch sequence: H2D, kernels, D2H of every sub-batch ----
    // TWO launchers when the call has several sub-batches and the pool at least two workers: this thread takes the
    // even sub-batches (lane 0), a pool worker the odd ones (lane 1).  A device call keeps its host thread until its
    // routing counters are back (place_device polls for them); on one thread the kernels of sub-batch k + 1 were only
    // enqueued once call k had returned, so the lanes never overlapped (1 M reads in 2 / 4 calls: 1.04 / 1.10 ms against
    // 1.05 in one).  Every launcher uploads its own sub-batches, the one after the next before the device call of this one.
    const bool two_launchers = big && S > 1 && mat->pool && mat->pool->workers() >= 2;
    struct Launcher { int rc = WEPP_OK; std::string msg; };
    Launcher launcher[2];
    double t_launch[wepp_mat::kPipeMax] = {};
    std::atomic<bool> rejected{false};
    bool up_ok[wepp_mat::kPipeMax] = {};
    hipError_t up_he[wepp_mat::kPipeMax] = {};
    // the words of sub-batch k go up chunk by chunk, each as soon as its reads are checked and staged; up_ok[k] = none
    // of its reads was rejected
    auto upload = [&](uint32_t k, const Launcher& me) {
        const uint32_t* src_off = in_pinned ? read_off : pin_off;
        const uint32_t* src_word = in_pinned ? read_word : pin_word;
        hipError_t he = hipSuccess;
        bool ok = me.rc == WEPP_OK && !rejected.load(std::memory_order_acquire);
        for (uint32_t ck = k * CPS; ck < (k + 1) * CPS && ok && he == hipSuccess; ck++) {
            while (staged[ck].load(std::memory_order_acquire) < PS) std::this_thread::yield();
            // (the first offset of a sub-batch is staged with the chunk before it)
            if (ck == k * CPS && ck > 0) while (staged[ck - 1].load(std::memory_order_acquire) < PS) std::this_thread::yield();
            for (uint32_t t = ck * PS; t < (ck + 1) * PS && ok; t++) ok = bad[t] == 0xFFFFFFFFu;
            if (!ok) break;
            const uint32_t clo = chunk_lo(ck), chi = chunk_lo(ck + 1);
            const uint32_t w0 = read_off[clo], w1 = read_off[chi];
            // offset `clo` went up with the chunk before (the kernels of the sub-batch before may be reading it) -- but
            // the first chunk of a SUB-BATCH sends its own too: the chunk before belongs to the other launcher, whose
            // copy may not be enqueued yet (the same four bytes written twice)
            const uint32_t o0 = (ck && ck != k * CPS) ? clo + 1 : clo;
            he = hipMemcpyAsync(d_off + o0, src_off + o0, (size_t)(chi + 1 - o0) * 4, hipMemcpyHostToDevice, mat->pipe_h2d[k % wepp_mat::kLanes]);
            if (he == hipSuccess && w1 > w0)
                he = hipMemcpyAsync(d_word + w0, src_word + w0, (size_t)(w1 - w0) * 4, hipMemcpyHostToDevice, mat->pipe_h2d[k % wepp_mat::kLanes]);
        }
        if (ok && he == hipSuccess) he = hipEventRecord(mat->pipe_up[k], mat->pipe_h2d[k % wepp_mat::kLanes]);
        if (!ok) rejected.store(true, std::memory_order_release);      // nothing of a rejected batch reaches a kernel, and nothing behind it is worth placing
        up_ok[k] = ok;
        up_he[k] = he;
    };
    auto launch_from = [&](uint32_t first, uint32_t step, Launcher& me) {
        if (first >= S) return;
        (void)hipSetDevice(mat->device);                              // (a worker thread starts on device 0)
        upload(first, me);
        for (uint32_t k = first; k < S; k += step) {
            const uint32_t lo = sub_lo(k), hi = sub_lo(k + 1);
            if (k + step < S) upload(k + step, me);
            hipError_t he = up_he[k];
            if (!up_ok[k]) {
                launched[k].store(-1, std::memory_order_release);
                continue;
            }
            const uint32_t ln = k % wepp_mat::kLanes;             // (sub-batches alternate between the handle's two lanes)
            hipStream_t cs = mat->pipe_compute[ln];
            if (he == hipSuccess) he = hipStreamWaitEvent(cs, mat->pipe_up[k], 0);
            if (he == hipSuccess) {
                // (the offsets of a sub-batch index the whole call's word array: no rebasing)
                const int prc = place_device(mat, d_off + lo, d_word, hi - lo, d_out + lo, (int32_t*)(d_out + n_reads) + lo,
                                             d_out + 2 * (size_t)n_reads + lo, d_out + 3 * (size_t)n_reads + lo, cs, lo, n_reads, ln);
                if (prc != WEPP_OK) { me.rc = prc; me.msg = wepp_last_error(); launched[k].store(-1, std::memory_order_release); continue; }
            }
            if (he == hipSuccess) he = hipEventRecord(mat->pipe_done[k], cs);
            hipStream_t ds = mat->pipe_d2h[ln];
            if (he == hipSuccess) he = hipStreamWaitEvent(ds, mat->pipe_done[k], 0);
            if (merged_out) {
                // the four result arrays of the sub-batch in ONE (2-D) copy into the staging buffer, which has the device
                // buffer's layout: a copy and an event are host calls of ~5 us each, and the host's calls are what a
                // call split into sub-batches runs out of
                if (he == hipSuccess) he = hipMemcpy2DAsync(po + lo, (size_t)n_reads * 4, d_out + lo, (size_t)n_reads * 4, (size_t)(hi - lo) * 4, 4, hipMemcpyDeviceToHost, ds);
                if (he == hipSuccess) he = hipEventRecord(mat->pipe_out[k * 4], ds);
            } else
            for (int i = 0; i < 4 && he == hipSuccess; i++) {
                if (!dst[i]) continue;
                uint32_t* to = out_pinned[i] ? (uint32_t*)dst[i] : po + (size_t)i * n_reads;
                he = hipMemcpyAsync(to + lo, d_out + (size_t)i * n_reads + lo, (size_t)(hi - lo) * 4, hipMemcpyDeviceToHost, ds);
                if (he == hipSuccess) he = hipEventRecord(mat->pipe_out[k * 4 + i], ds);   // an array is moved out while the next one comes down
            }
            if (he != hipSuccess) {
                me.rc = hip_fail(he, "H2D / kernels / D2H of a sub-batch");
                me.msg = wepp_last_error();
                launched[k].store(-1, std::memory_order_release);
                continue;
            }
            launched[k].store(1, std::memory_order_release);
            if (dbg_time) t_launch[k] = since();
        }
    };
    {
        // (the sub-batches' plan ids land in one array sized for the call: grown here, before two threads would)
        const int prc = S > 1 ? ensure_plan_buffers(mat, n_reads) : WEPP_OK;
        if (prc != WEPP_OK) return prc;
        std::lock_guard<std::mutex> lk(mat->stat_mu);
        mat->plan_reads_in = 0;
        mat->last_walk_reads = 0;
        mat->last_n_reads = 0;
    }
    if (big) {
        // two rounds on the pool: the staging tasks (and the second launcher, first in line) while this thread launches;
        // the copy-out tasks once every sub-batch's copies are enqueued.  (One round of both had the workers that ran
        // out of staging tasks spin -- yield in a loop -- for their sub-batch's launch: fifteen spinning threads beside
        // the launch thread, on a container of sixteen cores, delayed the thread they were waiting for.)
        const uint32_t extra = two_launchers ? 1u : 0u;
        const std::function<void(uint32_t)> stage_only = [&](uint32_t t) {
            if (t < extra) launch_from(1, 2, launcher[1]);
            else stage_task(t - extra);
        };
        mat->pool->start(n_stage + extra, stage_only);
        launch_from(0, two_launchers ? 2 : 1, launcher[0]);
        mat->pool->finish();
        if (dbg_time) t_launched = since();
        if (n_copy) {
            const std::function<void(uint32_t)> copy_only = [&](uint32_t t) { copy_task(t); };
            mat->pool->run(n_copy, copy_only);
        }
    } else {
        for (uint32_t t = 0; t < n_stage; t++) stage_task(t);
        launch_from(0, 1, launcher[0]);
        for (uint32_t t = 0; t < n_copy; t++) copy_task(t);
    }
    int rc = launcher[0].rc != WEPP_OK ? launcher[0].rc : launcher[1].rc;
    std::string rc_msg = launcher[0].rc != WEPP_OK ? launcher[0].msg : launcher[1].msg;
    // everything this call put on the handle's streams has finished before it returns -- also after an error, and
    // when no result array was asked for (the staging buffers belong to the next call then)
    {
        hipError_t se = hipSuccess;
        for (hipStream_t st : {mat->pipe_d2h[0], mat->pipe_d2h[1], mat->pipe_compute[0], mat->pipe_compute[1], mat->pipe_h2d[0], mat->pipe_h2d[1]}) {
            const hipError_t r = hipStreamSynchronize(st);
            if (se == hipSuccess) se = r;
        }
        if (se != hipSuccess && rc == WEPP_OK) { rc = hip_fail(se, "placement kernels / copies"); rc_msg = wepp_last_error(); }
    }
    for (uint32_t t = 0; t < n_stage; t++) {
        if (bad[t] == 0xFFFFFFFFu) continue;
        if (what[t] == 0) return set_error(WEPP_EINVAL, "read_off not monotone");
        if (what[t] == 1)
            return set_error(WEPP_EINVAL, "read " + std::to_string(bad[t]) + ": entries must be sorted by position with unique positions");
        return set_error(WEPP_EINVAL, "read " + std::to_string(bad[t]) + ": zero nucleotide mask");
    }
    if (rc != WEPP_OK) return set_error(rc, rc_msg);
    for (uint32_t t = 0; t < n_copy; t++)
        if (copy_err[t] != hipSuccess) return hip_fail(copy_err[t], "D2H copy of the results");
    if (per_node_scores) {
        int32_t* d_pns = nullptr;
        const size_t nb = (size_t)n_reads * mat->dev.N * sizeof(int32_t);
        e = hipMalloc((void**)&d_pns, nb);
        if (e != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc per_node_scores: ") + hipGetErrorString(e));
        e = launch_scores(mat->dev, mat->streams.back(), d_off, d_word, n_reads, d_pns, nullptr);   // (everything above has been synchronised)
        if (e == hipSuccess) e = hipMemcpy(per_node_scores, d_pns, nb, hipMemcpyDeviceToHost);
        (void)hipFree(d_pns);
        if (e != hipSuccess) return hip_fail(e, "per-node score kernel");
    }
    if (dbg_time) {
        fprintf(stderr, "[place_batch] %u reads in %u sub-batches (%s in, %s out): D2H of sub-batch k enqueued at", n_reads, S,
                in_pinned ? "caller-pinned" : "staged", any_staged_out ? "staged" : "caller-pinned");
        for (uint32_t k = 0; k < S; k++) fprintf(stderr, " %.3f", t_launch[k]);
        fprintf(stderr, " ms; every launch enqueued at %.3f ms; total %.3f ms\n", t_launched, since());
    }
    return WEPP_OK;
}

namespace {

// shared argument check of the calls that take a read CSR from the host: offsets start at 0, are monotone and
// stay inside the word array (whose size is read_off[n_reads])
int check_read_csr(const uint32_t* read_off, const uint32_t* read_word, uint32_t n_reads) {
    if (!read_off) return set_error(WEPP_EINVAL, "null read_off");
    if (read_off[0] != 0) return set_error(WEPP_EINVAL, "read_off[0] must be 0");
    for (uint32_t r = 0; r < n_reads; r++)
        if (read_off[r + 1] < read_off[r]) return set_error(WEPP_EINVAL, "read_off not monotone");
    if (read_off[n_reads] && !read_word) return set_error(WEPP_EINVAL, "null read_word");
    return WEPP_OK;
}

// Grow-only buffers of the handle for the pass-2 calls: `dev_bytes` of device memory in io_in and `pin_bytes`
// of pinned staging (the same buffers wepp_place_batch uses; the calls on a handle are serial).
int reserve_io(wepp_mat* mat, size_t dev_bytes, size_t pin_need) {
    if (dev_bytes > mat->io_in_bytes) {
        if (mat->io_in) { (void)hipFree(mat->io_in); mat->io_in = nullptr; mat->io_in_bytes = 0; }
        hipError_t e = hipMalloc(&mat->io_in, dev_bytes + dev_bytes / 4);
        if (e != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        mat->io_in_bytes = dev_bytes + dev_bytes / 4;
    }
    if (pin_need > mat->pin_bytes) {
        if (mat->pin) { (void)hipHostFree(mat->pin); mat->pin = nullptr; mat->pin_bytes = 0; }
        hipError_t e = hipHostMalloc(&mat->pin, pin_need + pin_need / 4, hipHostMallocDefault);
        if (e != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        mat->pin_bytes = pin_need + pin_need / 4;
    }
    return WEPP_OK;
}

inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

extern "C" int wepp_imputed_mutations(wepp_mat_t* mat, const uint32_t* read_off, const uint32_t* read_word,
                                      uint32_t n_reads, const uint32_t* best_bfs_j, uint32_t* imp_off,
                                      int32_t* imp_pos, uint8_t* imp_nuc, uint64_t capacity) {
    if (!mat || !read_off || !best_bfs_j || !imp_off) return set_error(WEPP_EINVAL, "null argument");
    {
        int rc = check_read_csr(read_off, read_word, n_reads);
        if (rc != WEPP_OK) return rc;
    }
    std::vector<uint32_t> pairs;
    imp_off[0] = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        if (best_bfs_j[r] >= mat->dev.N) return set_error(WEPP_EINVAL, "best_bfs_j out of range");
        for (uint32_t k = read_off[r]; k < read_off[r + 1]; k++) {
            const uint32_t w = read_word[k], a = (w >> 24) & 15u;
            if (!((w >> 28) & 1u) && (a & (a - 1))) { pairs.push_back(r); pairs.push_back(k); }
        }
        imp_off[r + 1] = (uint32_t)(pairs.size() / 2);
    }
    const uint32_t np = (uint32_t)(pairs.size() / 2);
    if (np > capacity) return set_error(WEPP_ELIMIT, "imputed-mutation buffers too small: need " + std::to_string(np));
    if (np == 0) return WEPP_OK;
    if (!imp_pos || !imp_nuc) return set_error(WEPP_EINVAL, "null argument");
    for (uint32_t i = 0; i < np; i++) imp_pos[i] = (int32_t)(read_word[pairs[2 * i + 1]] & 0xFFFFFu);
    HIP_TRY(hipSetDevice(mat->device));
    // one staged upload (offsets | words | placements | pairs) and one download (nucleotides) through the
    // handle's pinned buffer; device memory from its grow-only input buffer
    const uint64_t nw = read_off[n_reads];
    const size_t b_off = pad256((size_t)(n_reads + 1) * 4), b_word = pad256(std::max<size_t>(nw * 4, 16));
    const size_t b_best = pad256((size_t)n_reads * 4), b_pairs = pad256((size_t)np * 8), b_nuc = pad256(np);
    const size_t up = b_off + b_word + b_best + b_pairs;
    {
        int rc = reserve_io(mat, up + b_nuc, up);
        if (rc != WEPP_OK) return rc;
    }
    char* hp = (char*)mat->pin;
    std::memcpy(hp, read_off, (size_t)(n_reads + 1) * 4);
    if (nw) std::memcpy(hp + b_off, read_word, nw * 4);
    std::memcpy(hp + b_off + b_word, best_bfs_j, (size_t)n_reads * 4);
    std::memcpy(hp + b_off + b_word + b_best, pairs.data(), (size_t)np * 8);
    char* dp = (char*)mat->io_in;
    HIP_TRY(hipMemcpy(dp, hp, up, hipMemcpyHostToDevice));
    uint8_t* d_nuc = (uint8_t*)(dp + up);
    HIP_TRY(launch_imputed(mat->dev, (const uint32_t*)dp, (const uint32_t*)(dp + b_off), (const uint32_t*)(dp + b_off + b_word),
                           (const uint32_t*)(dp + b_off + b_word + b_best), np, d_nuc, nullptr));
    hipError_t e = hipMemcpy(hp, d_nuc, np, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "imputed-mutation kernel");
    std::memcpy(imp_nuc, hp, np);
    return WEPP_OK;
}

extern "C" int wepp_excess_mutations(wepp_mat_t* mat, const uint32_t* read_off, const uint32_t* read_word,
                                     uint32_t n_reads, uint32_t n_pairs, const uint32_t* pair_read,
                                     const uint32_t* pair_bfs_j, uint64_t* exc_off, int32_t* exc_pos, uint8_t* exc_ref,
                                     uint8_t* exc_par, uint8_t* exc_mut, uint64_t capacity) {
    if (!mat || !read_off || !exc_off) return set_error(WEPP_EINVAL, "null argument");
    exc_off[0] = 0;
    if (n_pairs == 0) return WEPP_OK;
    if (!pair_read || !pair_bfs_j) return set_error(WEPP_EINVAL, "null argument");
    {
        int rc = check_read_csr(read_off, read_word, n_reads);
        if (rc != WEPP_OK) return rc;
    }
    for (uint32_t i = 0; i < n_pairs; i++) {
        if (pair_read[i] >= n_reads) return set_error(WEPP_EINVAL, "pair_read out of range");
        if (pair_bfs_j[i] >= mat->dev.N) return set_error(WEPP_EINVAL, "pair_bfs_j out of range");
    }
    const uint64_t nw = read_off[n_reads];
    HIP_TRY(hipSetDevice(mat->device));
    // upload (offsets | words | pair reads | pair nodes), count, size the output, emit: all through the
    // handle's grow-only device buffer and pinned staging buffer
    const size_t b_off = pad256((size_t)(n_reads + 1) * 4), b_word = pad256(std::max<size_t>(nw * 4, 16));
    const size_t b_pair = pad256((size_t)n_pairs * 4), b_ooff = pad256((size_t)n_pairs * 8);
    const size_t up = b_off + b_word + 2 * b_pair;
    const size_t fixed = up + b_pair /* counts */ + b_ooff;
    {
        int rc = reserve_io(mat, fixed, std::max(up, b_ooff));
        if (rc != WEPP_OK) return rc;
    }
    char* hp = (char*)mat->pin;
    std::memcpy(hp, read_off, (size_t)(n_reads + 1) * 4);
    if (nw) std::memcpy(hp + b_off, read_word, nw * 4);
    std::memcpy(hp + b_off + b_word, pair_read, (size_t)n_pairs * 4);
    std::memcpy(hp + b_off + b_word + b_pair, pair_bfs_j, (size_t)n_pairs * 4);
    char* dp = (char*)mat->io_in;
    HIP_TRY(hipMemcpy(dp, hp, up, hipMemcpyHostToDevice));
    const uint32_t *d_off = (const uint32_t*)dp, *d_word = (const uint32_t*)(dp + b_off);
    const uint32_t *d_pr = (const uint32_t*)(dp + b_off + b_word), *d_pj = (const uint32_t*)(dp + b_off + b_word + b_pair);
    uint32_t* d_cnt = (uint32_t*)(dp + up);
    unsigned long long* d_ooff = (unsigned long long*)(dp + up + b_pair);
    HIP_TRY(launch_excess(mat->dev, d_off, d_word, d_pr, d_pj, n_pairs, nullptr, d_cnt, nullptr, nullptr));
    hipError_t e = hipMemcpy(hp, d_cnt, (size_t)n_pairs * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "excess-mutation count");
    uint64_t total = 0;
    {
        const uint32_t* counts = (const uint32_t*)hp;
        for (uint32_t i = 0; i < n_pairs; i++) { total += counts[i]; exc_off[i + 1] = total; }
    }
    if (total > capacity) return set_error(WEPP_ELIMIT, "excess-mutation buffers too small: need " + std::to_string(total));
    if (total == 0) return WEPP_OK;
    if (!exc_pos || !exc_ref || !exc_par || !exc_mut) return set_error(WEPP_EINVAL, "null output array");
    {
        // the device buffer may move when it grows: the inputs are uploaded again behind the new base
        const bool grows = fixed + pad256(total * 4) > mat->io_in_bytes;
        int rc = reserve_io(mat, fixed + pad256(total * 4), std::max({up, b_ooff, (size_t)total * 4}));
        if (rc != WEPP_OK) return rc;
        hp = (char*)mat->pin;
        if (grows) {
            std::memcpy(hp, read_off, (size_t)(n_reads + 1) * 4);
            if (nw) std::memcpy(hp + b_off, read_word, nw * 4);
            std::memcpy(hp + b_off + b_word, pair_read, (size_t)n_pairs * 4);
            std::memcpy(hp + b_off + b_word + b_pair, pair_bfs_j, (size_t)n_pairs * 4);
            dp = (char*)mat->io_in;
            HIP_TRY(hipMemcpy(dp, hp, up, hipMemcpyHostToDevice));
            d_off = (const uint32_t*)dp; d_word = (const uint32_t*)(dp + b_off);
            d_pr = (const uint32_t*)(dp + b_off + b_word); d_pj = (const uint32_t*)(dp + b_off + b_word + b_pair);
            d_cnt = (uint32_t*)(dp + up);
            d_ooff = (unsigned long long*)(dp + up + b_pair);
        }
    }
    {
        unsigned long long* ho = (unsigned long long*)hp;
        for (uint32_t i = 0; i < n_pairs; i++) ho[i] = exc_off[i];
        HIP_TRY(hipMemcpy(d_ooff, ho, (size_t)n_pairs * 8, hipMemcpyHostToDevice));
    }
    uint32_t* d_out = (uint32_t*)(dp + fixed);
    HIP_TRY(launch_excess(mat->dev, d_off, d_word, d_pr, d_pj, n_pairs, d_ooff, d_cnt, d_out, nullptr));
    e = hipMemcpy(hp, d_out, total * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "excess-mutation kernel");
    const uint32_t* packed = (const uint32_t*)hp;
    for (uint64_t q = 0; q < total; q++) {
        exc_pos[q] = (int32_t)(packed[q] & 0xFFFFFu);
        exc_ref[q] = (uint8_t)((packed[q] >> 20) & 15u);
        exc_par[q] = (uint8_t)((packed[q] >> 24) & 15u);
        exc_mut[q] = (uint8_t)((packed[q] >> 28) & 15u);
    }
    return WEPP_OK;
}

// best_j_vec: every optimal node of every read (usher_common.cpp:376-381, 413-446)
extern "C" int wepp_best_nodes(wepp_mat_t* mat, const uint32_t* read_off, const uint32_t* read_word, uint32_t n_reads,
                               const int32_t* score, const uint32_t* num_best, uint64_t* best_off, uint32_t* best_nodes,
                               uint64_t capacity) {
    if (!mat || !read_off || !score || !num_best || !best_off) return set_error(WEPP_EINVAL, "null argument");
    {
        int rc = check_read_csr(read_off, read_word, n_reads);
        if (rc != WEPP_OK) return rc;
    }
    best_off[0] = 0;
    for (uint32_t r = 0; r < n_reads; r++) best_off[r + 1] = best_off[r] + num_best[r];
    const uint64_t total = best_off[n_reads];
    if (total > capacity) return set_error(WEPP_ELIMIT, "best-node list too small: need " + std::to_string(total));
    if (n_reads == 0 || total == 0) return WEPP_OK;
    if (!best_nodes) return set_error(WEPP_EINVAL, "null argument");
    if (total >= (1ull << 32)) return set_error(WEPP_ELIMIT, "more than 2^32 optimal nodes in one call; split the batch");
    const uint64_t nw = read_off[n_reads];
    HIP_TRY(hipSetDevice(mat->device));
    // device buffers (grow-only io_in): offsets | words | scores | CSR offsets | cursors | routing arrays | the list
    const size_t b_off = pad256((size_t)(n_reads + 1) * 4), b_word = pad256(std::max<size_t>(nw * 4, 16));
    const size_t b_r = pad256((size_t)n_reads * 4), b_o64 = pad256((size_t)(n_reads + 1) * 8), b_r1 = pad256(n_reads);
    const size_t up = b_off + b_word + b_r + b_o64;                       // uploaded in one staged copy
    const size_t fixed = up + b_r /* cursors */ + b_r1 /* plan ids */ + 4 * b_r /* list, root score, slot, jobs */;
    {
        int rc = reserve_io(mat, fixed + pad256(total * 4), std::max(up, (size_t)total * 4));
        if (rc != WEPP_OK) return rc;
    }
    char* hp = (char*)mat->pin;
    std::memcpy(hp, read_off, (size_t)(n_reads + 1) * 4);
    if (nw) std::memcpy(hp + b_off, read_word, nw * 4);
    std::memcpy(hp + b_off + b_word, score, (size_t)n_reads * 4);
    std::memcpy(hp + b_off + b_word + b_r, best_off, (size_t)(n_reads + 1) * 8);
    char* dp = (char*)mat->io_in;
    HIP_TRY(hipMemcpy(dp, hp, up, hipMemcpyHostToDevice));
    const uint32_t *d_off = (const uint32_t*)dp, *d_word = (const uint32_t*)(dp + b_off);
    const int32_t* d_best = (const int32_t*)(dp + b_off + b_word);
    const unsigned long long* d_o64 = (const unsigned long long*)(dp + b_off + b_word + b_r);
    char* q = dp + up;
    uint32_t* d_cursor = (uint32_t*)q; q += b_r;
    uint8_t* d_plan = (uint8_t*)q; q += b_r1;
    uint32_t* d_list = (uint32_t*)q; q += b_r;
    int32_t* d_root = (int32_t*)q; q += b_r;
    uint32_t* d_slot = (uint32_t*)q; q += b_r;
    uint32_t* d_jobs = (uint32_t*)q; q += b_r;
    uint32_t* d_nodes = (uint32_t*)q;
    HIP_TRY(hipMemsetAsync(d_cursor, 0, (size_t)n_reads * 4, nullptr));
    // the stream of every read: the routing of a placement call with the walks off.  A read inside one genome window
    // lists its nodes on the window's stream when that is the window's candidate crown (real nodes only, a few
    // thousand of them whatever the read's root score: device_mat.hpp win_n) and smaller than the tree-wide stream of
    // its theta; a whole-tree window stream folds runs of nodes into pseudo-nodes: nothing to list there
    DevMAT dm = mat->dev;
    dm.wc_windows = 0;            // (no per-read window crowns: k_scores takes one stream per launch)
    uint32_t* tier_info = mat->lane[0].d_info + mat->lane[0].info_idx * TI_WORDS;
    uint32_t* tier_info_next = mat->lane[0].d_info + (mat->lane[0].info_idx ^ 1u) * TI_WORDS;
    uint32_t* blk_counts = mat->lane[0].d_info + 2 * TI_WORDS;
    HIP_TRY(launch_route(dm, d_off, d_word, n_reads, mat->use_crowns ? 3 : 0, 0u, WALK_JOB_EVENTS | (WALK_JOB_EVENTS << 16), WALK8_ROWS, WALK16_ROWS,
                         0xFFFFFFFFu, 0u, d_jobs, d_plan, d_root, blk_counts, tier_info, d_slot, tier_info_next, d_jobs, RouteDirect{}, nullptr));
    mat->lane[0].info_idx ^= 1u;
    HIP_TRY(launch_scatter(d_plan, d_slot, n_reads, blk_counts, tier_info, d_list, false, nullptr));
    HIP_TRY(hipMemcpy(mat->lane[0].h_info, tier_info, TI_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
    const uint32_t* info = mat->lane[0].h_info;
    const bool debug_plans = mat->tun.debug_plans;
    for (uint32_t id = 0; id < MAX_PLANS; id++) {
        const uint32_t count = info[TI_COUNT + id];
        if (!count) continue;
        const bool win = plan_class(id) == PLAN_WIN;
        if (win ? (plan_index(id) >= mat->wstreams.size() || mat->wstreams[plan_index(id)].ncnt != nullptr)
                : (plan_class(id) != PLAN_SWEEP || plan_index(id) >= mat->streams.size()))
            return set_error(WEPP_EDEVICE, "routing produced an invalid plan id");
        if (debug_plans) fprintf(stderr, "[best_nodes] %s %u: %u reads\n", win ? "window" : "stream", plan_index(id), count);
        HIP_TRY(launch_best_nodes(mat->dev, win ? mat->wstreams[plan_index(id)] : mat->streams[plan_index(id)], d_off, d_word,
                                  d_list + info[TI_OFF + id], count, d_best, d_o64, d_cursor, d_nodes, nullptr));
    }
    {
        hipError_t e = hipMemcpy(hp, d_nodes, total * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(e, "best-node kernel");
        std::vector<uint32_t> taken(n_reads);
        e = hipMemcpy(taken.data(), d_cursor, (size_t)n_reads * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(e, "best-node kernel");
        for (uint32_t r = 0; r < n_reads; r++)
            if (taken[r] != num_best[r])
                return set_error(WEPP_EINVAL, "read " + std::to_string(r) + ": score / num_best are not this read's placement on this tree (" +
                                              std::to_string(taken[r]) + " nodes attain the score, num_best says " + std::to_string(num_best[r]) + ")");
    }
    std::memcpy(best_nodes, hp, total * 4);
    // (the kernel fills a read's slice in the order its waves get there: ascending BFS index is the contract)
    for (uint32_t r = 0; r < n_reads; r++)
        if (num_best[r] > 1) std::sort(best_nodes + best_off[r], best_nodes + best_off[r + 1]);
    return WEPP_OK;
}

// diagnostic: the sweep stream every read of the handle's most recent placement call was routed to
extern "C" int wepp_mat_last_tiers(wepp_mat_t* mat, uint8_t* tiers, uint32_t n_reads) {
    if (!mat || !tiers) return set_error(WEPP_EINVAL, "null argument");
    if (n_reads != mat->last_n_reads || !mat->d_plan_of)
        return set_error(WEPP_EINVAL, "n_reads differs from the handle's last placement call");
    HIP_TRY(hipSetDevice(mat->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(tiers, mat->d_plan_of, n_reads, hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < n_reads; r++)       // the workspace holds plan ids (device_mat.hpp: plan_id)
        tiers[r] = (plan_class(tiers[r]) == PLAN_WIN || plan_class(tiers[r]) == PLAN_SEED) ? (uint8_t)(mat->dev.n_streams - 1) : (uint8_t)plan_index(tiers[r]);
    return WEPP_OK;
}

// diagnostic: the full plan id (class and stream) of every read of the handle's most recent placement call
extern "C" int wepp_mat_last_plans(wepp_mat_t* mat, uint8_t* plan_class_out, uint8_t* plan_stream_out, uint32_t n_reads) {
    if (!mat || !plan_class_out || !plan_stream_out) return set_error(WEPP_EINVAL, "null argument");
    if (n_reads != mat->last_n_reads || !mat->d_plan_of)
        return set_error(WEPP_EINVAL, "n_reads differs from the handle's last placement call");
    HIP_TRY(hipSetDevice(mat->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(plan_stream_out, mat->d_plan_of, n_reads, hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < n_reads; r++) {
        const uint32_t id = plan_stream_out[r];
        plan_class_out[r] = (uint8_t)plan_class(id);
        plan_stream_out[r] = (uint8_t)plan_index(id);
    }
    return WEPP_OK;
}

// diagnostic: the window crown (window, crown of the window) of every read of the last call that walked one
extern "C" int wepp_mat_last_crowns(wepp_mat_t* mat, uint8_t* window_out, uint8_t* crown_out, uint32_t n_reads) {
    if (!mat || !window_out || !crown_out) return set_error(WEPP_EINVAL, "null argument");
    if (n_reads != mat->last_n_reads || !mat->d_plan_of)
        return set_error(WEPP_EINVAL, "n_reads differs from the handle's last placement call");
    HIP_TRY(hipSetDevice(mat->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint32_t> sid(n_reads);
    std::vector<uint8_t> plan(n_reads);
    HIP_TRY(hipMemcpy(sid.data(), mat->d_wsid_of, (size_t)n_reads * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(plan.data(), mat->d_plan_of, n_reads, hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < n_reads; r++) {
        const bool in_arena = plan_class(plan[r]) != PLAN_WIN && plan_class(plan[r]) != PLAN_SWEEP && plan_index(plan[r]) == WC_SLOT;
        window_out[r] = in_arena ? (uint8_t)(sid[r] / WC_MAX) : (uint8_t)255;
        crown_out[r] = in_arena ? (uint8_t)(sid[r] % WC_MAX) : (uint8_t)255;
    }
    return WEPP_OK;
}

// diagnostic: how the most recent placement call placed its reads
extern "C" int wepp_mat_last_walk(wepp_mat_t* mat, uint64_t* reads_walked, uint64_t* walk_iterations) {
    if (!mat) return set_error(WEPP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(mat->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long it = 0;
    {
        std::vector<unsigned long long> slots(WALK_COUNTERS);
        HIP_TRY(hipMemcpy(slots.data(), mat->d_work, WALK_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (unsigned long long v : slots) it += v;
    }
    if (reads_walked) *reads_walked = mat->last_walk_reads;
    if (walk_iterations) *walk_iterations = it;
    if (mat->tun.walk_debug) {
        std::vector<unsigned long long> slots(WALK_COUNTERS);
        HIP_TRY(hipMemcpy(slots.data(), mat->d_work, WALK_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long a = 0, b = 0;
        for (uint32_t i = 0; i < WALK_COUNTERS / 2; i++) { a += slots[i]; b += slots[WALK_COUNTERS / 2 + i]; }
        fprintf(stderr, "[walk] wave-iterations since reset: plain %llu chunked %llu\n", a, b);
        // -DWEPP_WALK_STATS builds: slots 0..6 / 16..22 = wave cycles by phase (decode, stage, start state, walk, write), waves, iterations
        for (int c = 0; c < 2; c++) {
            const unsigned long long* v = slots.data() + c * 16;
            if (v[5] && v[5] < (1ull << 40))
                fprintf(stderr, "[walk stats] %s: waves %llu iterations/wave %.1f cycles/wave: decode %.0f stage %.0f start-state %.0f walk %.0f write %.0f\n",
                        c ? "chunked" : "plain", v[5], (double)v[6] / v[5], (double)v[0] / v[5], (double)v[1] / v[5], (double)v[2] / v[5],
                        (double)v[3] / v[5], (double)v[4] / v[5]);
            if (v[5] && v[5] < (1ull << 40))
                fprintf(stderr, "[walk stats] %s: lanes with work %llu  lane-iterations %llu (%.1f per lane, %.1f%% of wave-iterations x 64)  node events %llu  "
                        "table bytes %llu  exact range queries %llu\n", c ? "chunked" : "plain", v[11], v[7], (double)v[7] / std::max(1ull, v[11]),
                        100.0 * v[7] / std::max(1.0, 64.0 * v[6]), v[8], v[9], v[10]);
        }
    }
    return WEPP_OK;
}

// diagnostic: the whole-genome samples the handle seeded since the last timing reset, the chunks of the whole-tree
// stream they evaluated and the chunks they could have evaluated (samples x chunks of the tree)
extern "C" int wepp_mat_last_seeds(wepp_mat_t* mat, uint64_t* samples, uint64_t* chunks_evaluated, uint64_t* chunks_total,
                                   uint64_t* most_per_sample, uint64_t* histogram8) {
    if (!mat) return set_error(WEPP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(mat->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long v[12] = {};
    HIP_TRY(hipMemcpy(v, mat->d_work + 2 * WALK_COUNTERS, sizeof(v), hipMemcpyDeviceToHost));
    if (samples) *samples = v[0];
    if (chunks_evaluated) *chunks_evaluated = v[1];
    if (chunks_total) *chunks_total = v[2];
    if (most_per_sample) *most_per_sample = v[3];
    if (histogram8)
        for (int i = 0; i < 8; i++) histogram8[i] = v[4 + i];
    return WEPP_OK;
}
