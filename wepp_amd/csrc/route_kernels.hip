// route_kernels.hip -- routing of a batch: which stream / plan every read takes (k_route), the read lists per plan
// (k_scatter) and the sort keys of the lists that are position-sorted.
//
// Work skipping (exact): score(n) >= base(n) - |S| for every node, and the best score is <=
// the root's score, so a read with theta = score(root) + |S| only needs the "crown" of nodes with base <= theta
// (plus ancestors); a read confined to a genome window needs the window crown its ROOT score admits.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_mat.hpp"
#include "place_dev.hpp"

namespace wepp {

#ifdef WEPP_ROUTE_STATS   // (profiling build, tools/build_variant.sh: clock ticks of block 0's first wave by phase of k_route)
__device__ unsigned long long g_route_stats[16];
#define ROUTE_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_route_stats[i] += now_ - last_; last_ = now_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#define ROUTE_STAMP_NW(i) do { __builtin_amdgcn_sched_barrier(0); { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_route_stats[i] += now_ - last_; last_ = now_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ROUTE_STAMP(i)
#define ROUTE_STAMP_NW(i)
#endif

// -----------------------------------------------------------------------------
// k_route: theta(read) = score(root) + |S| -> index of the smallest stream whose
// tau covers it.  Per-(block, tier) counts go to blk_counts, per-tier totals and
// the largest read of each tier to tier_info.
// -----------------------------------------------------------------------------
__global__ __launch_bounds__(ROUTE_THREADS) void k_route(DevMAT m, const uint32_t* __restrict__ read_off,
                                                          const uint32_t* __restrict__ read_word, uint32_t n_reads,
                                                          int use_crowns, uint32_t walk_max_events, uint32_t job_events,
                                                          uint32_t stack8, uint32_t stack16,
                                                          uint32_t seed_min_hard, uint32_t seed_min_nodes,
                                                          uint32_t* __restrict__ job_n, uint8_t* __restrict__ tier_of,
                                                          int32_t* __restrict__ root_score,
                                                          uint32_t* __restrict__ blk_counts,
                                                          uint32_t* __restrict__ tier_info,
                                                          uint32_t* __restrict__ slot_in_blk,
                                                          uint32_t* __restrict__ tier_info_next,
                                                          uint32_t* __restrict__ wsid, RouteDirect direct) {
#ifdef WEPP_ROUTE_STATS
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
    __shared__ uint32_t cnt[MAX_PLANS], mx[MAX_PLANS], jobs_of[2 * MAX_STREAMS], open_of[4], events_of[2], resolved_of[1];
    // the routing table of every stream a read can walk -- the window crowns of every genome window, then the tree-wide
    // streams (wc_info[tw_base + t]) -- in LDS: bound, size, where its position index starts in the arena, and its
    // stream-wide aggregate (what a read without events in it is placed by).  Read from the 80-byte records of the
    // device table, every one of these was a load of its own in the read's chain of dependent loads.
    constexpr uint32_t RT_MAX = MAX_WINDOWS * WC_MAX + MAX_STREAMS;
    __shared__ int32_t wc_tau[RT_MAX], rt_wbase[RT_MAX];
    __shared__ uint32_t wc_n[RT_MAX], rt_head[RT_MAX], rt_nest[RT_MAX], rt_wbfs[RT_MAX], rt_wcnt[RT_MAX];
    for (uint32_t i = threadIdx.x; i < m.tw_base + m.n_streams && i < RT_MAX; i += blockDim.x) {
        const WcInfo q = m.wc_info[i];
        wc_tau[i] = q.tau; wc_n[i] = q.n; rt_head[i] = q.head_off; rt_nest[i] = q.nest_off;
        rt_wbase[i] = q.whole.base; rt_wbfs[i] = q.whole_bfs; rt_wcnt[i] = (q.whole.cnt << 1) | (q.whole.hu ? 1u : 0u);
    }
    // the two scans of a read -- the first tree-wide stream whose bound covers theta, the first crown of its window whose
    // bound covers its root score -- as table lookups: the bounds are small integers (tree-wide: the flattener's theta
    // levels; window crowns: base(root) + 0 .. WC_MAX_DTAU, then "any"), so theta and root score - base(root) index
    // them.  (tab_ok == 0 -- a bound the tables do not reach -- leaves the scans in place.)
    constexpr uint32_t TT_MAX = 64, WT_D = 8;
    __shared__ uint8_t t_tab[TT_MAX], sid_tab[MAX_WINDOWS * WT_D];
    __shared__ uint32_t tab_ok[1];
    if (threadIdx.x == 0) tab_ok[0] = (m.n_streams >= 2 && m.tau[m.n_streams - 2] < (int32_t)TT_MAX && m.tau[0] >= 0) ? 1u : 0u;
    __syncthreads();
    if (threadIdx.x < TT_MAX) {
        uint32_t t = m.n_streams - 1;
        for (uint32_t i = 0; i + 1 < m.n_streams; i++)
            if ((int32_t)threadIdx.x <= m.tau[i]) { t = i; break; }
        t_tab[threadIdx.x] = (uint8_t)t;
    }
    if (threadIdx.x < m.wc_windows * WT_D && threadIdx.x < MAX_WINDOWS * WT_D) {
        const uint32_t wi = threadIdx.x / WT_D, d = threadIdx.x % WT_D;
        // d < WT_D - 1: root score = base(root) + d exactly; d == WT_D - 1: that or more -- only a crown without a
        // bound serves those, which the table can say as long as every bounded crown's bound is below base(root) + d
        // (from the LDS copies of the bounds: the barrier above)
        uint32_t pick = 0xFFu;
        for (uint32_t i = 0; i < WC_MAX; i++) {
            const int32_t tau = wc_tau[wi * WC_MAX + i];
            if (!wc_n[wi * WC_MAX + i]) break;
            if (tau != 0x7FFFFFFF && (tau < m.root_base || tau >= m.root_base + (int32_t)(WT_D - 1))) atomicAnd(&tab_ok[0], 0u);
            if (pick == 0xFFu && m.root_base + (int32_t)d <= tau) pick = i;
        }
        sid_tab[threadIdx.x] = (uint8_t)pick;
    }
    __shared__ uint32_t wl_count[2], wl_base[2], cj_count[2], cl_count[2], cj_base[2], cl_base[2], c_ok[2], ww_count[2], ww_base[2], w16_to_wave[1], ww_cand[2];
    if (threadIdx.x < 2) { cj_count[threadIdx.x] = 0; cl_count[threadIdx.x] = 0; }
    if (threadIdx.x == 0) ww_count[0] = ww_count[1] = ww_cand[0] = ww_cand[1] = 0;
    if (threadIdx.x == 0) resolved_of[0] = 0;
    if (threadIdx.x < 2) wl_count[threadIdx.x] = 0;
    // the counters of the NEXT call (the handle alternates between two sets) are cleared here: no memset
    // (two fill kernels, ~10 us) in front of every call
    if (blockIdx.x == 0 && threadIdx.x < TI_WORDS) tier_info_next[threadIdx.x] = 0;
    if (threadIdx.x < MAX_PLANS) { cnt[threadIdx.x] = 0; mx[threadIdx.x] = 0; }
    if (threadIdx.x < 4) open_of[threadIdx.x] = 0;
    if (threadIdx.x < 2) events_of[threadIdx.x] = 0;
    if (threadIdx.x < 2 * MAX_STREAMS) jobs_of[threadIdx.x] = 0;
    __syncthreads();
    ROUTE_STAMP(0);      // prologue
    const uint32_t per = (n_reads + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per, hi = min(n_reads, lo + per);
    const IxHead* __restrict__ ar_head = m.walks[WC_SLOT].ix_head;      // the walk arena: every stream's index is a slice of it
    const uint8_t* __restrict__ ar_nest = m.walks[WC_SLOT].ix_nest;
    const uint32_t root_w0 = m.node_woff[0], root_w1 = m.node_woff[1];
    const bool use_tabs = tab_ok[0] != 0;
    // four reads per thread and round: their offsets, then their first two words, are requested together
    // (one read after the other, every read cost its thread three memory round trips in a row)
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // (every thread of the block runs the same number of rounds: the walkers' list slots are reserved per block and round)
    for (uint32_t rbase = lo; rbase < hi; rbase += 4 * blockDim.x) {
      const uint32_t r0 = rbase + threadIdx.x;
      uint32_t so4[4], k4[4], fw[4][2];
      uint32_t app4[4] = {0, 0, 0, 0}, aslot4[4] = {0, 0, 0, 0};     // plain walk class + 1 and slot among the block's walkers of the class
      uint32_t cj4[4] = {0, 0, 0, 0}, cjs4[4] = {0, 0, 0, 0}, cls4[4] = {0, 0, 0, 0}, cnj4[4] = {0, 0, 0, 0};   // chunked class + 1, job / list slot in the block, jobs
      uint32_t ww4[4] = {0, 0, 0, 0};                                   // slot + 1 among the block's wave-per-read reads
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
          const uint32_t r = r0 + u * blockDim.x;
          so4[u] = r < hi ? read_off[r] : 0u;
          k4[u] = r < hi ? read_off[r + 1] - so4[u] : 0u;
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; u++)
#pragma unroll
          for (uint32_t j = 0; j < 2; j++) fw[u][j] = k4[u] > j ? read_word[so4[u] + j] : 0u;
      ROUTE_STAMP(1);    // offsets and first words
      // ---- phase A, the four reads one after the other: root score, tree-wide stream, window crown -- words and LDS
      // tables only -- and the loads the classification below starts with (the list heads and nesting depths of the
      // read's first two positions in the stream it will most likely walk), requested for all four before any is used:
      // read by read, every read cost its thread a memory round trip of its own, one after the other
      int c4[4] = {0, 0, 0, 0};
      uint32_t nh4[4] = {0, 0, 0, 0}, t4[4] = {0, 0, 0, 0}, sid4[4] = {NONE, NONE, NONE, NONE}, wi4[4] = {0, 0, 0, 0}, inwin4[4] = {0, 0, 0, 0};
      uint32_t pre_nest[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, pre_o0[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, pre_o1[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
        const uint32_t r = r0 + u * blockDim.x;
        const uint32_t k = k4[u], so = so4[u];
        if (r < hi) {
            int c = 0;
            auto count = [&](uint32_t sw) { if (!rw_missing(sw)) c += ((rw_mut(sw) & rw_ref(sw)) == 0) ? 1 : 0; };
            if (k > 0) count(fw[u][0]);
            if (k > 1) count(fw[u][1]);
            for (uint32_t j = 2; j < k; j++) count(read_word[so + j]);
            nh4[u] = (uint32_t)c;              // entries whose alleles exclude their reference base (seed_kernels.hip)
            for (uint32_t w = root_w0; w < root_w1; w++) {   // the root's own mutations
                const uint32_t tw = m.words[w];
                uint32_t sw = NONE;
                if (k <= 2) {
                    if (k > 0 && w_pos(fw[u][0]) == w_pos(tw)) sw = fw[u][0];
                    if (k > 1 && w_pos(fw[u][1]) == w_pos(tw)) sw = fw[u][1];
                } else sw = find_entry(read_word, so, k, w_pos(tw));
                if (sw != NONE) c += enter_delta(tw, sw);
            }
            c4[u] = c;
            ROUTE_STAMP_NW(2);  // entry counts, the root's mutations
            const int theta = m.root_base + c + (int)k;
            uint32_t t = m.n_streams - 1;
            if (use_crowns) {
                if (use_tabs) t = theta < 0 ? 0u : theta < (int)TT_MAX ? (uint32_t)t_tab[theta] : m.n_streams - 1;
                else
                    for (uint32_t i = 0; i + 1 < m.n_streams; i++)
                        if (theta <= m.tau[i]) { t = i; break; }
            }
            t4[u] = t;
            ROUTE_STAMP_NW(3);  // tree-wide stream
            // a read inside one genome window: the window crown its ROOT score admits (flatmat.hpp: wcrowns) holds
            // every node that can win or tie -- far fewer than the tree-wide crown of theta = root score + |S|
            if (use_crowns && k > 0) {
                const uint32_t p_lo = w_pos(fw[u][0]), p_hi = w_pos(k > 2 ? read_word[so + k - 1] : k == 2 ? fw[u][1] : fw[u][0]);
                const uint32_t wi = p_lo / WIN_STRIDE;
                wi4[u] = wi;
                inwin4[u] = p_hi < wi * WIN_STRIDE + WIN_SIZE ? 1u : 0u;      // all listed positions inside genome window wi
                if (inwin4[u] && wi < m.wc_windows) {
                    const int rs = m.root_base + c;
                    if (use_tabs) {
                        const uint32_t i = sid_tab[wi * WT_D + (uint32_t)min(max(c, 0), (int)WT_D - 1)];
                        if (i != 0xFFu && wc_n[wi * WC_MAX + i] < wc_n[m.tw_base + t]) sid4[u] = wi * WC_MAX + i;
                    } else
                        for (uint32_t i = 0; i < WC_MAX; i++) {
                            const uint32_t qn = wc_n[wi * WC_MAX + i];
                            if (!qn) break;
                            if (rs <= wc_tau[wi * WC_MAX + i]) { if (qn < wc_n[m.tw_base + t]) sid4[u] = wi * WC_MAX + i; break; }
                        }
                }
            }
            ROUTE_STAMP_NW(4);  // window crown
            if (walk_max_events && k <= WALK16_K) {
                const uint32_t tab = sid4[u] != NONE ? sid4[u] : m.tw_base + t;
#pragma unroll
                for (uint32_t j = 0; j < 2; j++) {
                    const uint32_t p = w_pos(fw[u][j]);
                    if (j < k && p <= m.max_pos) {
                        pre_nest[u][j] = (uint32_t)ar_nest[rt_nest[tab] + p];
                        pre_o0[u][j] = ar_head[rt_head[tab] + p].off;
                        pre_o1[u][j] = ar_head[rt_head[tab] + p + 1].off;
                    }
                }
            }
        }
      }
      ROUTE_STAMP(5);    // the first loads of the classification
      // ---- phase B: class and plan of every read ----
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
        const uint32_t r = r0 + u * blockDim.x;
        const bool valid = r < hi;
        uint32_t t_id = 0, append = 0;         // plan id; 1 / 2: joins the list of the plain walk class 8 / 16 (direct mode)
        bool resolved = false;
        const uint32_t k = k4[u];
        if (valid) {
        const uint32_t so = so4[u];
        const int c = c4[u];
        const uint32_t n_hard = nh4[u];
        uint32_t t = t4[u];
        const uint32_t sid = sid4[u], wi = wi4[u];
        const bool in_win = inwin4[u] != 0;
        const uint32_t pre_tab = sid != NONE ? sid : m.tw_base + t;     // the stream phase A asked about
        // how the read is placed (device_mat.hpp): by walking its own events when it lists few positions and
        // the intervals it can hold open at once fit the walk's stack, by a sweep of the stream otherwise
        // A walk runs a read's events one after the other: reads with many events in their stream (a
        // frequently mutated position) are left to the sweeps, whose cost does not depend on it.
        uint32_t cls = PLAN_SWEEP;
        // how a read with at most WALK16_K entries would walk the stream of routing-table entry `tab` (its position index
        // is a slice of the arena): plain, cut into jobs, or not at all (more open intervals than a walk's stack holds)
        auto classify = [&](uint32_t tab, uint32_t& nj_out, uint32_t& open_out, uint32_t& ev_out) -> uint32_t {
            const IxHead* ix_head = ar_head + rt_head[tab];
            const uint8_t* ix_nest = ar_nest + rt_nest[tab];
            uint32_t open_max = 0, events = 0, longest = 0;
            for (uint32_t j = 0; j < k; j++) {
                const uint32_t p = w_pos(j == 0 ? fw[u][0] : j == 1 ? fw[u][1] : read_word[so + j]);
                if (p <= m.max_pos) {
                    uint32_t nest, o0, o1;
                    if (j < 2 && tab == pre_tab) { nest = j ? pre_nest[u][1] : pre_nest[u][0]; o0 = j ? pre_o0[u][1] : pre_o0[u][0]; o1 = j ? pre_o1[u][1] : pre_o1[u][0]; }
                    else { nest = (uint32_t)ix_nest[p]; o0 = ix_head[p].off; o1 = ix_head[p + 1].off; }
                    open_max += nest;
                    const uint32_t len = o1 - o0 - 1u;      // (every list ends in a sentinel)
                    events += len;
                    longest = max(longest, len);
                }
            }
            open_out = open_max;
            ev_out = events;
            nj_out = 0;
            // (stack8 <= WALK8_STACK, stack16 <= WALK16_STACK: the stack rows the walks' workgroups get; the few
            // reads that could hold more intervals open are left to the sweeps)
            if (events <= walk_max_events) {
                if (k <= WALK8_K && open_max <= stack8) return PLAN_WALK8;
                if (open_max <= stack16) return PLAN_WALK16;
                return PLAN_SWEEP;
            }
            if (open_max > stack16) return PLAN_SWEEP;
            // many events: jobs of about `job_events`, cut at quantiles of the longest list
            const uint32_t small = (k <= WALK8_K && open_max <= stack8) ? 1u : 0u;
            const uint32_t je = small ? (job_events & 0xFFFFu) : (job_events >> 16);   // (per class, capi.cpp)
            nj_out = min((events + je - 1) / je, longest);
            return small ? PLAN_WALKC8 : PLAN_WALKC16;
        };
        if (walk_max_events && k <= WALK16_K) {
            uint32_t nj = 0, open_max = 0, events = 0;
            if (sid != NONE) {
                cls = classify(sid, nj, open_max, events);
                if (cls != PLAN_SWEEP) { wsid[r] = sid; t = WC_SLOT; }
            }
            if (cls == PLAN_SWEEP) cls = classify(m.tw_base + t, nj, open_max, events);
            if (cls == PLAN_WALK8 || cls == PLAN_WALK16) {
                // (a handle that cuts such reads into jobs lets up to 16 events walk plainly: they stay counted as what a
                // wave each would have to take, or the host's choice for the next call would swing back and forth)
                if (events > direct.cand_min) atomicAdd(&ww_cand[events > 64u ? 1 : 0], 1u);
                // the deepest stack a walk of the class can need in this call: its kernel's LDS request
                if (open_max > open_of[cls]) atomicMax(&open_of[cls], open_max);
                // every stream is a slice of the walk arena: the read walks the slice wsid names (a window crown's, set
                // above, or its tree-wide stream's)
                if (t != WC_SLOT) wsid[r] = m.tw_base + t;
                if (direct.wlist[0]) {
                    if (events == 0) {
                        // none of the read's positions is mutated in its stream: every node scores base + c, and the
                        // stream-wide aggregate is the answer (what k_walk finds without a single range query) -- the
                        // root is in every stream and always competes, so the aggregate never loses to the bound
                        const uint32_t at = t == WC_SLOT ? sid : m.tw_base + t;
                        if (direct.best_bfs_j) direct.best_bfs_j[r] = rt_wbfs[at];
                        if (direct.score) direct.score[r] = rt_wbase[at] + c;
                        if (direct.num_best) direct.num_best[r] = rt_wcnt[at] >> 1;
                        if (direct.flags) direct.flags[r] = (rt_wcnt[at] & 1u) ? WEPP_FLAG_HAS_UNIQUE_DEV : 0u;
                        resolved = true;
                    } else append = cls == PLAN_WALK8 ? 1u : 2u;
                }
            } else if (cls == PLAN_WALKC8 || cls == PLAN_WALKC16) {
                const uint32_t small = cls == PLAN_WALKC8 ? 1u : 0u;
                if (t != WC_SLOT) wsid[r] = m.tw_base + t;
                // many events, but few enough for a wave per 64 of them to hold them all (lane = list entry, wave_kernels.hip):
                // no jobs; the read keeps its chunked class in the diagnostics.  (<= 64: a wave of its own, listed from the
                // front; more: the four waves of a block, listed from the back.)  Up to direct.ww_max_* of them per routing
                // block and round -- all or none in practice: the all-pairs pass is the cheaper way for the one read in a
                // thousand of a sequencing run, the jobs for a batch FULL of such reads (a star-like tree: 29 000 of a
                // million -- 0.60 ms a step by waves, 0.36 by jobs), and a call that needs both pays for both; the host
                // picks from the counts of the handle's previous call (capi.cpp)
                bool by_wave = false;
                if (direct.wwlist && events <= WAVE_WALK_MAX_EVENTS) {
                    const uint32_t big = events > 64u ? 1u : 0u;
                    atomicAdd(&ww_cand[big], 1u);          // (how many there are, whichever way they go: the host's hint for the next call)
                    const uint32_t slot = atomicAdd(&ww_count[big], 1u);          // (the reservation clamps the count)
                    if (slot < (big ? direct.ww_max_big : direct.ww_max_small)) {
                        by_wave = true;
                        job_n[r] = 0;
                        ww4[u] = ((slot + 1u) << 1) | big;
                        nj = 0;
                        events = 0;
                    }
                }
                if (!by_wave) job_n[r] = nj;
                if (nj && direct.jobs[0]) {      // (few reads: an LDS atomic each)
                    cj4[u] = small ? 1u : 2u;
                    cnj4[u] = nj;
                    cjs4[u] = atomicAdd(&cj_count[small ? 0 : 1], nj);
                    cls4[u] = atomicAdd(&cl_count[small ? 0 : 1], 1u);
                }
                atomicAdd(&events_of[small ? 0 : 1], events);
                atomicAdd(&jobs_of[(small ? 0u : MAX_STREAMS) + t], nj);
                if (open_max > open_of[small ? 2 : 3]) atomicMax(&open_of[small ? 2 : 3], open_max);
            }
        }
        // (use_crowns & 2 -- wepp_best_nodes, which lists nodes and so takes streams of real nodes only: every read
        // inside a window whose stream is the window's candidate crown takes it when it is the smaller one)
        if (cls == PLAN_SWEEP && in_win && wi < m.n_windows &&
            ((use_crowns & 2) ? m.win_n[wi] < wc_n[m.tw_base + t]
                              : k > WIN_MIN_ENTRIES ? (m.win_n[wi] < wc_n[m.tw_base + t] || t + 1 == m.n_streams) : (sid == NONE && t + 1 == m.n_streams))) {
            // many entries, all inside one genome window: a tile of such reads sweeps the window's stream -- the window's
            // candidates (a crown of a few thousand nodes, whatever the root score) or, for the reads no crown serves,
            // the whole tree as the window sees it
            cls = PLAN_WIN;
            t = wi;
        } else if (cls == PLAN_SWEEP && sid != NONE) {
            // it cannot walk (more than WALK16_K entries or too deep a stack): waves of its own sweep its window crown (k_sweep_arena)
            wsid[r] = sid;
            t = WC_SLOT;
        } else if (cls == PLAN_SWEEP && n_hard >= seed_min_hard && m.seed_chunks && k <= SEED_MAX_ENTRIES && wc_n[m.tw_base + t] >= seed_min_nodes) {
            // a whole-genome sample (no window holds it, too many entries to walk): the chunk signatures rule out nearly
            // all of the tree for it, whatever its tree-wide bound admits (seed_kernels.hip)
            cls = PLAN_SEED;
            t = 0;
        }
        t_id = plan_id(cls, t);
        tier_of[r] = (uint8_t)t_id;
        // (the root always competes: an upper bound of the best score.  Stored here, behind the loads of both phases: a
        // store in front of a load makes the wait for the load a wait for the store's acknowledgement too)
        root_score[r] = m.root_base + c;
        if (!(resolved || append) && k > mx[t_id]) atomicMax(&mx[t_id], k);
        }   // valid
        // ---- counters, one LDS atomic per wave and distinct value instead of one per read (all lanes take part) ----
        // position among this block's reads of the plan (k_scatter): the reads of a wave that share a plan take
        // consecutive slots
        // (the reads k_route has placed or listed for the blind walks itself -- nearly all of a sequencing run -- take no
        // slot: nobody reads their plan's list, k_scatter skips them, and the loop below runs once per plan AMONG THE
        // OTHERS of the wave, i.e. not at all for most waves)
        {
            const bool listed = valid && !(resolved || append);
            unsigned long long todo = __ballot(listed);
            uint32_t slot = 0;
            while (todo) {
                const int first = __builtin_ctzll(todo);
                const uint32_t tt = (uint32_t)__builtin_amdgcn_readlane((int)t_id, first);
                const unsigned long long peers = __ballot(listed && t_id == tt);
                uint32_t base = 0;
                if ((int)lane == first) base = atomicAdd(&cnt[tt], (uint32_t)__popcll(peers));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                if (listed && t_id == tt) slot = base + (uint32_t)__popcll(peers & lt_mask);
                todo &= ~peers;
            }
            if (listed) slot_in_blk[r] = slot;
        }
        {
            const unsigned long long rm = __ballot(resolved);
            if (rm && lane == (uint32_t)__builtin_ctzll(rm)) atomicAdd(&resolved_of[0], (uint32_t)__popcll(rm));
        }
        // the plain walkers' lists: a slot among the block's walkers of the class now, the block's range in the list below
        if (direct.wlist[0]) {
#pragma unroll
            for (uint32_t cc = 0; cc < 2; cc++) {
                const unsigned long long mk = __ballot(append == cc + 1);
                if (!mk) continue;
                const int first = __builtin_ctzll(mk);
                uint32_t base = 0;
                if ((int)lane == first) base = atomicAdd(&wl_count[cc], (uint32_t)__popcll(mk));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                if (append == cc + 1) { app4[u] = cc + 1; aslot4[u] = base + (uint32_t)__popcll(mk & lt_mask); }
            }
        }
        ROUTE_STAMP(6);  // class, plan, counters and slots
      }
      // ... one global atomic per block, round and class reserves the block's range of the class's list (a global atomic
      // per wave, ~60 K of them on one address per million reads, made k_route six times slower)
      if (direct.wlist[0]) {
          __syncthreads();
          if (threadIdx.x < 2) {
              const uint32_t cc = threadIdx.x;
              // a handful of plain walkers with 9 - 16 entries in the block (a sequencing run: one in a million reads): they
              // join the reads with many events (a wave each, k_walk_wave) -- a launch of their own starts when the walks
              // of the other class end, ~15 us at the end of every call, for a read or two
              const bool to_wave = cc == 1 && direct.wwlist && wl_count[1] && wl_count[1] <= WALK16_TO_WAVE_MAX;
              if (cc == 1) w16_to_wave[0] = to_wave ? 1u : 0u;
              if (to_wave) {
                  wl_base[1] = atomicAdd(&tier_info[TI_WWCUR], wl_count[1]);
                  atomicAdd(&tier_info[TI_W16WAVE], wl_count[1]);
              } else
                  wl_base[cc] = wl_count[cc] ? atomicAdd(&tier_info[TI_WCUR + cc], wl_count[cc]) : 0u;
              wl_count[cc] = 0;
              // the chunked classes' job tables and lists: the block's ranges, or -- a table outgrown -- the class is
              // flagged and left to the host's planned launch (the blind kernels leave at once)
              c_ok[cc] = 0;
              if (cl_count[cc] && direct.jobs[0]) {
                  cj_base[cc] = atomicAdd(&tier_info[TI_JCUR + cc], cj_count[cc]);
                  cl_base[cc] = atomicAdd(&tier_info[TI_CCUR + cc], cl_count[cc]);
                  if (cj_base[cc] + cj_count[cc] > BLIND_JOB_CAP || cl_base[cc] + cl_count[cc] > BLIND_CHUNKED_READS) tier_info[TI_JOVER + cc] = 1u;
                  else c_ok[cc] = 1;
              }
              cj_count[cc] = 0;
              cl_count[cc] = 0;
              if (cc == 0) {
                  for (uint32_t b = 0; b < 2; b++) {
                      const uint32_t n = min(ww_count[b], b ? direct.ww_max_big : direct.ww_max_small);
                      ww_base[b] = n ? atomicAdd(&tier_info[TI_WWCUR + b], n) : 0u;
                      ww_count[b] = 0;
                  }
              }
          }
          __syncthreads();
#pragma unroll
          for (uint32_t u = 0; u < 4; u++) {
              const uint32_t r = r0 + u * blockDim.x;
              if (app4[u]) (app4[u] == 1 ? direct.wlist[0] : w16_to_wave[0] ? direct.wwlist : direct.wlist[1])[wl_base[app4[u] - 1] + aslot4[u]] = r;     // (no indexing of the argument's arrays: that would copy them to scratch)
              if (ww4[u]) {
                  const uint32_t big = ww4[u] & 1u, at = ww_base[big] + (ww4[u] >> 1) - 1u;
                  direct.wwlist[big ? n_reads - 1u - at : at] = r;
              }
              if (cj4[u] && c_ok[cj4[u] - 1]) {
                  const uint32_t cc = cj4[u] - 1, j0 = cj_base[cc] + cjs4[u];
                  (cc ? direct.clist[1] : direct.clist[0])[cl_base[cc] + cls4[u]] = r;
                  direct.job_first[r] = j0;
                  uint32_t* jt = cc ? direct.jobs[1] : direct.jobs[0];
                  for (uint32_t c2 = 0; c2 < cnj4[u]; c2++) jt[j0 + c2] = r;
              }
          }
          __syncthreads();        // (c_ok and the bases are rewritten in the next round)
      }
      ROUTE_STAMP(7);    // the lists: barrier, reservation, writes
    }
    __syncthreads();
    if (threadIdx.x < MAX_PLANS) {
        blk_counts[blockIdx.x * MAX_PLANS + threadIdx.x] = cnt[threadIdx.x];
        if (cnt[threadIdx.x]) {
            atomicAdd(&tier_info[TI_COUNT + threadIdx.x], cnt[threadIdx.x]);
            atomicMax(&tier_info[TI_MAXK + threadIdx.x], mx[threadIdx.x]);
        }
    }
    if (threadIdx.x < 2 * MAX_STREAMS && jobs_of[threadIdx.x]) atomicAdd(&tier_info[TI_JOBS + threadIdx.x], jobs_of[threadIdx.x]);
    if (threadIdx.x < 4 && open_of[threadIdx.x]) atomicMax(&tier_info[TI_OPEN + threadIdx.x], open_of[threadIdx.x]);
    if (threadIdx.x < 2 && events_of[threadIdx.x]) atomicAdd(&tier_info[TI_EVENTS + threadIdx.x], (events_of[threadIdx.x] + 63) >> 6);
    if (threadIdx.x < 2 && ww_cand[threadIdx.x]) atomicAdd(&tier_info[TI_WWCAND + threadIdx.x], ww_cand[threadIdx.x]);
    if (threadIdx.x == 0 && resolved_of[0]) {
        atomicAdd(&tier_info[TI_RESOLVED], resolved_of[0]);
        // what a resolved read asked memory for: its offsets and words, a list head per entry, the aggregate, the result
        if (direct.work_counter) atomicAdd(direct.work_counter + WALK_COUNTERS + (blockIdx.x & (WALK_COUNTERS - 1)), 48ull * resolved_of[0]);
    }
    ROUTE_STAMP(8);      // epilogue
#ifdef WEPP_ROUTE_STATS
    if (blockIdx.x == 0 && threadIdx.x == 0) g_route_stats[9] += 1;
#endif
}
#ifdef WEPP_ROUTE_STATS
extern "C" int wepp_debug_route_stats(unsigned long long* out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_route_stats), sizeof(g_route_stats));
}
#endif

// entry indices of one stream's slice of the walk arena made absolute (capi.cpp: the tree-wide streams are copied from
// the image into their slices as they are)
__global__ void k_rebase_index(IxHead* __restrict__ heads, uint32_t n_heads, IxEnt* __restrict__ ents, uint32_t n_ents, uint32_t ent_off) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_heads) heads[i].off += ent_off;
    if (i < n_ents && ents[i].up != IX_NONE) ents[i].up += ent_off;
}

// -----------------------------------------------------------------------------
// k_scatter: list[] = read indices grouped by tier (same block decomposition as
// k_route; a block's reads of one tier occupy a contiguous range).
// -----------------------------------------------------------------------------
__global__ __launch_bounds__(ROUTE_THREADS) void k_scatter(const uint8_t* __restrict__ tier_of,
                                                            const uint32_t* __restrict__ slot_in_blk, uint32_t n_reads,
                                                            const uint32_t* __restrict__ blk_counts,
                                                            uint32_t* __restrict__ tier_info,
                                                            uint32_t* __restrict__ list, int skip_plain_walks) {
    __shared__ uint32_t base[MAX_PLANS], before[MAX_PLANS];
    if (threadIdx.x < MAX_PLANS) before[threadIdx.x] = 0;
    __syncthreads();
    // reads of every tier in the blocks before this one: thread = (tier, one earlier block in 64), so that a
    // wave's sixteen-lane groups read whole 64-byte rows and only four lanes of a wave add to the same LDS
    // word (one thread per row with sixteen adds each serialised 64 lanes on every word)
    {
        const uint32_t t = threadIdx.x & (MAX_PLANS - 1), c0 = threadIdx.x / MAX_PLANS;
        uint32_t acc = 0;
        for (uint32_t b = c0; b < blockIdx.x; b += blockDim.x / MAX_PLANS) acc += blk_counts[b * MAX_PLANS + t];
        if (acc) atomicAdd(&before[t], acc);
    }
    __syncthreads();
    if (threadIdx.x < MAX_PLANS) {
        const uint32_t t = threadIdx.x;
        uint32_t off = 0;                       // start of plan t in the list
        for (uint32_t i = 0; i < t; i++) off += tier_info[TI_COUNT + i];
        base[t] = off + before[t];
        if (blockIdx.x == 0) {
            tier_info[TI_OFF + t] = off;
            if (t == MAX_PLANS - 1) tier_info[TI_OFF + MAX_PLANS] = off + tier_info[TI_COUNT + t];
        }
    }
    __syncthreads();
    const uint32_t per = (n_reads + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per, hi = min(n_reads, lo + per);
    // no atomics here.  Four reads per thread and round: their loads are issued together (a thread's
    // reads used to cost it one memory round trip after the other: 28 us per 1 M reads)
    for (uint32_t r0 = lo + threadIdx.x; r0 < hi; r0 += 4 * blockDim.x) {
        uint32_t t[4], sl[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t r = r0 + u * blockDim.x;
            t[u] = r < hi ? tier_of[r] : 0u;
            sl[u] = r < hi ? slot_in_blk[r] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t r = r0 + u * blockDim.x;
            // (skip_plain_walks: a placement call -- k_route placed or listed the reads of the plain walk classes itself and
            // gave them no slot)
            const uint32_t cls = plan_class(t[u]);
            if (r < hi && !(skip_plain_walks && (cls == PLAN_WALK8 || cls == PLAN_WALK16))) list[base[t[u]] + sl[u]] = r;
        }
    }
}

// -----------------------------------------------------------------------------
// k_first_pos: sort key of the reads of one stream's list = first listed position (reads that
// list nothing first).  Sorted by it, the reads of a tile list the same or neighbouring
// positions: the tile looks at an event of the stream once per DISTINCT position, every read
// that lists it takes the delta in the same instructions, and equal reads are evaluated once.
// -----------------------------------------------------------------------------
__global__ void k_first_pos(const uint32_t* __restrict__ list, uint32_t n, const uint32_t* __restrict__ read_off,
                            const uint32_t* __restrict__ read_word, uint32_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = list[i], so = read_off[r];
    keys[i] = (read_off[r + 1] > so) ? w_pos(read_word[so]) + 1u : 0u;
}

// sort key of the reads of a walk class's list: (stream, first listed position) -- the list is grouped by stream
// already (plan ids ascend), so a sort by this key reorders the reads inside every stream's range only
__global__ void k_walk_keys(const uint32_t* __restrict__ list, uint32_t n, const uint8_t* __restrict__ tier_of,
                            const uint32_t* __restrict__ read_off, const uint32_t* __restrict__ read_word,
                            uint32_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = list[i], so = read_off[r];
    const uint32_t first = (read_off[r + 1] > so) ? w_pos(read_word[so]) + 1u : 0u;
    keys[i] = (plan_index(tier_of[r]) << SORT_KEY_BITS) | first;
}

// -----------------------------------------------------------------------------
// launchers (called from capi.cpp)
// -----------------------------------------------------------------------------
hipError_t launch_route(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word, uint32_t n_reads,
                        int use_crowns, uint32_t walk_max_events, uint32_t job_events, uint32_t stack8, uint32_t stack16,
                        uint32_t seed_min_hard, uint32_t seed_min_nodes, uint32_t* job_n, uint8_t* tier_of,
                        int32_t* root_score, uint32_t* blk_counts, uint32_t* tier_info, uint32_t* slot_in_blk,
                        uint32_t* tier_info_next, uint32_t* wsid, const RouteDirect& direct, hipStream_t stream) {
    hipLaunchKernelGGL(k_route, dim3(ROUTE_BLOCKS), dim3(ROUTE_THREADS), 0, stream, m, d_read_off, d_read_word,
                       n_reads, use_crowns, walk_max_events, job_events, std::min(stack8, WALK8_STACK), std::min(stack16, WALK16_STACK),
                       seed_min_hard, seed_min_nodes, job_n, tier_of, root_score, blk_counts, tier_info, slot_in_blk, tier_info_next, wsid, direct);
    return hipGetLastError();
}

hipError_t launch_rebase_index(IxHead* heads, uint32_t n_heads, IxEnt* ents, uint32_t n_ents, uint32_t ent_off, hipStream_t stream) {
    const uint32_t n = std::max(n_heads, n_ents);
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_rebase_index, dim3((n + 255) / 256), dim3(256), 0, stream, heads, n_heads, ents, n_ents, ent_off);
    return hipGetLastError();
}

hipError_t launch_first_pos(const uint32_t* list, uint32_t n, const uint32_t* d_read_off, const uint32_t* d_read_word,
                            uint32_t* keys, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_first_pos, dim3((n + 255) / 256), dim3(256), 0, stream, list, n, d_read_off, d_read_word, keys);
    return hipGetLastError();
}

hipError_t launch_walk_keys(const uint32_t* list, uint32_t n, const uint8_t* tier_of, const uint32_t* d_read_off,
                            const uint32_t* d_read_word, uint32_t* keys, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_walk_keys, dim3((n + 255) / 256), dim3(256), 0, stream, list, n, tier_of, d_read_off, d_read_word, keys);
    return hipGetLastError();
}

hipError_t launch_scatter(const uint8_t* tier_of, const uint32_t* slot_in_blk, uint32_t n_reads, const uint32_t* blk_counts,
                          uint32_t* tier_info, uint32_t* list, bool skip_plain_walks, hipStream_t stream) {
    hipLaunchKernelGGL(k_scatter, dim3(ROUTE_BLOCKS), dim3(ROUTE_THREADS), 0, stream, tier_of, slot_in_blk, n_reads,
                       blk_counts, tier_info, list, skip_plain_walks ? 1 : 0);
    return hipGetLastError();
}

}  // namespace wepp
