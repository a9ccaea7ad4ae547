// walk_kernels.hip -- the per-read walk: a read visits only the events of the positions IT lists, through the
// stream's position index, with range queries over the static scores in between.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_mat.hpp"
#include "place_dev.hpp"

namespace wepp {

// -----------------------------------------------------------------------------
// k_walk: a read visits only the events of the positions IT lists.
//
// lane = read.  The stream's position index (DevWalk) gives, per listed position, the mutations of
// the stream's nodes at that position in stream order, each with the node's index and the end of its
// subtree: the read merges its (at most KW) lists, keeps the intervals it has entered on a stack
// (they are nested: subtrees), and between two consecutive events -- where its running c_S is
// constant and no node carries one of its positions -- asks a range query for the best statically
// eligible node: a sparse table of the minimum static score says whether anything in the range can
// reach the read's best (almost never), and only then a segment tree gives the exact (score, rank,
// count).  The node of an event is evaluated with the formula of the sweep's node-by-node path.
// Work per read ~ events at its positions in the stream, instead of the whole stream per tile:
// 2 events instead of 270 blocks on the 17 K-node crown, ~3.5 K instead of 250 K blocks on the
// whole tree for a read with three entries.  Same results (tests/walk_model.py is the CPU model).
// -----------------------------------------------------------------------------
template <int KW, int SD, bool CHUNKED>
__global__ __launch_bounds__(64 * WALK_WAVES) void k_walk(DevMAT m, WalkPlans pl, WalkJobs jb, uint32_t sd_rows,
                                              const uint32_t* __restrict__ read_off,
                                              const uint32_t* __restrict__ read_word,
                                              const int32_t* __restrict__ root_score, uint32_t* __restrict__ best_bfs_j,
                                              int32_t* __restrict__ score_out, uint32_t* __restrict__ num_best,
                                              uint32_t* __restrict__ flags, unsigned long long* __restrict__ work_counter,
                                              const uint32_t* __restrict__ wsid, const uint32_t* __restrict__ blind_list,
                                              const uint32_t* __restrict__ blind_count) {
    // wave-private LDS: the allele fields of the read words, 16 bits each [KW / 2][64]; the list cursors [KW][64];
    // the interval stack [sd_rows][64] -- sd_rows = the deepest stack a read of this launch can need (k_route's
    // maximum of open_max over the class, <= SD).  The walk waits on memory: what it gains from a wave more
    // per SIMD is nearly proportional, and its LDS request is what limits them.
    extern __shared__ uint32_t lds_all[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* S16 = lds_all + wv * (KW / 2 + KW + sd_rows) * 64;
    uint32_t* cur_l = S16 + (KW / 2) * 64;
    uint32_t* stk = cur_l + KW * 64;
    // the read word of list j rebuilt from its 9 allele bits (the position is not needed again)
    auto sword = [&](int j) -> uint32_t { return ((S16[(j >> 1) * 64 + lane] >> ((j & 1) * 16)) & 0x1FFu) << 20; };
    const uint32_t unit = blockIdx.x * WALK_WAVES + wv;
    // BLIND (plain walks of a placement call): the launch was sized for the worst case before the routing counters
    // reached the host; the reads are k_route's list blind_list[0 .. *blind_count) and every read walks the arena slice
    // wsid names.  Waves beyond the count leave at once.
    // (chunked classes sized blind: blind_list = k_route's job table, job j belongs to read blind_list[j], its chunk is
    // j - jb.job_first[read]; a class the host plans instead -- *jb.skip != 0 -- leaves at once)
    const bool blind = blind_count != nullptr;
    if (CHUNKED && blind && *jb.skip) return;
    const uint32_t n_blind = blind ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*blind_count) : 0u;
    if (unit >= (blind ? walk_plan_waves(n_blind) : pl.p[pl.n - 1].wave_end)) return;
#ifdef WEPP_WALK_STATS   // (profiling build: wave cycles by phase into the work counters, tools/walk_probe.py prints them)
    unsigned long long ts_[6];
    ts_[0] = __builtin_amdgcn_s_memtime();
#define WALK_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); ts_[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WALK_STAMP(i)
#endif
    // what the walk asks memory for, in bytes (DESIGN.md 4.2 "algorithmic bytes"): the rare requests are added per
    // lane (lane_bytes), the two of the loop body -- a 32-byte index entry per node event, a sparse-table byte per
    // range pre-test -- are counted per wave with a population count of the lanes that issue them
    uint32_t lane_bytes = 0, n_ent = 0, n_spb = 0;
    uint32_t pi = 0;
    if (!blind) while (pi + 1 < pl.n && unit >= pl.p[pi].wave_end) pi++;
    WalkPlanDev q = pl.p[pi];
    if (blind) { q.tier = WC_SLOT; q.n_list = n_blind; q.wave_end = walk_plan_waves(n_blind); q.job0 = 0; q.list = blind_list; }
    // Workgroups are handed to the eight XCDs round-robin (workgroup b runs on XCD b % 8), each with its own L2.  A
    // plan's waves start at a workgroup index that is a multiple of 8 and their number is a multiple of 16
    // (walk_plan_waves), so the plan's workgroup i is on XCD i % 8: XCD x takes the x-th CONTIGUOUS eighth of the
    // plan's tiles -- neighbouring reads of the position-sorted list, i.e. the same few amplicons' lists of the index.
    // (Per plan, not per launch: the plans of a launch differ in cost per read by orders of magnitude.)
    uint32_t tile;
    {
        const uint32_t first = (pi && !blind) ? pl.p[pi - 1].wave_end : 0u;
        const uint32_t wgs = (q.wave_end - first) / WALK_WAVES, wg = (unit - first) / WALK_WAVES;
        tile = ((wg % WALK_XCDS) * (wgs / WALK_XCDS) + wg / WALK_XCDS) * WALK_WAVES + wv;
    }
    const DevWalk ix = m.walks[q.tier];
    const uint32_t slot = tile * 64 + lane;
    const bool have = slot < q.n_list;
    // plain: a lane = a read of the plan's list.  CHUNKED: a lane = a job = (read, chunk of its walk)
    uint32_t rd = 0, chunk = 0, n_chunks = 1, job = 0;
    if (CHUNKED && blind) {
        if (have) {
            job = slot;
            rd = blind_list[job];
            chunk = job - jb.job_first[rd];
            n_chunks = jb.job_n[rd];
            lane_bytes += 4 + 4 + 4;
        }
    } else if (CHUNKED) {
        // the read a job belongs to = the last list position whose first job is <= job (job_off ascends).
        // The wave's jobs are consecutive: its first job is located by a bisection on wave-uniform values
        // (scalar loads), the other lanes' reads lie within the next 64 list positions, whose offsets go to
        // LDS for a short per-lane bisection.
        const uint32_t job_first = q.job0 + tile * 64;
        uint32_t lo = 0, hi = jb.n_list;                  // invariant: job_off[lo] <= job_first < job_off[hi] (or hi == n_list)
        while (hi - lo > 1) {
            const uint32_t mid = (uint32_t)__builtin_amdgcn_readfirstlane((int)((lo + hi) >> 1));
            if (jb.job_off[mid] <= job_first) lo = mid; else hi = mid;
        }
        const uint32_t lp0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
        stk[lane] = lp0 + 1 + lane < jb.n_list ? jb.job_off[lp0 + 1 + lane] : 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (have) {
            job = job_first + lane;
            uint32_t a = 0, b = 64;                       // number of the 64 offsets that are <= job
            while (a < b) {
                const uint32_t mid = (a + b) >> 1;
                if (stk[mid] <= job) a = mid + 1; else b = mid;
            }
            const uint32_t lp = lp0 + a;
            rd = q.list[lp];
            chunk = job - (a ? stk[a - 1] : jb.job_off[lp0]);
            n_chunks = jb.job_n[rd];
            lane_bytes += 4 + 4 + 4;                  // its job offset, list entry and job count
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        rd = have ? q.list[slot] : 0u;
    }
    WALK_STAMP(1);          // job / read decoded
    const uint32_t so = have ? read_off[rd] : 0u;
    const uint32_t k = have ? read_off[rd + 1] - so : 0u;
    const int root_sc = have ? root_score[rd] : 0;
    // The stream's numbers, per LANE: the plans of slot WC_SLOT hold reads of different window crowns (k_route picked
    // one per read: wsid), whose structures are slices of the arena the slot's DevWalk points at (device_mat.hpp:
    // WcInfo); on every other plan the offsets are zero and the numbers the plan's own.
    uint32_t L_n = ix.n, L_rqb = ix.rq_blocks, L_last = ix.last_ent, L_pre = ix.has_pre, L_node = 0, L_head = 0, L_dst = 0;
    size_t L_sp = 0;
    SegNode L_whole = ix.whole;
    if (q.tier == WC_SLOT && have) {
        const WcInfo wi = m.wc_info[wsid[rd]];
        L_n = wi.n; L_rqb = wi.rq_blocks; L_last = wi.last_ent; L_pre = wi.has_pre; L_node = wi.node_off; L_head = wi.head_off;
        L_dst = wi.dst_off; L_sp = (size_t)wi.sp_off; L_whole = wi.whole;
        lane_bytes += 4 + 64;
    }
    // list entry, two offsets, root score; per listed position its word and list head (chunked: also the next head)
    if (have) lane_bytes += (CHUNKED ? 0u : 4u) + 8u + 4u + k * (CHUNKED ? 20u : 12u);

    // ---- set-up: the read's words, the start of every position's list, its first node ----
    uint32_t head[KW];
    int c = 0;
    uint32_t long_off = 0, long_len = 0;     // CHUNKED: the read's longest list (the chunks are its quantiles)
    uint32_t s_pair = 0;
#pragma unroll
    for (int j = 0; j < KW; j++) {
        head[j] = NONE;
        uint32_t s9 = 0;
        if ((uint32_t)j < k) {
            const uint32_t w = read_word[so + j];
            const uint32_t p = w_pos(w);
            // a position beyond the tree's last mutated one has no list: the last sentinel stands in
            IxHead h{L_last, NONE};
            if (p <= m.max_pos) h = ix.ix_head[L_head + p];
            const uint32_t e = h.off;
            s9 = (w >> 20) & 0x1FFu;
            cur_l[j * 64 + lane] = e;
            if (CHUNKED) {
                const uint32_t len = p <= m.max_pos ? ix.ix_head[L_head + p + 1].off - e - 1u : 0u;
                if (len > long_len) { long_len = len; long_off = e; }
            } else {
                head[j] = h.first_node;
            }
            if (!rw_missing(w)) c += ((rw_mut(w) & rw_ref(w)) == 0) ? 1 : 0;
        }
        if (j & 1) S16[(j >> 1) * 64 + lane] = s_pair | (s9 << 16);
        else s_pair = s9;
    }
    WALK_STAMP(2);          // words and list heads staged
    uint32_t n = L_n;               // one past the last node this lane looks at
    uint32_t pos = 0;               // next node nobody has looked at
    uint32_t sp = 0;                // open intervals on the stack
    uint32_t top_end = NONE;
    int top_d = 0;
    if (CHUNKED) {
        // this job's nodes [pos, n): cut at the quantiles of the longest list (k_route made sure it has at
        // least n_chunks entries)
        if (chunk + 1 < n_chunks) { n = ix.ix_ent[long_off + (uint32_t)(((uint64_t)(chunk + 1) * long_len) / n_chunks)].node; lane_bytes += 4; }
        if (chunk) { pos = ix.ix_ent[long_off + (uint32_t)(((uint64_t)chunk * long_len) / n_chunks)].node; lane_bytes += 4; }
        // ---- the state of a sequential walk when it reaches `pos` ----
        // every list's cursor at its first entry >= pos: binary searches, four lists side by side (their
        // loads in flight together); cursors and words live in LDS, so the loops over the lists stay rolled
        const bool mid_stream = have && chunk != 0;
#pragma unroll 1
        for (uint32_t j0 = 0; j0 < (uint32_t)KW; j0 += 4) {
            if (!__ballot(mid_stream && j0 < k)) break;
            uint32_t lo[4], hi[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                lo[u] = 0; hi[u] = 0;
                if (j0 + u < k && mid_stream) {
                    const uint32_t p = w_pos(read_word[so + j0 + u]);
                    lo[u] = cur_l[(j0 + u) * 64 + lane];
                    hi[u] = p <= m.max_pos ? ix.ix_head[L_head + p + 1].off - 1u : lo[u];       // (the sentinel stays out)
                    lane_bytes += 4 + 4;
                }
            }
            // first entry of the list goes to the stack region for a moment: the predecessor test below needs it
            bool searching = true;
            const uint32_t first0 = lo[0], first1 = lo[1], first2 = lo[2], first3 = lo[3];
            while (__ballot(searching)) {
                uint32_t probe[4];
                searching = false;
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) probe[u] = lo[u] < hi[u] ? ix.ix_ent[(lo[u] + hi[u]) >> 1].node : 0u;
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    if (lo[u] < hi[u]) {
                        lane_bytes += 4;
                        const uint32_t mid = (lo[u] + hi[u]) >> 1;
                        if (probe[u] < pos) lo[u] = mid + 1; else hi[u] = mid;
                        searching = searching || lo[u] < hi[u];
                    }
                }
            }
            // the intervals open at `pos`: a list's predecessor entry if its subtree reaches past pos, else the
            // first one up its chain of enclosing entries that does, and every entry enclosing that one
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                if (j0 + u < k && mid_stream) {
                    const uint32_t first = u == 0 ? first0 : u == 1 ? first1 : u == 2 ? first2 : first3;
                    const uint32_t sw = sword((int)(j0 + u));
                    cur_l[(j0 + u) * 64 + lane] = lo[u];
                    uint32_t e = lo[u] > first ? lo[u] - 1u : NONE;
                    IxEnt ent{};
                    while (e != NONE) {
                        ent = ix.ix_ent[e];
                        lane_bytes += 32;
                        if (ent.end > pos) break;
                        e = ent.up;
                    }
                    while (e != NONE) {
                        const int d = enter_delta(ent.word, sw);
                        c += d;
                        if (d != 0) {
                            // insertion by subtree end, outermost at the bottom (the intervals are nested)
                            const uint32_t end = ent.end;
                            uint32_t at = sp;
                            while (at > 0 && (stk[(at - 1) * 64 + lane] >> WALK_DELTA_BITS) < end) { stk[at * 64 + lane] = stk[(at - 1) * 64 + lane]; at--; }
                            stk[at * 64 + lane] = (end << WALK_DELTA_BITS) | (uint32_t)(d + (int)WALK_DELTA_BIAS);
                            sp++;
                        }
                        e = ent.up;
                        if (e != NONE) { ent = ix.ix_ent[e]; lane_bytes += 32; }
                    }
                }
            }
        }
        if (sp) {
            const uint32_t e = stk[(sp - 1) * 64 + lane];
            top_end = e >> WALK_DELTA_BITS;
            top_d = (int)(e & ((1u << WALK_DELTA_BITS) - 1u)) - (int)WALK_DELTA_BIAS;
        }
#pragma unroll
        for (int j = 0; j < KW; j++)
            if ((uint32_t)j < k) head[j] = ix.ix_ent[cur_l[j * 64 + lane]].node;
        lane_bytes += 4 * k;
    }
    if (!have) pos = n;
    WALK_STAMP(3);          // (chunked) start state found
    int bs = root_sc + 1;          // the root always competes: nothing worse can win or tie
    uint32_t br = 0xFFFFFFFFu, cnt = 0, bhu = 0;
    uint32_t iters = 0;

    // a candidate: score, tie-break rank, how many nodes it stands for, has_unique of the node of that rank
    auto take = [&](int sc, uint32_t rk, uint32_t kk, uint32_t hu) {
        if (sc < bs) { bs = sc; br = rk; cnt = kk; bhu = hu; }
        else if (sc == bs) { cnt += kk; if (rk < br) { br = rk; bhu = hu; } }
    };
    if (!CHUNKED) {
        // none of the read's positions is mutated in this stream (most reads of the small crowns): every node
        // scores base + c, and the stream-wide aggregate is the answer
        uint32_t any = head[0];
#pragma unroll
        for (int j = 1; j < KW; j++) any = min(any, head[j]);
        if (pos < n && any == NONE) {
            if (L_whole.cnt && L_whole.base + c <= bs) take(L_whole.base + c, L_whole.rank, L_whole.cnt, L_whole.hu);
            pos = n;
        }
    }

    // small streams: nearly every range between two events holds a node that can tie the best (a crown is
    // made of low-score nodes), so the exact query is issued at once, with the other loads of the iteration;
    // large ones ask the sparse table first (there nearly every range fails it)
    const bool eager = L_n <= m.walk_eager_nodes;
#ifdef WEPP_WALK_STATS
    uint32_t st_live = 0, st_pass = 0;
#endif
    while (__ballot(pos < n)) {
        iters++;
#ifdef WEPP_WALK_STATS
        st_live += (uint32_t)__popcll(__ballot(pos < n));
#endif
        uint32_t i_next = head[0];
#pragma unroll
        for (int j = 1; j < KW; j++) i_next = min(i_next, head[j]);
        const bool live = pos < n;
        const uint32_t stop = min(min(i_next, top_end), n);
        const bool at_node = live && i_next < top_end && i_next < n;
        const unsigned long long b_at = __ballot(at_node);
        n_ent += (uint32_t)__popcll(b_at);
        // ---- everything this iteration reads from memory is requested here, together ----
        // the node of the next event: its record, the list entry that carries it and that list's next node
        IxEnt ent{};
        uint32_t sw = 0, ecur = 0;
        int js = 0;
        if (at_node) {
#pragma unroll
            for (int j = KW - 1; j >= 0; j--) js = head[j] == i_next ? j : js;
            sw = sword(js);
            ecur = cur_l[js * 64 + lane];
            ent = ix.ix_ent[ecur];             // 32 bytes: the mutation, the list's next node and the node's own record
        }
        // the sparse-table byte of [pos, stop): the minimum over [pos, pos + 2^lvl), the first level that reaches
        // `stop`.  A lane whose range ends at the fetched entry's node asks that entry's byte first (use_pre);
        // every other lane's table byte is requested here, with the entry, not behind it
        const bool ranged = live && stop > pos;
        const bool use_pre = ranged && !eager && L_pre && at_node && k >= IX_PRE_MIN_LISTS;
        size_t sp_at = 0;
        uint32_t mn_early = SP_NONE;
        if (ranged && !eager) {
            const uint32_t len = stop - pos;
            const uint32_t lvl = len > 1 ? 32u - (uint32_t)__builtin_clz(len - 1) : 0u;
            sp_at = L_sp + (size_t)lvl * L_n + pos;
            if (!use_pre) mn_early = ix.sp[sp_at];
        }
        n_spb += (uint32_t)__popcll(__ballot(ranged && !eager && !use_pre));
        // ---- the nodes [pos, stop): none of them carries a listed position, c is constant ----
        if (live && stop > pos) {
            const uint32_t last = stop - 1;
            const uint32_t ba = pos / RQ_BLK, bl = last / RQ_BLK;
            bool pass = true;
            if (!eager) {
                // the range ends at the node of the entry fetched above: that entry's byte is the minimum of a
                // superset (everything since its list's previous entry), no table byte needed unless it passes
                // (a read with one or two lists gains nothing: its ranges ARE the ranges between its list's entries, and
                // when such a range cannot be skipped the table byte would be fetched after the entry instead of with it)
                bool by_entry = false;
                uint32_t mn = mn_early;
                if (use_pre) {
                    const uint32_t pb = ent.rank >> IX_RANK_BITS;
                    by_entry = pb == SP_NONE || (pb < SP_CLAMP && (int)pb + c > bs);
                    if (!by_entry) { mn = ix.sp[sp_at]; lane_bytes += 1; }   // (rare on the large streams: the byte of the table after all)
                }
                pass = !by_entry && mn != SP_NONE && (mn >= SP_CLAMP || (int)mn + c <= bs);
            }
#ifdef WEPP_WALK_STATS
            st_pass += (uint32_t)__popcll(__ballot(pass));
#endif
            if (__ballot(pass)) {
                if (pass) {
                    // exact aggregate of the statically eligible nodes of [pos, stop): suffix of the first node's
                    // block, disjoint sparse table over the whole blocks in between, prefix of the last node's
                    // block -- four independent 16-byte loads (flatmat.hpp)
                    SegNode ag{SCORE_INF_DEV, 0xFFFFFFFFu, 0u, 0u};
                    auto join = [&](const SegNode x) {
                        if (x.base < ag.base) ag = x;
                        else if (x.base == ag.base) { ag.cnt += x.cnt; if (x.rank < ag.rank) { ag.rank = x.rank; ag.hu = x.hu; } }
                    };
                    if (ba == bl) {
                        // inside one block: its prefix up to the last node, unless the range starts behind the
                        // block's first node -- then node by node
                        lane_bytes += 16;
                        if (pos == ba * RQ_BLK) join(ix.rq_pre[L_node + last]);
                        else if (stop == L_n || stop == (ba + 1) * RQ_BLK) join(ix.rq_suf[L_node + pos]);
                        else
                            for (uint32_t i = pos; i < stop; i++) {
                                if (i > pos) lane_bytes += 16;
                                const NodeRec x = ix.nrec[L_node + i];
                                if (x.nstat & NS_ELIG0_DEV) {
                                    const uint32_t hu = (x.nstat & NS_ROOT_DEV) ? 0u : (x.nstat & NS_MASKED_DEV) ? 1u :
                                                        (((x.nstat >> 14) & NS_CNT_MASK_DEV) < (x.nstat & NS_CNT_MASK_DEV) ? 1u : 0u);
                                    join(SegNode{x.base, x.rank, 1u, hu});
                                }
                            }
                    } else {
                        const uint32_t lo = ba + 1, hi = bl - 1;
                        const SegNode none{SCORE_INF_DEV, 0xFFFFFFFFu, 0u, 0u};
                        const uint32_t L = lo < hi ? 31u - (uint32_t)__builtin_clz(lo ^ hi) : 0u;
                        const SegNode* trow = ix.rq_dst + L_dst + (size_t)L * L_rqb;
                        const SegNode s1 = ix.rq_suf[L_node + pos], s2 = ix.rq_pre[L_node + last];
                        const SegNode s3 = lo <= hi ? trow[lo] : none, s4 = lo < hi ? trow[hi] : none;
                        join(s1); join(s2); join(s3); join(s4);
                        lane_bytes += 32 + (lo <= hi ? 16 : 0) + (lo < hi ? 16 : 0);
                    }
                    if (ag.cnt && ag.base + c <= bs) take(ag.base + c, ag.rank, ag.cnt, ag.hu);
                }
            }
            pos = stop;
        }
        if (live && pos < n) {
            if (!at_node) {
                // the innermost open interval ends here: its nodes are behind us
                c -= top_d;
                sp--;
                if (sp) {
                    const uint32_t e = stk[(sp - 1) * 64 + lane];
                    top_end = e >> WALK_DELTA_BITS;
                    top_d = (int)(e & ((1u << WALK_DELTA_BITS) - 1u)) - (int)WALK_DELTA_BIAS;
                } else { top_end = NONE; top_d = 0; }
            }
        }
        if (b_at) {
            if (at_node) {
                // every listed mutation the node carries (nearly always one: the entry fetched above)
                const uint32_t node = i_next;
                int adj = 0, dcom = 0, dsum = 0;
                const uint32_t end = ent.end;
                bool more = true;
                while (more) {
                    cur_l[js * 64 + lane] = ecur + 1;
                    own_adjust(ent.word, sw, adj, dcom);
                    // descendants take the allele; the root also scores itself with it (usher_mapper.cpp:266-271)
                    if (end > node + 1 || node == 0) dsum += enter_delta(ent.word, sw);
                    more = false;
                    int jn = 0;
#pragma unroll
                    for (int j = KW - 1; j >= 0; j--) {
                        if (j == js) head[j] = ent.next_node;
                        if (head[j] == node) { more = true; jn = j; }
                    }
                    if (more) {
                        js = jn;
                        sw = sword(js);
                        ecur = cur_l[js * 64 + lane];
                        ent = ix.ix_ent[ecur];
                        lane_bytes += 32;
                    }
                }
                const uint32_t nst = ent.nstat;
                const uint32_t nmut = nst & NS_CNT_MASK_DEV, ncom0 = (nst >> 14) & NS_CNT_MASK_DEV;
                const bool leaf = nst & NS_LEAF_DEV, masked = nst & NS_MASKED_DEV, root = nst & NS_ROOT_DEV;
                bool elig;
                int sc;
                uint32_t hu = 0;
                if (root) { elig = true; sc = ent.base + c + dsum; }
                else if (masked) { elig = false; sc = 0; }
                else {
                    sc = ent.base + c + adj;
                    const int ncom = (int)ncom0 + dcom;
                    elig = leaf ? (ncom > 0) : (ncom > 0 || ncom == (int)nmut);     // usher_mapper.cpp:455-456
                    hu = ncom < (int)nmut ? 1u : 0u;                                // :184,199,262
                }
                if (elig && sc <= bs) take(sc, L_pre ? ent.rank & IX_RANK_MASK : ent.rank, 1u, hu);
                if (dsum != 0 && end > node + 1) {
                    // k_route admits a read only if its open intervals always fit (sum of ix_nest <= SD)
                    stk[sp * 64 + lane] = (end << WALK_DELTA_BITS) | (uint32_t)(dsum + (int)WALK_DELTA_BIAS);
                    sp++;
                    top_end = end;
                    top_d = dsum;
                }
                c += dsum;
                pos = node + 1;
            }
        }
    }
    WALK_STAMP(4);          // walked
    if (have) {
        if (CHUNKED) {
            jb.part_score[job] = bs;
            jb.part_rank[job] = br;
            jb.part_cnt[job] = (cnt << 1) | bhu;     // (the job's has_unique rides in bit 0)
            lane_bytes += 12;
        } else {
            lane_bytes += 4 + 16;                    // rank -> BFS index, the four results
            // the root always competes, so br is a rank; the clamp only keeps a broken invariant in bounds
            if (best_bfs_j) best_bfs_j[rd] = m.rank2bfs[br < m.N ? br : 0u];
            if (score_out) score_out[rd] = bs;
            if (num_best) num_best[rd] = cnt;
            if (flags) flags[rd] = bhu ? WEPP_FLAG_HAS_UNIQUE_DEV : 0u;
        }
    }
    // (1024 counters: thousands of waves adding to ONE address queue up at the memory side)
#ifdef WEPP_WALK_STATS
    WALK_STAMP(5);          // results written
    if (work_counter && lane == 0) {
        unsigned long long* wc = work_counter + (CHUNKED ? 16 : 0);
        for (int i = 0; i < 5; i++) atomicAdd(wc + i, ts_[i + 1] - ts_[i]);
        atomicAdd(wc + 5, 1ull);
        atomicAdd(wc + 6, (unsigned long long)iters);
        // lane-level counts: live lanes summed over the iterations, node events, table bytes, exact queries, lanes with work
        atomicAdd(wc + 7, (unsigned long long)st_live);
        atomicAdd(wc + 8, (unsigned long long)n_ent);
        atomicAdd(wc + 9, (unsigned long long)n_spb);
        atomicAdd(wc + 10, (unsigned long long)st_pass);
        atomicAdd(wc + 11, (unsigned long long)__popcll(__ballot(have)));
    }
#else
    // plain walks count in the first half of the slots, chunked ones in the second; the bytes in a second array of
    // WALK_COUNTERS slots behind the iterations
    {
        const uint32_t wave_bytes = wave_sum_u32(lane_bytes);
        if (work_counter && lane == 0) {
            const uint32_t at = (CHUNKED ? WALK_COUNTERS / 2 : 0) + (unit & (WALK_COUNTERS / 2 - 1));
            atomicAdd(work_counter + at, (unsigned long long)iters);
            atomicAdd(work_counter + WALK_COUNTERS + at, (unsigned long long)wave_bytes + 32ull * n_ent + n_spb);
        }
    }
#endif
}

// job counts in list order (the input of the scan)
__global__ void k_gather_jobs(const uint32_t* __restrict__ list, uint32_t n_list, const uint32_t* __restrict__ job_n,
                              uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_list) out[i] = job_n[list[i]];
}
// a wave per 64 reads of the chunked class: a read with few jobs is combined by its own lane, one with many
// by the whole wave (lane-strided loads, butterfly reduction).  part_cnt = (count << 1) | has_unique.
__global__ __launch_bounds__(256) void k_finalize_jobs(DevMAT m, const uint32_t* __restrict__ list, uint32_t n_list, WalkJobs jb,
                                uint32_t* __restrict__ best_bfs_j, int32_t* __restrict__ score,
                                uint32_t* __restrict__ num_best, uint32_t* __restrict__ flags, const uint32_t* __restrict__ n_list_dev) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // (blind: the class's reads are k_route's list of *n_list_dev reads, a read's jobs start at jb.job_first[read])
    if (n_list_dev) {
        if (*jb.skip) return;
        n_list = *n_list_dev;
    }
    const bool valid = i < n_list;
    const uint32_t r = valid ? list[i] : 0u;
    const uint32_t j0 = valid ? (n_list_dev ? jb.job_first[r] : jb.job_off[i]) : 0u, nj = valid ? jb.job_n[r] : 0u;
    int bs = 0x7FFFFFFF;
    uint32_t br = 0xFFFFFFFFu, cnt = 0, bhu = 0;
    auto take = [&](int& b, uint32_t& rk, uint32_t& ct, uint32_t& h, int s, uint32_t pr, uint32_t pc, uint32_t ph) {
        if (pc == 0) return;
        if (s < b) { b = s; rk = pr; ct = pc; h = ph; }
        else if (s == b) { ct += pc; if (pr < rk) { rk = pr; h = ph; } }
    };
    constexpr uint32_t SMALL = 48;
    if (nj <= SMALL) {
        // (four partials per round: their loads are in flight together)
        uint32_t c = 0;
        for (; c + 4 <= nj; c += 4) {
            uint32_t pc[4], pr[4];
            int ps[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) { pc[u] = jb.part_cnt[j0 + c + u]; ps[u] = jb.part_score[j0 + c + u]; pr[u] = jb.part_rank[j0 + c + u]; }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) take(bs, br, cnt, bhu, ps[u], pr[u], pc[u] >> 1, pc[u] & 1u);
        }
        for (; c < nj; c++) {
            const uint32_t pc = jb.part_cnt[j0 + c];
            take(bs, br, cnt, bhu, jb.part_score[j0 + c], jb.part_rank[j0 + c], pc >> 1, pc & 1u);
        }
    }
    unsigned long long big = __ballot(nj > SMALL);
    while (big) {
        const int l = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t bj0 = (uint32_t)__builtin_amdgcn_readlane((int)j0, l), bnj = (uint32_t)__builtin_amdgcn_readlane((int)nj, l);
        int ws = 0x7FFFFFFF;
        uint32_t wr = 0xFFFFFFFFu, wc = 0, wh = 0;
        for (uint32_t c = lane; c < bnj; c += 64) {
            const uint32_t pc = jb.part_cnt[bj0 + c];
            take(ws, wr, wc, wh, jb.part_score[bj0 + c], jb.part_rank[bj0 + c], pc >> 1, pc & 1u);
        }
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) {
            const int os = __shfl_xor(ws, msk, 64);
            const uint32_t orr = (uint32_t)__shfl_xor((int)wr, msk, 64), oc = (uint32_t)__shfl_xor((int)wc, msk, 64);
            const uint32_t oh = (uint32_t)__shfl_xor((int)wh, msk, 64);
            take(ws, wr, wc, wh, os, orr, oc, oh);
        }
        if ((int)lane == l) { bs = ws; br = wr; cnt = wc; bhu = wh; }
    }
    // (a read of the class without jobs was placed by a wave of its own, wave_kernels.hip: nothing to combine)
    if (valid && nj) {
        if (best_bfs_j) best_bfs_j[r] = m.rank2bfs[br < m.N ? br : 0u];
        if (score) score[r] = bs;
        if (num_best) num_best[r] = cnt;
        if (flags) flags[r] = bhu ? WEPP_FLAG_HAS_UNIQUE_DEV : 0u;
    }
}

// -----------------------------------------------------------------------------
// launchers (called from capi.cpp)
// -----------------------------------------------------------------------------
// LDS of a walk workgroup: WALK_WAVES waves, each KW / 2 rows of read alleles + KW rows of cursors + sd_rows of
// stack (64 lanes x 4 bytes a row)
static uint32_t walk_lds_bytes(uint32_t kw, uint32_t sd_rows) { return WALK_WAVES * (kw / 2 + kw + sd_rows) * 256; }
// stack rows of a launch: the deepest stack its reads can need (k_route), at least one row for the job decode's
// scratch, never more than the class admits
static uint32_t walk_stack_rows(uint32_t open_max, uint32_t sd) { return std::min(sd, std::max(open_max, 2u)); }

hipError_t launch_walk(const DevMAT& m, const WalkPlans& pl, uint32_t cls, uint32_t open_max, const uint32_t* d_read_off,
                       const uint32_t* d_read_word, const int32_t* root_score, uint32_t* best_bfs_j, int32_t* score,
                       uint32_t* num_best, uint32_t* flags, unsigned long long* work_counter, const uint32_t* wsid, hipStream_t stream) {
    if (pl.n == 0) return hipSuccess;
    const uint32_t waves = pl.p[pl.n - 1].wave_end;
    const dim3 grid((waves + WALK_WAVES - 1) / WALK_WAVES), block(64 * WALK_WAVES);
    const WalkJobs none{};
    if (cls == PLAN_WALK8) {
        const uint32_t sd = walk_stack_rows(open_max, WALK8_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK8_K, (int)WALK8_STACK, false>), grid, block, walk_lds_bytes(WALK8_K, sd), stream, m, pl,
                           none, sd, d_read_off, d_read_word, root_score, best_bfs_j, score, num_best, flags, work_counter, wsid,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    } else {
        const uint32_t sd = walk_stack_rows(open_max, WALK16_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK16_K, (int)WALK16_STACK, false>), grid, block, walk_lds_bytes(WALK16_K, sd), stream, m, pl,
                           none, sd, d_read_off, d_read_word, root_score, best_bfs_j, score, num_best, flags, work_counter, wsid,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    }
    return hipGetLastError();
}

hipError_t launch_walk_blind(const DevMAT& m, uint32_t cls, uint32_t stack_rows, uint32_t max_reads, const uint32_t* list, const uint32_t* count,
                             const uint32_t* d_read_off, const uint32_t* d_read_word, const int32_t* root_score, uint32_t* best_bfs_j,
                             int32_t* score, uint32_t* num_best, uint32_t* flags, unsigned long long* work_counter, const uint32_t* wsid,
                             hipStream_t stream) {
    if (max_reads == 0) return hipSuccess;
    const uint32_t waves = walk_plan_waves(max_reads);
    const dim3 grid(waves / WALK_WAVES), block(64 * WALK_WAVES);
    WalkPlans pl{};
    pl.n = 1;
    const WalkJobs none{};
    if (cls == PLAN_WALK8) {
        const uint32_t sd = walk_stack_rows(stack_rows, WALK8_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK8_K, (int)WALK8_STACK, false>), grid, block, walk_lds_bytes(WALK8_K, sd), stream, m, pl,
                           none, sd, d_read_off, d_read_word, root_score, best_bfs_j, score, num_best, flags, work_counter, wsid, list, count);
    } else {
        const uint32_t sd = walk_stack_rows(stack_rows, WALK16_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK16_K, (int)WALK16_STACK, false>), grid, block, walk_lds_bytes(WALK16_K, sd), stream, m, pl,
                           none, sd, d_read_off, d_read_word, root_score, best_bfs_j, score, num_best, flags, work_counter, wsid, list, count);
    }
    return hipGetLastError();
}

hipError_t launch_gather_jobs(const uint32_t* list, uint32_t n_list, const uint32_t* job_n, uint32_t* out, hipStream_t stream) {
    if (n_list == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_jobs, dim3((n_list + 255) / 256), dim3(256), 0, stream, list, n_list, job_n, out);
    return hipGetLastError();
}

hipError_t launch_walk_jobs(const DevMAT& m, const WalkPlans& pl, uint32_t cls, uint32_t open_max, const WalkJobs& jb,
                            const uint32_t* d_read_off, const uint32_t* d_read_word, const int32_t* root_score,
                            unsigned long long* work_counter, const uint32_t* wsid, hipStream_t stream) {
    if (pl.n == 0) return hipSuccess;
    const uint32_t waves = pl.p[pl.n - 1].wave_end;
    const dim3 grid((waves + WALK_WAVES - 1) / WALK_WAVES), block(64 * WALK_WAVES);
    if (cls == PLAN_WALKC8) {
        const uint32_t sd = walk_stack_rows(open_max, WALK8_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK8_K, (int)WALK8_STACK, true>), grid, block, walk_lds_bytes(WALK8_K, sd), stream, m, pl, jb,
                           sd, d_read_off, d_read_word, root_score, (uint32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr,
                           (uint32_t*)nullptr, work_counter, wsid, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    } else {
        const uint32_t sd = walk_stack_rows(open_max, WALK16_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK16_K, (int)WALK16_STACK, true>), grid, block, walk_lds_bytes(WALK16_K, sd), stream, m, pl, jb,
                           sd, d_read_off, d_read_word, root_score, (uint32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr,
                           (uint32_t*)nullptr, work_counter, wsid, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    }
    return hipGetLastError();
}

hipError_t launch_finalize_jobs(const DevMAT& m, const uint32_t* list, uint32_t n_list, const WalkJobs& jb,
                                const uint32_t* d_read_off, const uint32_t* d_read_word, uint32_t* best_bfs_j,
                                int32_t* score, uint32_t* num_best, uint32_t* flags, hipStream_t stream) {
    if (n_list == 0) return hipSuccess;
    hipLaunchKernelGGL(k_finalize_jobs, dim3((n_list + 255) / 256), dim3(256), 0, stream, m, list, n_list, jb,
                       best_bfs_j, score, num_best, flags, (const uint32_t*)nullptr);
    return hipGetLastError();
}

hipError_t launch_walk_jobs_blind(const DevMAT& m, uint32_t cls, uint32_t stack_rows, const WalkJobs& jb, const uint32_t* jobs, const uint32_t* n_jobs,
                                  const uint32_t* clist, const uint32_t* n_class, const uint32_t* d_read_off, const uint32_t* d_read_word,
                                  const int32_t* root_score, uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best, uint32_t* flags,
                                  unsigned long long* work_counter, const uint32_t* wsid, hipStream_t stream) {
    const uint32_t waves = walk_plan_waves(BLIND_JOB_CAP);
    const dim3 grid(waves / WALK_WAVES), block(64 * WALK_WAVES);
    WalkPlans pl{};
    pl.n = 1;
    if (cls == PLAN_WALKC8) {
        const uint32_t sd = walk_stack_rows(stack_rows, WALK8_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK8_K, (int)WALK8_STACK, true>), grid, block, walk_lds_bytes(WALK8_K, sd), stream, m, pl, jb,
                           sd, d_read_off, d_read_word, root_score, (uint32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr,
                           (uint32_t*)nullptr, work_counter, wsid, jobs, n_jobs);
    } else {
        const uint32_t sd = walk_stack_rows(stack_rows, WALK16_STACK);
        hipLaunchKernelGGL((k_walk<(int)WALK16_K, (int)WALK16_STACK, true>), grid, block, walk_lds_bytes(WALK16_K, sd), stream, m, pl, jb,
                           sd, d_read_off, d_read_word, root_score, (uint32_t*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr,
                           (uint32_t*)nullptr, work_counter, wsid, jobs, n_jobs);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_finalize_jobs, dim3((BLIND_CHUNKED_READS + 255) / 256), dim3(256), 0, stream, m, clist, 0u, jb,
                       best_bfs_j, score, num_best, flags, n_class);
    return hipGetLastError();
}

}  // namespace wepp
