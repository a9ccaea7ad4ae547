// device_mat.hpp -- the flat MAT as the kernels see it (device pointers) and
// the launcher prototypes shared between the kernel units (route / sweep / walk / pass2 / seed _kernels.hip) and capi.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flatmat.hpp"

namespace wepp {

// device-side mirrors of the flatmat.hpp constants (plain ints for kernels)
constexpr int32_t SCORE_INF_DEV = SCORE_INF;
constexpr uint32_t NS_CNT_MASK_DEV = NS_CNT_MASK;
constexpr uint32_t NS_LEAF_DEV = NS_LEAF, NS_MASKED_DEV = NS_MASKED, NS_ELIG0_DEV = NS_ELIG0, NS_ROOT_DEV = NS_ROOT;
constexpr uint32_t EV_OFF_MASK_DEV = EV_OFF_MASK;
constexpr uint32_t W_EXIT_DEV = W_EXIT, W_LEAF_DEV = W_LEAF, W_PAD_DEV = W_PAD;
constexpr uint32_t WEPP_FLAG_HAS_UNIQUE_DEV = WEPP_FLAG_HAS_UNIQUE;

// padding behind the device copies of a stream's ev_word / ev_lb (events) and blk_sum (summaries): every lane
// of a sweep loads its two events of a block, and the summaries of the next three blocks are prefetched,
// without a look at the end of the stream
constexpr uint32_t EV_TAIL_PAD = 128, SUM_TAIL_PAD = 4;
// one sweep stream (a crown or the whole tree), see flatmat.hpp
struct DevStream {
    uint32_t n, NB, cp_stride, ncp;
    uint32_t eager;   // 1 on crown streams: the per-event bounds (ev_lb) prune there, the block minimum elsewhere
    uint32_t tier;    // index of the stream in the handle (profiling counters)
    uint32_t e_pad;   // E: index of the first padding event behind the stream (lanes past a block's events load it)
    const int64_t* nkey;
    const uint32_t* nstat;
    const uint32_t* blk_node0;
    const uint32_t* blk_eoff;
    const BlkSum* blk_sum;
    const uint32_t* ev_word;
    const uint8_t* ev_meta;
    const uint8_t* ev_lb;
    const uint32_t* cp_off;
    const uint32_t* cp_word;
    const uint32_t* ncnt;     // window streams: nodes an element stands for (nullptr on the others: one each)
};

// the position index and range-query structures of one stream (flatmat.hpp), for k_walk
struct DevWalk {
    uint32_t n, rq_blocks, last_ent, has_pre;   // nodes, blocks of the exact range query, index of the last (sentinel) entry,
                                                // 1: the top byte of an entry's rank is the pre-test of the range it ends (flatmat.hpp: ix_pre)
    SegNode whole;                          // aggregate of the whole stream
    const IxHead* ix_head;
    const IxEnt* ix_ent;
    const uint8_t* ix_nest;
    const NodeRec* nrec;
    const SegNode* rq_pre;
    const SegNode* rq_suf;
    const SegNode* rq_dst;
    const uint8_t* sp;
};

// One window crown (flatmat.hpp: wcrowns) inside the ARENA: the walk structures of all window crowns are
// concatenated, array by array (the DevWalk of slot WC_SLOT holds the arena's base pointers), and a crown is the
// offsets of its slices.  Entry indices (ix_head::off, IxEnt::up) are absolute in the arena; node indices are local.
// Which crown a read walks is a per-READ value (k_route), so a wave of the slot's plans holds lanes of different
// crowns: k_walk keeps these numbers per lane.
struct WcInfo {
    uint32_t n, rq_blocks, last_ent, has_pre;     // n == 0: no such crown
    uint32_t node_off, head_off, nest_off, dst_off;   // into nrec / rq_pre / rq_suf, ix_head, ix_nest, rq_dst
    unsigned long long sp_off;                    // into sp
    int32_t tau;
    uint32_t whole_bfs;                           // BFS index of whole.rank (k_route places reads without events from its table)
    SegNode whole;
};

// tree-wide arrays (global DFS indices)
struct DevMAT {
    uint32_t N, bm_words, max_pos, n_streams;
    uint32_t n_windows;           // window streams (flatmat.hpp: WIN_SIZE positions every WIN_STRIDE)
    uint32_t walk_eager_nodes;    // streams up to this many nodes: exact range query without the sparse pre-test
    int32_t root_base;
    int32_t tau[MAX_STREAMS];
    const uint32_t* node_woff;
    const uint32_t* words;
    const uint32_t* nstat;
    const uint32_t* rank2dfs;
    const uint32_t* dfs2bfs;
    const uint32_t* rank2bfs;     // BFS index of the node with tie-break rank r
    const uint32_t* bfs2dfs;      // inverse of dfs2bfs
    const uint32_t* parent_dfs;   // DFS index of the parent (root: 0)
    const uint8_t* maxnest;       // [max_pos + 1] most mutations at one position along a root path
    const DevWalk* walks;         // [MAX_STREAMS] device array: the streams' walk structures; [WC_SLOT] = the window crowns' arena
    const WcInfo* wc_info;        // [wc_windows * WC_MAX] the window crowns of every genome window, increasing tau
    const DevStream* wc_streams;  // [wc_windows * WC_MAX] their sweep streams (k_sweep_arena)
    uint32_t wc_windows;          // 0: none built
    uint32_t tw_base;             // wc_info[tw_base + t] = tree-wide stream t as a slice of the same arena (what a read that walks it gets as wsid)
    uint32_t win_n[MAX_WINDOWS];  // nodes of window w's stream when it is a crown (all the window's candidates, flatmat.hpp), 0xFFFFFFFF when the whole tree
    // seed signatures (flatmat.hpp): nibble (position, chunk of the whole-tree stream); seed_chunks == 0: none built
    const uint32_t* seed_sig;
    uint32_t seed_stride, seed_chunks, seed_row_words;
};
// reads with more entries than this whose positions lie in one genome window share tile sweeps of the window's stream
// (PLAN_WIN) when that is a crown; shorter ones that cannot walk sweep their window crown alone (k_sweep_arena)
constexpr uint32_t WIN_MIN_ENTRIES = 32;

// A placement call sorts its reads into PLANS: plan id = class * MAX_STREAMS + stream, the windows' plans behind
// them (plan_id / plan_class / plan_index; k_scatter's prefix over the routing blocks costs per plan id, so the
// ids are dense).  Class = how the read is
// placed: by the per-read walk of its own events (k_walk) when it lists at most WALK8_K / WALK16_K positions and
// the intervals it can hold open at once (sum of maxnest over its positions) fit WALK8_STACK / WALK16_STACK, by a
// sweep of the whole stream otherwise.
constexpr uint32_t MAX_PLANS = 128;
constexpr uint32_t PLAN_WALK8 = 0, PLAN_WALK16 = 1, PLAN_SWEEP = 2, PLAN_WALKC8 = 3, PLAN_WALKC16 = 4, PLAN_WIN = 5, PLAN_SEED = 6;
constexpr uint32_t PLAN_WIN_BASE = PLAN_WIN * MAX_STREAMS;
constexpr uint32_t PLAN_SEED_ID = PLAN_WIN_BASE + MAX_WINDOWS;     // ONE plan behind the windows' plans
static_assert(PLAN_SEED_ID < MAX_PLANS, "plan ids");
__host__ __device__ inline uint32_t plan_id(uint32_t cls, uint32_t idx) { return cls == PLAN_SEED ? PLAN_SEED_ID : cls * MAX_STREAMS + idx; }   // (PLAN_WIN is the last class with an index)
__host__ __device__ inline uint32_t plan_class(uint32_t id) { return id == PLAN_SEED_ID ? PLAN_SEED : id >= PLAN_WIN_BASE ? PLAN_WIN : id / MAX_STREAMS; }
__host__ __device__ inline uint32_t plan_index(uint32_t id) { return id == PLAN_SEED_ID ? 0u : id >= PLAN_WIN_BASE ? id - PLAN_WIN_BASE : id % MAX_STREAMS; }
// PLAN_SEED (seed_kernels.hip): a sample that cannot walk and fits no genome window -- a whole-genome
// sample -- but lists at least SEED_MIN_HARD alleles that exclude its reference base: a workgroup of its own counts,
// per chunk of the whole-tree stream, how many of those the chunk's signature could serve, and evaluates only the
// chunks whose count leaves a score within reach of the best one found so far.
constexpr uint32_t SEED_MIN_HARD = 3;               // fewer such entries: no chunk is ever ruled out
constexpr uint32_t SEED_MIN_STREAM_NODES = 65536;   // a read whose tree-wide stream is smaller keeps its tile sweep
constexpr uint32_t SEED_MAX_ENTRIES = 4096;         // entries of a seeded sample (its words are staged in LDS)
#ifndef WEPP_SEED_THREADS
#define WEPP_SEED_THREADS 512
#endif
constexpr uint32_t SEED_THREADS = WEPP_SEED_THREADS;   // one workgroup per sample (256 / 512 / 1024: 8.7 / 5.6 / 6.3 ms per 20 000 samples of ~67 entries at 16 M nodes)
constexpr uint32_t SEED_HEAVY_LEVEL_MIN = 256, SEED_HEAVY_PARTS = 32, SEED_HEAVY_CAP = 128;   // second pass of k_seed: a level of that many chunks hands the sample over; workgroups per such sample; samples per call
constexpr uint32_t SEED_MAX_HARD = 255;             // hard entries counted per chunk (byte counters; a subset is a valid bound)
// PLAN_WIN: a read with more entries than a walk takes, all inside one genome window, sweeps that window's
// stream (the whole tree reduced to the nodes that mutate the window + pseudo-nodes) instead of the whole tree.
// PLAN_WALKC8 / 16: a read with many events at its positions (a frequently mutated site) walks them as several
// independent JOBS of about WALK_JOB_EVENTS events each -- node ranges cut at quantiles of its longest list;
// a job finds the state of a sequential walk at its first node by binary searches in the read's lists and
// the chains of enclosing entries (ix_up) -- whose (score, rank, count) partials k_finalize_jobs combines.
constexpr uint32_t WALK_JOB_EVENTS = 8;
// ... by default; a handle follows its traffic: the events per job of a chunked class in the NEXT call are the class's
// events in this call over WALK_TARGET_JOBS, within [WALK_JOB_EVENTS, WALK_JOB_EVENTS_MAX] -- MANY SHORT chains when a
// class holds a few thousand reads (the default batch: 1 400 reads cut into jobs of 32 events made 66 waves of 32+
// dependent iterations each, 171 us -- the longest kernel of a 0.3 ms step; jobs of 8 events: 78 us), long jobs,
// whose start state (a third of a short job's cycles and bytes) is amortised, when it holds 10^8 events (tree-wide
// walks of N-rich batches).  Speed only: results never depend on it.
constexpr uint32_t WALK_JOB_EVENTS_MAX = 256, WALK_TARGET_JOBS = 1u << 20;
// most stack rows a walk workgroup gets (<= WALK8_STACK / WALK16_STACK, what the kernels take; WEPP_WALK_STACK8 /
// WEPP_WALK_STACK16 lower them): a read that could hold more intervals open at once (sum of ix_nest over its
// positions) is left to the sweeps.  A launch asks LDS for the deepest stack its reads can need (k_route's maximum
// per class), which on the synthetic SARS-CoV-2-like MAT is 17 (up to 8 entries) / 25 (9-16) of 200 K reads.
// Measured (16 M nodes, 1 M reads with 5 % N): rows 16/32 -> 12/20 (2 -> 3.5 waves per SIMD for the 16-entry
// class) 21.5 -> 19.9 ms, nothing on the other legs: the walks are not short of waves.
constexpr uint32_t WALK_WAVES = 2;         // waves per workgroup of k_walk (their LDS regions are private)
constexpr uint32_t WALK_XCDS = 8;          // XCDs of an MI355X: workgroup b of a launch runs on XCD b % 8
// waves of a walk plan are padded to a multiple of this, so that every plan starts at a workgroup index that is a
// multiple of the XCD count and its waves can be dealt to the XCDs in contiguous runs (k_walk)
constexpr uint32_t WALK_PLAN_ALIGN = WALK_XCDS * WALK_WAVES;
__host__ __device__ inline uint32_t walk_plan_waves(uint32_t n_lanes) { return ((n_lanes + 63) / 64 + WALK_PLAN_ALIGN - 1) / WALK_PLAN_ALIGN * WALK_PLAN_ALIGN; }
constexpr uint32_t WALK_EAGER_MAX_NODES = 0;   // streams up to this size would skip the sparse pre-test of a range query (measured slower at every size: off)
constexpr uint32_t WALK_MAX_EVENTS = 6;    // reads with more events at their positions (in their stream) do not walk lane-per-read: a wave per 64 events
                                           // (k_walk_wave) or jobs.  The LONGEST walk of a launch is the launch's time -- an event is two or three dependent
                                           // loop iterations of ~1.5 us, whatever the batch: 16 events 56 - 65 us for 125 K as for 1 M reads, 6 events 37 us
constexpr uint32_t WALK_MAX_EVENTS_BY_JOBS = 16;   // ... while the handle cuts the reads beyond into jobs (a batch full of them: capi.cpp)
constexpr uint32_t WALK_COUNTERS = 1024;   // slots of the walks' iteration counter (summed by the host)
constexpr uint32_t WALK8_K = 8, WALK8_STACK = 16, WALK16_K = 16, WALK16_STACK = 32;
// an open interval on a walk's stack is one dword: (subtree end << WALK_DELTA_BITS) | (delta + WALK_DELTA_BIAS).  The
// delta of one node is a sum over at most WALK16_K listed positions of values in [-2, 2]; the end keeps 25 bits, so
// streams of 2^25 nodes or more cannot be walked (capi.cpp leaves such a tree to the sweeps)
constexpr uint32_t WALK_DELTA_BITS = 7, WALK_DELTA_BIAS = 64;
static_assert(2 * WALK16_K < WALK_DELTA_BIAS, "a node's delta must fit the stack entry's delta field");
constexpr uint32_t WALK8_ROWS = WALK8_STACK, WALK16_ROWS = WALK16_STACK;   // (default stack rows: see WALK_WAVES above)
struct WalkPlanDev {
    uint32_t tier, n_list, wave_end, job0;   // wave_end = first wave (of the launch) after this plan;
    const uint32_t* list;                    // chunked: n_list jobs numbered from job0, list = the class's whole list
};
// what a chunked walk needs beside its plans
struct WalkJobs {
    uint32_t n_list, pad;        // reads of the class
    const uint32_t* job_first;   // blind launches (k_route's tables): [n_reads] first job of a read; job j belongs to read blind_list[j]
    const uint32_t* skip;        // blind launches: != 0 -> the class was left to the host's planned launch: leave at once
    const uint32_t* job_off;     // [n_list] first job of the read at a list position (ascending: a job finds its read by bisection)
    const uint32_t* job_n;       // [n_reads] jobs of a read (by read index)
    int32_t* part_score;         // [jobs]
    uint32_t* part_rank;
    uint32_t* part_cnt;
};
struct WalkPlans {
    uint32_t n;
    WalkPlanDev p[MAX_STREAMS];
};
// cls = 0 / 1: the plans of one class in ONE launch, a wave = 64 reads; writes the final per-read results
hipError_t launch_walk(const DevMAT& m, const WalkPlans& pl, uint32_t cls, uint32_t open_max, const uint32_t* d_read_off,
                       const uint32_t* d_read_word, const int32_t* root_score, uint32_t* best_bfs_j, int32_t* score,
                       uint32_t* num_best, uint32_t* flags, unsigned long long* work_counter, const uint32_t* wsid, hipStream_t stream);
// the plain walks of one class sized BLIND: the reads are list[0 .. *count) (k_route's list and cursor, both on the device),
// the grid is the worst case (max_reads lanes), every read walks the arena slice wsid names; waves beyond the count leave
hipError_t launch_walk_blind(const DevMAT& m, uint32_t cls, uint32_t stack_rows, uint32_t max_reads, const uint32_t* list, const uint32_t* count,
                             const uint32_t* d_read_off, const uint32_t* d_read_word, const int32_t* root_score, uint32_t* best_bfs_j,
                             int32_t* score, uint32_t* num_best, uint32_t* flags, unsigned long long* work_counter, const uint32_t* wsid,
                             hipStream_t stream);
// the chunked walks of one call: job counts gathered into list order (scan input), the walk itself (partials per job) and the combination per read
hipError_t launch_gather_jobs(const uint32_t* list, uint32_t n_list, const uint32_t* job_n, uint32_t* out, hipStream_t stream);
hipError_t launch_walk_jobs(const DevMAT& m, const WalkPlans& pl, uint32_t cls, uint32_t open_max, const WalkJobs& jb, const uint32_t* d_read_off,
                            const uint32_t* d_read_word, const int32_t* root_score, unsigned long long* work_counter,
                            const uint32_t* wsid, hipStream_t stream);
hipError_t launch_finalize_jobs(const DevMAT& m, const uint32_t* list, uint32_t n_list, const WalkJobs& jb,
                                const uint32_t* d_read_off, const uint32_t* d_read_word, uint32_t* best_bfs_j,
                                int32_t* score, uint32_t* num_best, uint32_t* flags, hipStream_t stream);
// the reads with many events: list[0 .. count[0]) a wave each, list[n_reads - count[1] .. n_reads) a block each (k_route's
// two-ended list and its cursors), a persistent grid
hipError_t launch_walk_wave(const DevMAT& m, const uint32_t* list, const uint32_t* count, uint32_t n_reads, const uint32_t* d_read_off,
                            const uint32_t* d_read_word, const int32_t* root_score, uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best,
                            uint32_t* flags, unsigned long long* work_counter, const uint32_t* wsid, hipStream_t stream);
// a chunked class sized blind: jobs[0 .. *n_jobs) (k_route's table), partials into jb.part_*, then the combination over
// clist[0 .. *n_class); both leave at once when *jb.skip != 0
hipError_t launch_walk_jobs_blind(const DevMAT& m, uint32_t cls, uint32_t stack_rows, const WalkJobs& jb, const uint32_t* jobs, const uint32_t* n_jobs,
                                  const uint32_t* clist, const uint32_t* n_class, const uint32_t* d_read_off, const uint32_t* d_read_word,
                                  const int32_t* root_score, uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best, uint32_t* flags,
                                  unsigned long long* work_counter, const uint32_t* wsid, hipStream_t stream);
hipError_t scan_u32_temp_bytes(uint32_t n, size_t* bytes);
hipError_t launch_exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, void* temp, size_t temp_bytes,
                                     hipStream_t stream);

// the plain plans of one call, fused into one launch (k_sweep_multi)
struct SweepPlanDev {
    DevStream st;
    const uint32_t* list;
    uint32_t n_list, T, ntiles, bpc, ent_cap, nchunks;
    uint32_t wg_end, fin_end;   // first sweep workgroup / finalize block after this plan
    int32_t* part_score;
    uint32_t* part_rank;
    uint32_t* part_cnt;
};
struct SweepPlans {
    uint32_t n;
    SweepPlanDev p[MAX_STREAMS];
};

// the window plans of one call (reads with many entries inside one genome window, a tile of them sweeps the window's
// stream), fused into one launch (k_sweep_windows): the streams are named by their index in the handle's device array
struct WinPlanDev {
    uint32_t sid, n_list, T, ntiles, bpc, ent_cap, win_base, nchunks;
    uint32_t wg_end, fin_end;   // first sweep workgroup / finalize block after this plan
    const uint32_t* list;
    int32_t* part_score;
    uint32_t* part_rank;
    uint32_t* part_cnt;
};
struct WinPlans {
    uint32_t n;
    WinPlanDev p[MAX_WINDOWS];
};
static_assert(sizeof(WinPlans) <= 3584, "the window plans travel as a kernel argument");

#ifndef WEPP_ROUTE_BLOCKS
#define WEPP_ROUTE_BLOCKS 256
#endif
constexpr uint32_t ROUTE_BLOCKS = WEPP_ROUTE_BLOCKS;    // grid of k_route / k_scatter (contiguous slices of the reads); 1024 blocks: route
                                          // 30 -> 22 us but the per-block prefix over earlier blocks in k_scatter 29 -> 92 us
constexpr uint32_t ROUTE_THREADS = 1024;  // 16 waves per CU: the per-read chains of dependent loads overlap

// route: tier of every read + per-(block, tier) counts and per-tier max entries; also clears tier_info_next,
// the counters the next call will use (they must be zero before its k_route)
// seed_min_hard / seed_min_nodes: a read left to the sweeps with at least that many reference-excluding entries, on a
// tree-wide stream of at least that many nodes, becomes a seeded sample (seed_min_hard == 0xFFFFFFFF: none does)
// What k_route does beside the plan of every read when `direct` is given (a placement call; wepp_best_nodes only
// routes): a read that walks (plain classes) and has NO event in its stream -- none of its positions is mutated there
// -- is placed on the spot: the stream-wide aggregate is its answer (results into direct.best_bfs_j ...); the other
// plain walkers are appended to their class's list (direct.wlist[cls], cursor in tier_info[TI_WCUR + cls]), from which
// k_walk is launched WITHOUT waiting for the routing counters to reach the host.
struct RouteDirect {
    uint32_t* wlist[2];          // [n_reads] each, or nullptr: no direct placement / lists
    uint32_t* clist[2];          // [BLIND_CHUNKED_READS] the chunked classes' reads
    uint32_t* jobs[2];           // [BLIND_JOB_CAP] read of every job of a chunked class
    uint32_t* job_first;         // [n_reads] first job of a chunked read in its class's table
    uint32_t* wwlist;            // [n_reads] reads with WALK_MAX_EVENTS < events <= WAVE_WALK_MAX_EVENTS (wave_kernels.hip): <= 64 from the front, more from the back
    uint32_t* best_bfs_j;
    int32_t* score;
    uint32_t* num_best;
    uint32_t* flags;
    unsigned long long* work_counter;
    uint32_t cand_min;           // events above which a read counts as a candidate for k_walk_wave (the handle's walk limit while it uses the waves)
    uint32_t ww_max_small, ww_max_big;   // reads per routing block and round listed for k_walk_wave (<= 64 events / more): all or none (capi.cpp), WEPP_WW_BLOCK_MAX_*
};
hipError_t launch_route(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word, uint32_t n_reads,
                        int use_crowns, uint32_t walk_max_events, uint32_t job_events, uint32_t stack8, uint32_t stack16,
                        uint32_t seed_min_hard, uint32_t seed_min_nodes,
                        uint32_t* job_n, uint8_t* tier_of, int32_t* root_score, uint32_t* blk_counts,
                        uint32_t* tier_info, uint32_t* slot_in_blk, uint32_t* tier_info_next, uint32_t* wsid, const RouteDirect& direct,
                        hipStream_t stream);
// entry indices of a stream's slice of the walk arena (IxHead::off, IxEnt::up) made absolute: += ent_off
hipError_t launch_rebase_index(IxHead* heads, uint32_t n_heads, IxEnt* ents, uint32_t n_ents, uint32_t ent_off, hipStream_t stream);
// the seeded samples of one call: a workgroup per sample writes its final results (seed_kernels.hip)
uint32_t seed_lds_bytes(const DevMAT& m, uint32_t ent_cap);
hipError_t seed_set_max_lds(uint32_t bytes);
size_t seed_heavy_bytes();      // the second pass's table (handle.hpp: d_seed_heavy)
hipError_t launch_seed(const DevMAT& m, const DevStream& full, const uint32_t* list, uint32_t n_list, uint32_t ent_cap,
                       const uint32_t* d_read_off, const uint32_t* d_read_word, const int32_t* root_score,
                       uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best, uint32_t* flags,
                       unsigned long long* work_counter, void* heavy_table, hipStream_t stream);
// order of the reads that sweep the whole-tree stream: by first listed position (sort_reads.hip)
constexpr uint32_t SORT_KEY_BITS = 21;      // position + 1 (0 = the read lists nothing)
constexpr uint32_t SORT_MIN_READS = 4096;   // below this a sweep costs less than the sort
// the walks' lists go by (stream, first listed position): reads of one amplicon -- the same few hundred position lists
// of the index -- then sit in neighbouring lanes and waves, and a wave's gathers hit lines its neighbours have just
// brought into the L2 instead of opening a DRAM row each.  Key = stream << SORT_KEY_BITS | position + 1.
constexpr uint32_t WALK_SORT_KEY_BITS = SORT_KEY_BITS + 4;
constexpr uint32_t WALK_SORT_MIN_READS = 32768;  // a class with fewer reads keeps the caller's order (the sort is ~8 launches on its chain: 50 us of a 0.3 ms step)
static_assert(MAX_STREAMS <= 16, "stream index in the walks' sort key");
hipError_t launch_first_pos(const uint32_t* list, uint32_t n, const uint32_t* d_read_off, const uint32_t* d_read_word,
                            uint32_t* keys, hipStream_t stream);
hipError_t launch_walk_keys(const uint32_t* list, uint32_t n, const uint8_t* tier_of, const uint32_t* d_read_off,
                            const uint32_t* d_read_word, uint32_t* keys, hipStream_t stream);
hipError_t sort_reads_temp_bytes(uint32_t n, size_t* bytes);
hipError_t launch_sort_reads(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                             uint32_t n, void* temp, size_t temp_bytes, hipStream_t stream, uint32_t key_bits = SORT_KEY_BITS);
// scatter: read indices grouped by tier into `list` (tier t occupies [tier_off[t], tier_off[t+1]))
hipError_t launch_scatter(const uint8_t* tier_of, const uint32_t* slot_in_blk, uint32_t n_reads, const uint32_t* blk_counts,
                          uint32_t* tier_info, uint32_t* list, bool skip_plain_walks, hipStream_t stream);
hipError_t launch_sweep(const DevMAT& m, const DevStream& st, const uint32_t* d_read_off,
                        const uint32_t* d_read_word, const int32_t* root_score, const uint32_t* list,
                        uint32_t n_list, uint32_t T,
                        uint32_t ntiles, uint32_t nchunks, uint32_t blocks_per_chunk, bool s_in_lds, bool dense,
                        bool win_table, uint32_t ent_cap, uint32_t key_cap, uint32_t lds_bytes, int32_t* part_score,
                        uint32_t* part_rank, uint32_t* part_cnt, hipStream_t stream);
// waves of a dense / window workgroup (one tile: they share its tables and sweep a chunk each).  16: a window's
// candidate crown is a few hundred blocks, and a tile's set-up -- a third of the cycles there -- is paid once per
// workgroup: 1.2 kb reads 3.03 -> 2.83 ms per 200 K against 8 (4: 3.99); on whole-tree window streams (small trees)
// 8 was 6 % faster (107 against 114 ms)
#ifndef WEPP_DENSE_WAVES
#define WEPP_DENSE_WAVES 16
#endif
#ifndef WEPP_SWEEP_WAVES
#define WEPP_SWEEP_WAVES 1
#endif
constexpr uint64_t SWEEP_CHUNK_BYTES = 1ull << 20;   // 1 MB of stream per chunk (an XCD's L2 holds 4 MB; 0.75 / 1 / 1.5 / 2 / 3 MB: 236 / 235 / 238 / 240 / 250 ms)
constexpr uint64_t SWEEP_MAX_PARTIAL_BYTES = 16ull << 30;   // per-(chunk, read) partial results of one stream: chunks get longer beyond
constexpr uint32_t SWEEP_WAVES = WEPP_SWEEP_WAVES;            // independent sweeps (waves) per workgroup of k_sweep_multi
constexpr uint32_t DENSE_WAVES_PER_WG = WEPP_DENSE_WAVES;     // waves sharing one tile's LDS index in the dense variant
// LDS bytes of a k_sweep workgroup: bitmap + read words (+ dense: sorted keys + owners + accumulators)
constexpr uint32_t DENSE_WINDOW = 4096;   // positions behind the tile's smallest one with a direct index into the sorted keys
// window plans of long reads (k_sweep<true, true, true>): table entries (window positions + a sentinel, padded so
// that what follows stays 16-byte aligned) and the per-wave scratch (64 event records of 16 B, net[64], H[64],
// bound[64], marker[64])
constexpr uint32_t WIN_TAB = (WIN_SIZE + 1 + 7) & ~7u;
constexpr uint32_t WIN_WAVE_BYTES = 64 * 16 + 192 * 4 + 64 * 4;
inline uint32_t sweep_lds_bytes(uint32_t bm_words, uint32_t ent_cap, uint32_t key_cap, bool dense, bool win_table = false) {
    // window plans of long reads: read words (position-major) + per window position the mask of the tile's
    // reads and the index of their first word + per-wave scratch + scan scratch
    if (win_table) return ent_cap * 4 + WIN_TAB * 10 + DENSE_WAVES_PER_WG * (WIN_WAVE_BYTES + 4) + 64 * 4;   // (+ the tile's best scores so far)
    return bm_words * 4 + ent_cap * (dense ? 5 : 4) +
           (dense ? key_cap * 4 + DENSE_WAVES_PER_WG * 3 * 64 * 4 + DENSE_WINDOW * 2 : 0);
}
// read words per tile: the plain variant is bounded by its LDS request, the dense variant by
// the 13-bit entry index of its sorted keys (pos:19 | idx:13, hence also max_pos < 2^19)
constexpr uint32_t MAX_TILE_ENTRIES = 8192;
constexpr uint32_t DENSE_MAX_POS = (1u << 19) - 1;
constexpr uint32_t DENSE_MIN_READ_WORDS = 16; // tiles of reads this long keep the sorted position index
constexpr uint32_t ARENA_CHUNKS = 16;      // waves (chunks of its window crown) per read of k_sweep_arena
hipError_t launch_sweep_arena(const DevMAT& m, const DevStream* wc_streams, const uint32_t* wsid, const uint32_t* d_read_off,
                              const uint32_t* d_read_word, const int32_t* root_score, const uint32_t* list, uint32_t n_list,
                              uint32_t ent_cap, uint32_t lds_bytes, int32_t* part_score, uint32_t* part_rank, uint32_t* part_cnt,
                              hipStream_t stream);
// every window plan of a call in one launch, and their combination in one
hipError_t launch_sweep_windows(const DevMAT& m, const DevStream* d_wstreams, const WinPlans& pl, const uint32_t* d_read_off,
                                const uint32_t* d_read_word, const int32_t* root_score, uint32_t lds_bytes, hipStream_t stream);
hipError_t launch_finalize_windows(const DevMAT& m, const WinPlans& pl, const uint32_t* d_read_off, const uint32_t* d_read_word,
                                   uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best, uint32_t* flags, hipStream_t stream);
hipError_t launch_sweep_multi(const DevMAT& m, const SweepPlans& pl, const uint32_t* d_read_off,
                              const uint32_t* d_read_word, const int32_t* root_score, uint32_t lds_bytes,
                              hipStream_t stream);
// finalize: 1, 4, 16 or 64 lanes per read by the number of chunk partials it has to combine (a wave holds
// 64 / lanes consecutive reads: partial loads stay coalesced); 256-thread blocks.  One wave per read for
// every plan beyond 8 chunks (the previous rule) made 60 % of the threads of a default step
__host__ __device__ inline uint32_t finalize_lanes_per_read(uint32_t nchunks) {
    return nchunks <= 8 ? 1u : nchunks <= 32 ? 4u : nchunks <= 128 ? 16u : 64u;
}
inline uint32_t finalize_blocks(uint32_t n_list, uint32_t nchunks) {
    const uint32_t reads_per_block = 256 / finalize_lanes_per_read(nchunks);
    return (n_list + reads_per_block - 1) / reads_per_block;
}
hipError_t launch_finalize_multi(const DevMAT& m, const SweepPlans& pl, const uint32_t* d_read_off,
                                 const uint32_t* d_read_word, uint32_t* best_bfs_j, int32_t* score, uint32_t* num_best,
                                 uint32_t* flags, hipStream_t stream);
hipError_t launch_finalize(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word,
                           const uint32_t* list, uint32_t n_list, uint32_t nchunks, const int32_t* part_score,
                           const uint32_t* part_rank, const uint32_t* part_cnt, uint32_t* best_bfs_j,
                           int32_t* score, uint32_t* num_best, uint32_t* flags, hipStream_t stream);
hipError_t launch_scores(const DevMAT& m, const DevStream& full, const uint32_t* d_read_off,
                         const uint32_t* d_read_word, uint32_t n_reads, int32_t* d_out, hipStream_t stream);
hipError_t launch_best_nodes(const DevMAT& m, const DevStream& st, const uint32_t* d_read_off, const uint32_t* d_read_word,
                             const uint32_t* list, uint32_t n_list, const int32_t* d_best, const unsigned long long* d_off,
                             uint32_t* d_cursor, uint32_t* d_nodes, hipStream_t stream);
// one thread per (read, ambiguous entry) pair listed in `pairs` (read index, word index)
hipError_t launch_imputed(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word,
                          const uint32_t* d_best_bfs_j, const uint32_t* d_pairs, uint32_t n_pairs,
                          uint8_t* d_nuc, hipStream_t stream);
// excess mutations of (read, node) pairs: d_out_off == nullptr counts into d_counts, else emits packed
// words pos:20 | ref:4 | par:4 | mut:4 at d_out + d_out_off[pair]
hipError_t launch_excess(const DevMAT& m, const uint32_t* d_read_off, const uint32_t* d_read_word,
                         const uint32_t* d_pair_read, const uint32_t* d_pair_bfs_j, uint32_t n_pairs,
                         const unsigned long long* d_out_off, uint32_t* d_counts, uint32_t* d_out, hipStream_t stream);
hipError_t sweep_set_max_lds(uint32_t bytes);

// layout of tier_info (uint32), indexed by plan id: counts, max entries of one read, offsets into the list
// then, per chunked class and stream, the jobs of its chunked walks
constexpr uint32_t TI_COUNT = 0, TI_MAXK = MAX_PLANS, TI_OFF = 2 * MAX_PLANS, TI_JOBS = 3 * MAX_PLANS + 1,
                   TI_OPEN = TI_JOBS + 2 * MAX_STREAMS,      // [4] deepest stack of the walk classes (WALK8, WALK16, WALKC8, WALKC16)
                   TI_EVENTS = TI_OPEN + 4,                  // [2] events (in units of 64) of the reads of the two chunked classes
                   TI_WCUR = TI_EVENTS + 2,                  // [2] reads k_route appended to the plain walk classes' lists (the walks' grids are sized blind)
                   TI_RESOLVED = TI_WCUR + 2,                // [1] reads k_route placed itself (no event in their stream: the stream-wide aggregate is the answer)
                   TI_JCUR = TI_RESOLVED + 1,                // [2] jobs k_route entered into the chunked classes' job tables (their walks are sized blind too)
                   TI_CCUR = TI_JCUR + 2,                    // [2] reads it entered into the chunked classes' lists
                   TI_JOVER = TI_CCUR + 2,                   // [2] != 0: the class outgrew its blind tables (or is large enough to be sorted): the host plans it
                   TI_WWCUR = TI_JOVER + 2,                  // [2] reads k_route listed for the walk without a walk (wave_kernels.hip): <= 64 events, more
                   TI_W16WAVE = TI_WWCUR + 2,                // [1] plain walkers of 9 - 16 entries k_route listed for k_walk_wave instead (few of them in their block)
                   TI_WWCAND = TI_W16WAVE + 1,               // [2] reads with 17 .. 64 / 65 .. 256 events in this call, whichever way they were placed
                   TI_WORDS = TI_WWCAND + 2;
// the chunked walk classes sized blind: tables of this many jobs / reads per class; a class that outgrows them (N-rich
// batches on the tree-wide streams: millions of jobs, which the planned path also sorts by position) is left to the host
constexpr uint32_t BLIND_JOB_CAP = 1u << 20, BLIND_CHUNKED_READS = 32768;
// a read with more than WALK_MAX_EVENTS and at most this many events is placed by a wave of its own, lane = list entry
// (wave_kernels.hip); beyond, its walk is cut into jobs (the chunked classes).  A multiple of 64.
constexpr uint32_t WALK16_TO_WAVE_MAX = 8;     // plain walkers of 9 - 16 entries per routing block up to which they go to k_walk_wave
constexpr uint32_t WAVE_WALK_MAX_EVENTS = 256;
constexpr uint32_t WW_CALL_MAX_SMALL = 16384, WW_CALL_MAX_BIG = 2048;     // reads with 17 .. 64 / 65 .. 256 events in a call of a million up to which the NEXT call gives them to k_walk_wave; beyond, their walks are cut into jobs

}  // namespace wepp
