// epp_capi.cpp -- wepp_epp_map: host side of WEPP's own read placement
// (wepp_filter::cartesian_map, src/WEPP/initial_filter.cpp:140-239).
//
// The reference walks, for every read, a range tree (arena.cpp:68-169) recursively and
// updates the haplotypes' scores under a mutex.  Here the reads are sorted by window,
// cut into tiles of 64 and groups of tiles; every group gets the slice of the MAT's EPP
// event stream that falls into its genome window, and two sweeps of (tile, chunk) jobs
// produce the per-read and per-haplotype results (epp_kernels.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "epp.hpp"
#include "handle.hpp"
#include "staged_copy.hpp"

namespace {

struct DevPool {                       // device allocations of one call, taken from / returned to the handle's cache
    wepp_mat_t* mat;
    std::vector<std::pair<void*, size_t>> used;
    explicit DevPool(wepp_mat_t* m) : mat(m) {}
    ~DevPool() {
        // (a call that fails half-way may still have kernels in flight on these blocks)
        (void)hipDeviceSynchronize();
        for (auto& b : used) mat->epp_cache.blocks.push_back(b);
    }
    template <typename T>
    hipError_t get(T** out, size_t n) {
        const size_t bytes = (std::max<size_t>(n * sizeof(T), 64) + 255) & ~(size_t)255;
        // the smallest cached block that holds the request without wasting more than half of itself
        size_t best = SIZE_MAX;
        auto& cache = mat->epp_cache.blocks;
        for (size_t i = 0; i < cache.size(); i++) {
            const size_t sz = cache[i].second;
            if (sz >= bytes && sz <= 2 * bytes + (1u << 20) && (best == SIZE_MAX || sz < cache[best].second)) best = i;
        }
        if (best != SIZE_MAX) {
            used.push_back(cache[best]);
            cache.erase(cache.begin() + (std::ptrdiff_t)best);
            *out = (T*)used.back().first;
            return hipSuccess;
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess && !cache.empty()) {
            // out of memory with blocks of other sizes parked in the cache: release them and try again
            for (auto& b : cache) (void)hipFree(b.first);
            cache.clear();
            e = hipMalloc(&p, bytes);
        }
        if (e == hipSuccess) used.emplace_back(p, bytes);
        *out = (T*)p;
        return e;
    }
};

// jobs (tile, stream chunk) a call aims at (WEPP_EPP_TARGET_JOBS): see wepp_epp_map
constexpr uint32_t EPP_TARGET_JOBS = 262144;   // measured at 16 M nodes, 1 M reads: 8192 (one chunk per tile, 15.6 K jobs) 576 ms, 32 K 440, 64 K 402, 128 K 382, 256 K 370, 1 M 362, 4 M 379 ms on the device
struct EppTiming { float select_ms = 0, sweep1_ms = 0, sweep2_ms = 0, finish_ms = 0; uint64_t events_swept = 0, stream_events = 0; uint32_t groups = 0, jobs = 0; };
thread_local EppTiming g_last;

}  // namespace

extern "C" int wepp_mat_dfs_order(const wepp_mat_t* mat, uint32_t* ids) {
    if (!mat || !ids) return set_error(WEPP_EINVAL, "null argument");
    std::memcpy(ids, mat->dfs2id.data(), mat->dfs2id.size() * sizeof(uint32_t));
    return WEPP_OK;
}

extern "C" int wepp_epp_last_timing(double* select_ms, double* sweep1_ms, double* sweep2_ms, double* finish_ms,
                                    uint64_t* events_swept, uint64_t* stream_events, uint32_t* groups,
                                    uint32_t* jobs) {
    if (select_ms) *select_ms = g_last.select_ms;
    if (sweep1_ms) *sweep1_ms = g_last.sweep1_ms;
    if (sweep2_ms) *sweep2_ms = g_last.sweep2_ms;
    if (finish_ms) *finish_ms = g_last.finish_ms;
    if (events_swept) *events_swept = g_last.events_swept;
    if (stream_events) *stream_events = g_last.stream_events;
    if (groups) *groups = g_last.groups;
    if (jobs) *jobs = g_last.jobs;
    return WEPP_OK;
}

extern "C" int wepp_epp_map(wepp_mat_t* mat, const wepp_epp_reads* rd, uint32_t genome_size, uint32_t max_cached_epp,
                            wepp_epp_out* out) {
    if (!mat || !rd || !out) return set_error(WEPP_EINVAL, "null argument");
    const uint32_t R = rd->n_reads;
    const uint32_t N = mat->dev.N;
    if (R && (!rd->read_off || !rd->start || !rd->end || !rd->degree)) return set_error(WEPP_EINVAL, "null read array");
    if (!out->max_parsimony || !out->multiplicity || !out->hap_score) return set_error(WEPP_EINVAL, "null output array");
    if ((out->epp_off == nullptr) != (out->epp_nodes == nullptr)) return set_error(WEPP_EINVAL, "epp_off and epp_nodes go together");
    if (genome_size < EPP_BINS) return set_error(WEPP_EINVAL, "genome_size must be at least NUM_RANGE_BINS (50)");
    const uint64_t W = R ? rd->read_off[R] : 0;
    if (W && !rd->read_word) return set_error(WEPP_EINVAL, "null read_word");
    if (W >= (1ull << 32)) return set_error(WEPP_ELIMIT, "more than 2^32 read words in one call");

    // ---- validation (the reference's preconditions, made explicit) ----------------------
    long long total_degree = 0;
    for (uint32_t r = 0; r < R; r++) {
        if (rd->read_off[r + 1] < rd->read_off[r]) return set_error(WEPP_EINVAL, "read_off is not monotone");
        if (rd->start[r] < 1 || rd->end[r] < rd->start[r] || (uint32_t)rd->end[r] > WEPP_MAX_POSITION)
            return set_error(WEPP_EINVAL, "read " + std::to_string(r) + ": window must satisfy 1 <= start <= end <= 2^20 - 2");
        if (rd->degree[r] < 0) return set_error(WEPP_EINVAL, "read " + std::to_string(r) + ": negative degree");
        total_degree += rd->degree[r];
        uint32_t prev = 0;
        for (uint32_t j = rd->read_off[r]; j < rd->read_off[r + 1]; j++) {
            const uint32_t w = rd->read_word[j];
            const uint32_t pos = w & 0xFFFFFu, ref = (w >> 20) & 15u, mut = (w >> 24) & 15u;
            if (pos == 0 || pos > WEPP_MAX_POSITION || pos <= prev)
                return set_error(WEPP_EINVAL, "read " + std::to_string(r) + ": mutations must be sorted by position, unique, in 1..2^20-2");
            if (mut == ref || mut == 0)
                return set_error(WEPP_EINVAL, "read " + std::to_string(r) + ": a listed mutation must differ from the reference base (sam2pb.cpp:521-535)");
            prev = pos;
        }
    }
    if (R == 0) {
        std::fill(out->hap_score, out->hap_score + N, 0.0);
        if (out->hap_read_counts) std::fill(out->hap_read_counts, out->hap_read_counts + (size_t)N * EPP_BINS, 0);
        if (out->hap_divergence) std::fill(out->hap_divergence, out->hap_divergence + N, std::nan(""));
        if (out->epp_off) out->epp_off[0] = 0;
        return WEPP_OK;
    }
    HIP_TRY(hipSetDevice(mat->device));
    hipStream_t stream = nullptr;

    // ---- reads in window order, tiles, groups ------------------------------------------
    // order by (start, end, index).  Window bounds are genome positions: two stable counting passes (end,
    // then start) instead of a comparison sort through two indirections (≈0.1 s per 1 M reads)
    std::vector<uint32_t> order(R);
    {
        int32_t lo = 0, hi = 0;
        for (uint32_t r = 0; r < R; r++) {
            lo = std::min({lo, rd->start[r], rd->end[r]});
            hi = std::max({hi, rd->start[r], rd->end[r]});
        }
        if (lo < 0 || (uint64_t)hi > (1ull << 24)) {          // not positions: the general way
            std::iota(order.begin(), order.end(), 0u);
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
                if (rd->start[a] != rd->start[b]) return rd->start[a] < rd->start[b];
                if (rd->end[a] != rd->end[b]) return rd->end[a] < rd->end[b];
                return a < b;
            });
        } else {
            std::vector<uint32_t> tmp(R), cnt((size_t)hi + 2);
            auto pass = [&](const int32_t* key, const uint32_t* in, uint32_t* out) {
                std::fill(cnt.begin(), cnt.end(), 0u);
                for (uint32_t s = 0; s < R; s++) cnt[(size_t)key[in ? in[s] : s] + 1]++;
                for (size_t k = 1; k < cnt.size(); k++) cnt[k] += cnt[k - 1];
                for (uint32_t s = 0; s < R; s++) {
                    const uint32_t r = in ? in[s] : s;
                    out[cnt[(size_t)key[r]]++] = r;
                }
            };
            pass(rd->end, nullptr, tmp.data());
            pass(rd->start, tmp.data(), order.data());
        }
    }
    // reads per lane.  4 shares the serial per-event work (broadcasts, flips, atomics) among 256 reads
    // per wave, but a tile that large lists almost every position of its window, so every event takes
    // the allele-lookup path: measured 1.4x slower than 1 -- kept selectable for
    // experiments, and only while its allele table leaves room for several waves per CU.
    uint32_t rpl = 1;
    if (const char* env = std::getenv("WEPP_EPP_RPL")) rpl = std::atoi(env) == 4 ? 4 : 1;
    if (rpl == 4) {
        int32_t longest = 0;
        for (uint32_t r = 0; r < R; r++) longest = std::max(longest, rd->end[r] - rd->start[r]);
        if ((((uint32_t)longest >> 3) + 1) * 64 * 4 * 4 > 48 * 1024) rpl = 1;
    }
    const uint32_t TS = 64 * rpl;
    const uint32_t ntiles = (R + TS - 1) / TS;
    const uint32_t tpg = std::max<uint32_t>(1, (ntiles + EPP_MAX_GROUPS - 1) / EPP_MAX_GROUPS);
    const uint32_t G = (ntiles + tpg - 1) / tpg;
    std::vector<EppGroup> groups(G);
    std::vector<uint32_t> we_max(G);
    uint32_t bm_words = 1, max_span = 0;
    for (uint32_t g = 0; g < G; g++) {
        EppGroup& gr = groups[g];
        gr = EppGroup{};
        gr.tile0 = g * tpg;
        gr.ntiles = std::min(tpg, ntiles - gr.tile0);
        gr.ws = 0xFFFFFFFFu;
        gr.we = 0;
        for (uint32_t t = gr.tile0; t < gr.tile0 + gr.ntiles; t++) {
            uint32_t ts = 0xFFFFFFFFu, te = 0;
            for (uint32_t s = t * TS; s < std::min<uint64_t>(R, (uint64_t)t * TS + TS); s++) {
                const uint32_t r = order[s];
                ts = std::min(ts, (uint32_t)rd->start[r]);
                te = std::max(te, (uint32_t)rd->end[r]);
                max_span = std::max(max_span, (uint32_t)(rd->end[r] - rd->start[r]));
            }
            bm_words = std::max(bm_words, ((te - ts) >> 5) + 1);
            gr.ws = std::min(gr.ws, ts);
            gr.we = std::max(gr.we, te);
        }
        we_max[g] = g ? std::max(we_max[g - 1], gr.we) : gr.we;
    }
    // per-read allele table: one nibble per window position, 8 positions per word, lane-interleaved
    const uint32_t tab_rows = (max_span >> 3) + 1;
    const uint32_t lds_bytes = (bm_words + tab_rows * 64 * rpl) * 4;
    if (lds_bytes > 150 * 1024)
        return set_error(WEPP_ELIMIT, "a tile of 64 reads needs " + std::to_string(lds_bytes) +
                                          " bytes of LDS (window bitmap + allele table): reads too long");

    // ---- device copies of the reads ------------------------------------------------------
    DevPool pool(mat);
    uint32_t *d_off, *d_word, *d_order, *d_wemax;
    int32_t *d_start, *d_end, *d_degree;
    EppGroup* d_groups;
    hipError_t e;
#define GET(p, n) if ((e = pool.get(&p, (n))) != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    GET(d_off, (size_t)R + 1) GET(d_word, W) GET(d_order, R) GET(d_start, R) GET(d_end, R) GET(d_degree, R)
    GET(d_groups, G) GET(d_wemax, G)
    HIP_TRY(hipMemcpyAsync(d_off, rd->read_off, ((size_t)R + 1) * 4, hipMemcpyHostToDevice, stream));
    if (W) HIP_TRY(hipMemcpyAsync(d_word, rd->read_word, W * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_order, order.data(), (size_t)R * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_start, rd->start, (size_t)R * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_end, rd->end, (size_t)R * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_degree, rd->degree, (size_t)R * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_groups, groups.data(), (size_t)G * sizeof(EppGroup), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_wemax, we_max.data(), (size_t)G * 4, hipMemcpyHostToDevice, stream));

    hipEvent_t ev[5];
    for (auto& x : ev) HIP_TRY(hipEventCreate(&x));
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int i = 0; i < 5; i++) (void)hipEventDestroy(e[i]); } } evg{ev};
    HIP_TRY(hipEventRecord(ev[0], stream));

    // ---- window streams --------------------------------------------------------------------
    const uint64_t E = mat->epp_events;
    const uint32_t nblk = (uint32_t)((E + EPP_SEL_EVENTS - 1) / EPP_SEL_EVENTS);
    uint32_t *d_cnt, *d_totals;
    GET(d_cnt, (size_t)G * std::max<uint32_t>(nblk, 1)) GET(d_totals, G)
    std::vector<uint32_t> totals(G, 0);
    if (nblk) {
        HIP_TRY(launch_epp_select_count(mat->epp_word, E, d_groups, d_wemax, G, nblk, d_cnt, stream));
        HIP_TRY(launch_epp_select_scan(d_cnt, G, nblk, d_totals, stream));
        HIP_TRY(hipMemcpyAsync(totals.data(), d_totals, (size_t)G * 4, hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    uint64_t total_events = 0, swept = 0;
    uint32_t n_max = 0;
    for (uint32_t g = 0; g < G; g++) {
        groups[g].n_events = totals[g];
        groups[g].soff = total_events;
        total_events += totals[g];
        n_max = std::max(n_max, totals[g]);
        swept += (uint64_t)totals[g] * groups[g].ntiles;
    }
    // enough jobs to fill the machine when there are few tiles
    static const uint32_t target_jobs = getenv("WEPP_EPP_TARGET_JOBS") ? (uint32_t)std::max(1, atoi(getenv("WEPP_EPP_TARGET_JOBS"))) : EPP_TARGET_JOBS;
    const uint32_t want_chunks = std::max<uint32_t>(1, (target_jobs + ntiles - 1) / ntiles);
    uint32_t chunk_events = std::max<uint32_t>(1024, (n_max + want_chunks - 1) / want_chunks);
    chunk_events = (chunk_events + 63) & ~63u;
    uint64_t n_jobs64 = 0;
    for (uint32_t g = 0; g < G; g++) {
        groups[g].nchunks = std::max<uint32_t>(1, (groups[g].n_events + chunk_events - 1) / chunk_events);
        groups[g].job0 = (uint32_t)n_jobs64;
        n_jobs64 += (uint64_t)groups[g].nchunks * groups[g].ntiles;
    }
    if (n_jobs64 >= (1ull << 31)) return set_error(WEPP_ELIMIT, "too many sweep jobs");
    const uint32_t n_jobs = (uint32_t)n_jobs64;
    HIP_TRY(hipMemcpyAsync(d_groups, groups.data(), (size_t)G * sizeof(EppGroup), hipMemcpyHostToDevice, stream));
    uint32_t *d_stw, *d_stn;
    GET(d_stw, total_events) GET(d_stn, total_events)
    if (nblk) HIP_TRY(launch_epp_select_scatter(mat->epp_word, mat->epp_node, E, d_groups, d_wemax, G, nblk, d_cnt, d_stw, d_stn, stream));
    HIP_TRY(hipEventRecord(ev[1], stream));

    // ---- pass 1, combine -------------------------------------------------------------------
    const size_t rows = (size_t)n_jobs * 64 * rpl;
    int32_t *d_pmin, *d_pnet, *d_best;
    uint32_t *d_pcnt, *d_mult;
    long long* d_fx;
    GET(d_pmin, rows) GET(d_pcnt, rows) GET(d_pnet, rows) GET(d_best, R) GET(d_mult, R) GET(d_fx, R)
    int fx_bits = 62;
    for (long long s = total_degree; s > 0; s >>= 1) fx_bits--;
    fx_bits = std::min(fx_bits, 52);
    EppSweepArgs a{};
    a.groups = d_groups; a.G = G; a.n_jobs = n_jobs; a.R = R; a.N = N;
    a.chunk_events = chunk_events; a.bm_words = bm_words; a.tab_rows = tab_rows;
    a.bin_size = genome_size / EPP_BINS;
    a.st_word = d_stw; a.st_node = d_stn;
    a.read_off = d_off; a.read_word = d_word; a.start = d_start; a.end = d_end; a.degree = d_degree; a.order = d_order;
    a.part_min = d_pmin; a.part_cnt = d_pcnt; a.part_net = d_pnet;
    a.best = d_best; a.mult = d_mult; a.delta_fx = d_fx;
    a.fx_scale = std::ldexp(1.0, fx_bits);
    HIP_TRY(launch_epp_sweep(a, 1, rpl, lds_bytes, stream));
    HIP_TRY(launch_epp_combine(a, tpg, rpl, stream));
    std::vector<int32_t> best_s(R);
    std::vector<uint32_t> mult_s(R);
    HIP_TRY(hipMemcpyAsync(best_s.data(), d_best, (size_t)R * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(mult_s.data(), d_mult, (size_t)R * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(ev[2], stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (uint32_t s = 0; s < R; s++) {
        out->max_parsimony[order[s]] = best_s[s];
        out->multiplicity[order[s]] = mult_s[s];
    }
    // EPP lists of the reads with few enough placements (initial_filter.cpp:205-210)
    std::vector<uint64_t> epp_base(R, ~0ull);
    uint64_t epp_total = 0;
    bool lists_overflow = false;
    if (out->epp_off) {
        out->epp_off[0] = 0;
        for (uint32_t r = 0; r < R; r++) {
            if (out->multiplicity[r] <= max_cached_epp) { epp_base[r] = epp_total; epp_total += out->multiplicity[r]; }
            out->epp_off[r + 1] = epp_total;
        }
        // too small a list buffer does not stop the call: everything else is computed and delivered, the lists stay
        // with the handle for wepp_epp_fetch_lists, and the call reports WEPP_ELIMIT at its end
        lists_overflow = epp_total > out->epp_capacity || (epp_total && !out->epp_nodes);
    }
    mat->epp_pending.clear();

    // ---- pass 2 ----------------------------------------------------------------------------
    const bool want_cnt = out->hap_read_counts || out->hap_divergence;
    uint64_t* d_ebase;
    uint32_t* d_enodes;
    unsigned long long* d_dscore;
    int* d_dcnt = nullptr;
    GET(d_ebase, R) GET(d_enodes, epp_total) GET(d_dscore, (size_t)N + 1)
    if (want_cnt) GET(d_dcnt, ((size_t)N + 1) * EPP_BINS)
    HIP_TRY(hipMemcpyAsync(d_ebase, epp_base.data(), (size_t)R * 8, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync(d_dscore, 0, ((size_t)N + 1) * 8, stream));
    if (want_cnt) HIP_TRY(hipMemsetAsync(d_dcnt, 0, ((size_t)N + 1) * EPP_BINS * 4, stream));
    a.epp_base = d_ebase; a.epp_nodes = d_enodes; a.diff_score = d_dscore; a.diff_cnt = d_dcnt;
    HIP_TRY(launch_epp_sweep(a, 2, rpl, lds_bytes, stream));
    HIP_TRY(hipEventRecord(ev[3], stream));

    // ---- prefix sums -> per-haplotype outputs ------------------------------------------------
    double *d_score, *d_div = nullptr;
    int *d_counts = nullptr, *d_true = nullptr;
    void* d_scratch;
    GET(d_score, N)
    {
        char* sc;
        GET(sc, epp_finish_scratch_bytes(N))
        d_scratch = sc;
    }
    int true_counts[EPP_BINS] = {0};
    if (want_cnt) {
        // arena::build_range_trees, arena.cpp:137-147
        for (uint32_t r = 0; r < R; r++)
            true_counts[std::min<uint32_t>((uint32_t)rd->start[r] / a.bin_size, EPP_BINS - 1)] += rd->degree[r];
        GET(d_true, EPP_BINS)
        HIP_TRY(hipMemcpyAsync(d_true, true_counts, sizeof(true_counts), hipMemcpyHostToDevice, stream));
        if (out->hap_read_counts) GET(d_counts, (size_t)N * EPP_BINS)
        if (out->hap_divergence) GET(d_div, N)
    }
    HIP_TRY(launch_epp_finish(N, d_dscore, 1.0 / a.fx_scale, d_score, d_dcnt, d_true, d_counts, d_div, d_scratch, stream));
    HIP_TRY(hipEventRecord(ev[4], stream));
    // the per-haplotype outputs are gigabytes at 16 M nodes: staged copies (staged_copy.hpp)
    HIP_TRY(d2h_staged(out->hap_score, d_score, (size_t)N * 8, stream));
    if (d_counts) HIP_TRY(d2h_staged(out->hap_read_counts, d_counts, (size_t)N * EPP_BINS * 4, stream));
    if (d_div) HIP_TRY(d2h_staged(out->hap_divergence, d_div, (size_t)N * 8, stream));
    if (epp_total && lists_overflow) {
        try { mat->epp_pending.resize(epp_total); } catch (const std::bad_alloc&) {
            return set_error(WEPP_ENOMEM, "out of host memory for " + std::to_string(epp_total) + " EPP list entries");
        }
        HIP_TRY(d2h_staged(mat->epp_pending.data(), d_enodes, epp_total * 4, stream));
    } else if (epp_total) HIP_TRY(d2h_staged(out->epp_nodes, d_enodes, epp_total * 4, stream));
    HIP_TRY(hipStreamSynchronize(stream));
#undef GET
    g_last = EppTiming{};
    (void)hipEventElapsedTime(&g_last.select_ms, ev[0], ev[1]);
    (void)hipEventElapsedTime(&g_last.sweep1_ms, ev[1], ev[2]);
    (void)hipEventElapsedTime(&g_last.sweep2_ms, ev[2], ev[3]);
    (void)hipEventElapsedTime(&g_last.finish_ms, ev[3], ev[4]);
    g_last.events_swept = swept;
    g_last.stream_events = total_events;
    g_last.groups = G;
    g_last.jobs = n_jobs;
    if (lists_overflow)
        return set_error(WEPP_ELIMIT, "epp_nodes holds " + std::to_string(out->epp_capacity) + " entries, " + std::to_string(epp_total) +
                                      " needed: every other output is complete, fetch the lists with wepp_epp_fetch_lists");
    return WEPP_OK;
}

// the EPP lists of the handle's last wepp_epp_map that did not fit the caller's buffer (nothing is computed again)
extern "C" int wepp_epp_fetch_lists(wepp_mat_t* mat, uint32_t* epp_nodes, uint64_t capacity) {
    if (!mat) return set_error(WEPP_EINVAL, "null argument");
    if (mat->epp_pending.empty()) return set_error(WEPP_EINVAL, "no EPP lists are pending on this handle");
    if (capacity < mat->epp_pending.size())
        return set_error(WEPP_ELIMIT, "epp_nodes holds " + std::to_string(capacity) + " entries, " + std::to_string(mat->epp_pending.size()) + " needed");
    if (!epp_nodes) return set_error(WEPP_EINVAL, "null argument");
    std::memcpy(epp_nodes, mat->epp_pending.data(), mat->epp_pending.size() * 4);
    std::vector<uint32_t>().swap(mat->epp_pending);
    return WEPP_OK;
}
