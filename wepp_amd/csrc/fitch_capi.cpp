// fitch_capi.cpp -- C-ABI of the per-site Fitch-Sankoff pass (mapper_body,
// src/usher_mapper.cpp:7-162, driven by read_vcf(create_new_mat = true),
// src/mutation_annotated_tree.cpp:1907-2031).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wepp_place.h"
#include "errors.hpp"
#include "fitch.hpp"
#include "staged_copy.hpp"
#include "flatmat.hpp"

using namespace wepp;

namespace {
int hipf(hipError_t e, const char* what) {
    return set_error(WEPP_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    // (a buffer that is allocated again -- a retry after a failed upload -- gives its old block back first)
    hipError_t alloc(size_t n) {
        if (p) { (void)hipFree(p); p = nullptr; }
        return hipMalloc(&p, std::max<size_t>(n, 16));
    }
    template <typename T> T* as() { return (T*)p; }
};
}  // namespace

namespace {
thread_local double g_fitch_ms[4] = {0, 0, 0, 0};     // host preparation of the rows, uploads, kernels, sort + decode + copy-out
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

extern "C" int wepp_fitch_last_timing(double* prep_ms, double* upload_ms, double* kernels_ms, double* output_ms) {
    if (prep_ms) *prep_ms = g_fitch_ms[0];
    if (upload_ms) *upload_ms = g_fitch_ms[1];
    if (kernels_ms) *kernels_ms = g_fitch_ms[2];
    if (output_ms) *output_ms = g_fitch_ms[3];
    return WEPP_OK;
}

// Everything about a tree the Fitch-Sankoff pass needs, computed and uploaded ONCE: read_vcf runs
// mapper_body row after row on one tree (src/mutation_annotated_tree.cpp:1962-2031); with a plan the
// flattening, the level / chunk tables and their device copies are not redone per call.
struct wepp_fitch_plan {
    int device = 0;
    FlatMAT f;
    uint32_t N = 0, D = 0, C = 0;
    bool sets_ok_tree = true;                  // no node has too many children for the set form's counters
    std::vector<uint32_t> meta, id2dfs, depth, chunk_start, chunk_depth, chunk_min, chunk_open;
    std::vector<uint32_t> level_off, l_coff, l_par;     // level-synchronous form (filled on first use)
    DevBuf d_meta, d_cs, d_cd, d_cm, d_co, d_lcoff, d_lpar, d_b2i, d_tables, d_id2bfs, d_id2dfs;
    size_t tables_bytes = 0;
    bool dev_topology = false, dev_levels = false, dev_b2i = false;
};

extern "C" int wepp_fitch_plan_create(const wepp_tree_desc* tree, int device, wepp_fitch_plan_t** out) {
    if (!tree || !out) return set_error(WEPP_EINVAL, "null argument");
    *out = nullptr;
    std::unique_ptr<wepp_fitch_plan> plan(new (std::nothrow) wepp_fitch_plan());
    if (!plan) return set_error(WEPP_ENOMEM, "out of host memory");
    plan->device = device;
    // topology only: the mutation lists of `tree` are ignored (a new MAT is being built)
    std::vector<uint32_t> zero_off((size_t)tree->n_nodes + 1, 0);
    wepp_tree_desc topo = *tree;
    topo.mut_off = zero_off.data();
    topo.mut_pos = nullptr; topo.mut_ref = nullptr; topo.mut_par = nullptr; topo.mut_mut = nullptr;
    FlatMAT& f = plan->f;
    std::string err;
    try {
        int rc = flatten_tree(topo, f, err, /*topology_only=*/true);
        if (rc != WEPP_OK) return set_error(rc, err);
    } catch (const std::bad_alloc&) {
        return set_error(WEPP_ENOMEM, "out of host memory while flattening the tree");
    }
    const uint32_t N = f.N;
    if (N >= (1u << 28)) return set_error(WEPP_ELIMIT, "more than 2^28 nodes");
    if (f.max_depth > FITCH_MAX_DEPTH)
        return set_error(WEPP_ELIMIT, "tree depth " + std::to_string(f.max_depth) + " exceeds the LDS stack (" +
                                          std::to_string(FITCH_MAX_DEPTH) + ")");
    std::vector<uint32_t>&meta = plan->meta, &id2dfs = plan->id2dfs, &depth = plan->depth;
    meta.assign(N, 0); id2dfs.assign(N, 0); depth.assign(N, 0);
    std::vector<uint32_t> nchild(N, 0);
    uint32_t max_children = 0;
    for (uint32_t d = 1; d < N; d++) max_children = std::max(max_children, ++nchild[f.parent_dfs[d]]);
    // the set form of the forward pass needs non-empty allele sets (checked per row below) and 15-bit counters
    plan->sets_ok_tree = max_children <= FITCH_SETS_MAX_CHILDREN;
    for (uint32_t d = 0; d < N; d++) {
        if (d) depth[d] = depth[f.parent_dfs[d]] + 1;
        meta[d] = depth[d] | ((f.nstat[d] & NS_LEAF) ? 0x80000000u : 0u);
        id2dfs[f.dfs2id[d]] = d;
    }
    // chunks of consecutive DFS nodes (one wave each) and what is open at their boundaries
    const uint32_t D = f.max_depth + 1;
    uint32_t C = std::max<uint32_t>(1, std::min<uint32_t>(256, N / 2048));
    if (const char* env = std::getenv("WEPP_FITCH_CHUNKS"))      // test hook: force the number of chunks
        C = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)std::atoi(env), N));
    std::vector<uint32_t>&chunk_start = plan->chunk_start, &chunk_depth = plan->chunk_depth, &chunk_min = plan->chunk_min,
                         &chunk_open = plan->chunk_open;
    chunk_start.assign(C + 1, 0); chunk_depth.assign(C + 1, 0); chunk_min.assign(C, 0); chunk_open.assign((size_t)(C + 1) * D, 0);
    for (uint32_t c = 0; c <= C; c++) chunk_start[c] = (uint32_t)((uint64_t)N * c / C);
    for (uint32_t c = 0; c <= C; c++) {
        // nodes open before node a (c < C): its strict ancestors; after the last node: the
        // strict ancestors of that leaf (the single node itself when the tree is one node)
        uint32_t x;
        if (c < C) { x = chunk_start[c]; chunk_depth[c] = depth[x]; }
        else if (N == 1) { chunk_depth[c] = 1; chunk_open[(size_t)c * D] = 0; continue; }
        else { x = N - 1; chunk_depth[c] = depth[x]; }
        uint32_t anc = x;
        for (uint32_t k = chunk_depth[c]; k-- > 0;) {
            anc = f.parent_dfs[anc];
            chunk_open[(size_t)c * D + k] = anc;
        }
    }
    for (uint32_t c = 0; c < C; c++) {
        uint32_t mn = 0xFFFFFFFFu;
        for (uint32_t d = chunk_start[c]; d < chunk_start[c + 1]; d++) mn = std::min(mn, depth[d]);
        chunk_min[c] = mn;
    }
    plan->N = N;
    plan->D = D;
    plan->C = C;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return set_error(WEPP_EDEVICE, "no HIP device available (the Fitch-Sankoff pass has no CPU fallback)");
    if (device < 0 || device >= ndev) return set_error(WEPP_EINVAL, "device index out of range");
    *out = plan.release();
    return WEPP_OK;
}

extern "C" int wepp_fitch_plan_destroy(wepp_fitch_plan_t* plan) {
    if (plan) {
        (void)hipSetDevice(plan->device);
        delete plan;
    }
    return WEPP_OK;
}

extern "C" int wepp_fitch_plan_run(wepp_fitch_plan_t* plan, uint32_t n_sites, const uint8_t* site_ref,
                                   const uint32_t* var_off, const uint32_t* var_node, const uint8_t* var_nuc,
                                   uint64_t capacity, uint64_t* n_out, uint32_t* out_site, uint32_t* out_node,
                                   uint8_t* out_par, uint8_t* out_mut) {
    if (!plan || !n_out || !var_off || (n_sites && !site_ref)) return set_error(WEPP_EINVAL, "null argument");
    *n_out = 0;
    if (n_sites == 0) return WEPP_OK;
    const int device = plan->device;
    FlatMAT& f = plan->f;
    const uint32_t N = plan->N, D = plan->D, C = plan->C;
    const double t_begin = now_ms();
    std::vector<uint32_t>&meta = plan->meta, &id2dfs = plan->id2dfs, &depth = plan->depth, &chunk_start = plan->chunk_start,
                         &chunk_depth = plan->chunk_depth, &chunk_min = plan->chunk_min, &chunk_open = plan->chunk_open;
    // rows: reference base index, tree samples sorted by node index (BFS for the level-synchronous
    // form, DFS for the two stack forms)
    // the CSR over the rows is checked before anything is sized from it or written through it
    if (var_off[0] != 0) return set_error(WEPP_EINVAL, "var_off[0] must be 0");
    for (uint32_t s = 0; s < n_sites; s++)
        if (var_off[s + 1] < var_off[s]) return set_error(WEPP_EINVAL, "var_off not monotone");
    const uint64_t nv = var_off[n_sites];
    if (nv && (!var_node || !var_nuc)) return set_error(WEPP_EINVAL, "null variant arrays");
    bool sets_ok = plan->sets_ok_tree;
    if (const char* env = std::getenv("WEPP_FITCH_SCORES"))      // test hook: force the score form
        if (env[0] == '1') sets_ok = false;
    for (uint64_t k = 0; k < nv; k++)
        if ((var_nuc[k] & 15) == 0) sets_ok = false;     // no base allowed: the scores leave the set forms' range
    bool levels = sets_ok;
    if (const char* env = std::getenv("WEPP_FITCH_DFS"))          // test hook: force the DFS stack forms
        if (env[0] == '1') levels = false;
    std::vector<uint8_t> ref_idx(n_sites);
    for (uint32_t s2 = 0; s2 < n_sites; s2++) {
        const uint8_t r = site_ref[s2] & 15;
        if (r == 0 || (r & (r - 1)))
            return set_error(WEPP_EINVAL, "site_ref must be a single nucleotide (row " + std::to_string(s2) + ")");
        ref_idx[s2] = (uint8_t)__builtin_ctz(r);
    }
    if (nv >= (1ull << 32)) return set_error(WEPP_ELIMIT, "more than 2^32 variants in one call; split the rows");
    // the rows themselves -- node ids to BFS / DFS indices, sorted per row, a node named twice keeping its later
    // entry (usher_mapper.cpp:57-62) -- are prepared on the device (sort_reads.hip: launch_fitch_prepare)
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hipf(e, "hipSetDevice");
    const double t_prep = now_ms();

    DevBuf d_ref, d_voff, d_vdfs, d_vnuc, d_count, d_out, d_inh, d_outp, d_vraw, d_vnraw, d_vdfs2, d_vnuc2, d_ptmp, d_pflags;
    DevBuf &d_meta = plan->d_meta, &d_cs = plan->d_cs, &d_cd = plan->d_cd, &d_cm = plan->d_cm, &d_co = plan->d_co,
           &d_lcoff = plan->d_lcoff, &d_lpar = plan->d_lpar, &d_tables = plan->d_tables;
    // level-synchronous form: the topology in BFS order (levels and sibling groups are contiguous); built once
    std::vector<uint32_t>&level_off = plan->level_off, &l_coff = plan->l_coff, &l_par = plan->l_par;
    if (levels && level_off.empty()) {
        std::vector<uint32_t> bfs2dfs(N);
        for (uint32_t d = 0; d < N; d++) bfs2dfs[f.dfs2bfs[d]] = d;
        l_coff.assign((size_t)N + 1, 0); l_par.assign(N, 0);
        for (uint32_t bidx = 0; bidx < N; bidx++) {
            const uint32_t d = bfs2dfs[bidx];
            if (bidx == 0 || depth[d] != depth[bfs2dfs[bidx - 1]]) level_off.push_back(bidx);
            if (d == 0) continue;
            const uint32_t pb = f.dfs2bfs[f.parent_dfs[d]];
            l_par[bidx] = pb;
            l_coff[pb + 1]++;                              // BFS visits a node's children consecutively, parents in order
        }
        l_coff[0] = 1;                                     // the first child (if any) is BFS node 1
        for (uint32_t i = 0; i < N; i++) l_coff[i + 1] += l_coff[i];
        level_off.push_back(N);
    }
    const uint32_t rows_per_batch = levels ? 64 * FITCH_ROWS_PER_LANE : 64;
    const uint32_t nbatches = (n_sites + rows_per_batch - 1) / rows_per_batch;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    // per batch of 64 rows: the decision tables and the two partial-sum scratch arrays
    const size_t part_bytes = levels ? 0 : (size_t)C * D * 64 * 16;
    const size_t per_batch = (size_t)N * rows_per_batch + 2 * part_bytes;
    // (the decision tables the plan already holds are part of what a run may use: without them in the sum every run
    // after the first saw half the memory and cut its rows into twice as many groups)
    const size_t budget = (free_b + plan->tables_bytes) / 2;
    const uint32_t group = (uint32_t)std::max<size_t>(1, std::min<size_t>(nbatches, budget / std::max<size_t>(per_batch, 1)));
    // per-call buffers; the decision tables (tens of GB at 16 M nodes) and the topology stay with the plan
    const size_t tables_need = (size_t)N * rows_per_batch * group;
    if (tables_need > plan->tables_bytes) {
        if (d_tables.p) { (void)hipFree(d_tables.p); d_tables.p = nullptr; plan->tables_bytes = 0; }
        if ((e = d_tables.alloc(tables_need)) != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        plan->tables_bytes = tables_need;
    }
    if ((e = d_ref.alloc(n_sites)) != hipSuccess ||
        (e = d_voff.alloc((size_t)(n_sites + 1) * 4)) != hipSuccess || (e = d_vdfs.alloc(nv * 4)) != hipSuccess ||
        (e = d_vnuc.alloc(nv)) != hipSuccess || (e = d_vraw.alloc(nv * 4)) != hipSuccess || (e = d_vnraw.alloc(nv)) != hipSuccess ||
        (e = d_vdfs2.alloc(nv * 4)) != hipSuccess || (e = d_vnuc2.alloc(nv)) != hipSuccess || (e = d_pflags.alloc(8)) != hipSuccess ||
        (e = d_inh.alloc(part_bytes * group)) != hipSuccess || (e = d_outp.alloc(part_bytes * group)) != hipSuccess ||
        (e = d_count.alloc(8)) != hipSuccess || (e = d_out.alloc(std::max<uint64_t>(capacity, 1) * 8)) != hipSuccess)
        return set_error(WEPP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    if (!plan->dev_topology) {
        if ((e = d_meta.alloc((size_t)N * 4)) != hipSuccess || (e = d_cs.alloc((C + 1) * 4)) != hipSuccess ||
            (e = d_cd.alloc((C + 1) * 4)) != hipSuccess || (e = d_cm.alloc(C * 4)) != hipSuccess ||
            (e = d_co.alloc(chunk_open.size() * 4)) != hipSuccess)
            return set_error(WEPP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        e = hipMemcpy(d_meta.p, meta.data(), (size_t)N * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_cs.p, chunk_start.data(), (C + 1) * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_cd.p, chunk_depth.data(), (C + 1) * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_cm.p, chunk_min.data(), C * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_co.p, chunk_open.data(), chunk_open.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) return hipf(e, "upload of the topology");
        plan->dev_topology = true;
    }
    if (levels && !plan->dev_levels) {
        if ((e = d_lcoff.alloc(l_coff.size() * 4)) != hipSuccess || (e = d_lpar.alloc(l_par.size() * 4)) != hipSuccess)
            return set_error(WEPP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        e = hipMemcpy(d_lcoff.p, l_coff.data(), l_coff.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_lpar.p, l_par.data(), l_par.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) return hipf(e, "upload of the level tables");
        plan->dev_levels = true;
    }
    // node id -> key of the form in use (BFS index for the level-synchronous kernels, DFS index for the stack forms)
    DevBuf& d_id2key = levels ? plan->d_id2bfs : plan->d_id2dfs;
    if (!d_id2key.p) {
        std::vector<uint32_t> id2key(N);
        for (uint32_t i = 0; i < N; i++) id2key[i] = levels ? f.dfs2bfs[id2dfs[i]] : id2dfs[i];
        if ((e = d_id2key.alloc((size_t)N * 4)) != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        if ((e = hipMemcpy(d_id2key.p, id2key.data(), (size_t)N * 4, hipMemcpyHostToDevice)) != hipSuccess) return hipf(e, "upload of the id map");
    }
    e = hipMemcpy(d_ref.p, ref_idx.data(), n_sites, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_voff.p, var_off, (size_t)(n_sites + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && nv) e = hipMemcpy(d_vraw.p, var_node, nv * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && nv) e = hipMemcpy(d_vnraw.p, var_nuc, nv, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_count.p, 0, 8);
    if (e != hipSuccess) return hipf(e, "upload");
    uint32_t* keys_final = d_vdfs.as<uint32_t>();
    uint8_t* nuc_final = d_vnuc.as<uint8_t>();
    {
        size_t ptmp = 0;
        e = fitch_rows_temp_bytes(nv, n_sites, &ptmp);
        if (e == hipSuccess) e = d_ptmp.alloc(ptmp);
        if (e != hipSuccess) return set_error(WEPP_ENOMEM, std::string("hipMalloc (row preparation): ") + hipGetErrorString(e));
        bool in_b = true;
        e = launch_fitch_prepare(d_vraw.as<uint32_t>(), d_vnraw.as<uint8_t>(), d_voff.as<uint32_t>(), n_sites, nv, N,
                                 d_id2key.as<uint32_t>(), d_vdfs2.as<uint32_t>(), d_vnuc2.as<uint8_t>(), d_vdfs.as<uint32_t>(),
                                 d_vnuc.as<uint8_t>(), d_pflags.as<uint32_t>(), d_ptmp.p, ptmp, &in_b, nullptr);
        if (e == hipErrorInvalidValue) { (void)hipGetLastError(); return set_error(WEPP_EINVAL, "var_node out of range"); }
        if (e != hipSuccess) return hipf(e, "preparation of the rows");
        if (!in_b) { keys_final = d_vdfs2.as<uint32_t>(); nuc_final = d_vnuc2.as<uint8_t>(); }
    }
    const double t_up = now_ms();
    FitchLevels fl{N, levels ? (uint32_t)level_off.size() - 1 : 0, d_lcoff.as<uint32_t>(), d_lpar.as<uint32_t>()};
    FitchTree ft{N, f.max_depth, C, d_meta.as<uint32_t>(), d_cs.as<uint32_t>(), d_cd.as<uint32_t>(),
                 d_cm.as<uint32_t>(), d_co.as<uint32_t>()};
    FitchSites fs{n_sites, d_ref.as<uint8_t>(), d_voff.as<uint32_t>(), keys_final, nuc_final};
    for (uint32_t b0 = 0; b0 < nbatches; b0 += group) {
        const uint32_t nb = std::min(group, nbatches - b0);
        if (levels) {
            e = launch_fitch_levels(fl, level_off.data(), fs, b0, nb, d_tables.as<uint8_t>(), d_count.as<unsigned long long>(),
                                    capacity, d_out.as<uint2>(), nullptr);
            if (e != hipSuccess) return hipf(e, "Fitch-Sankoff kernels");
            continue;
        }
        if (sets_ok)
            e = launch_fitch_forward_sets(ft, fs, b0, nb, d_tables.as<uint8_t>(), d_inh.as<uint2>(), d_outp.as<uint2>(), nullptr);
        else
            e = launch_fitch_forward(ft, fs, b0, nb, d_tables.as<uint8_t>(), d_inh.as<int4>(), d_outp.as<int4>(), nullptr);
        if (e == hipSuccess)
            e = launch_fitch_backward(ft, fs, b0, nb, d_tables.as<uint8_t>(), sets_ok, d_count.as<unsigned long long>(),
                                      capacity, d_out.as<uint2>(), nullptr);
        if (e != hipSuccess) return hipf(e, "Fitch-Sankoff kernels");
    }
    unsigned long long cnt = 0;
    e = hipMemcpy(&cnt, d_count.p, 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hipf(e, "Fitch-Sankoff kernels");
    const double t_kern = now_ms();
    struct Stamp { double a, b, c, d; ~Stamp() { g_fitch_ms[0] = b - a; g_fitch_ms[1] = c - b; g_fitch_ms[2] = d - c; g_fitch_ms[3] = now_ms() - d; } }
        stamp{t_begin, t_prep, t_up, t_kern};
    *n_out = cnt;
    if (cnt > capacity) return set_error(WEPP_ELIMIT, "output buffers too small: " + std::to_string(cnt) + " mutations");
    if (cnt == 0) return WEPP_OK;
    if (!out_site || !out_node || !out_par || !out_mut) return set_error(WEPP_EINVAL, "null output buffer");
    if (levels) {
        // sorted on the device: key = row << 28 | BFS index, 28 + bits(n_sites) key bits
        uint32_t site_bits = 1;
        while (site_bits < 32 && (1ull << site_bits) < n_sites) site_bits++;
        DevBuf d_k0, d_k1, d_v0, d_v1, d_tmp;
        size_t tmp_bytes = 0;
        e = sort_u64_u32_temp_bytes(cnt, 28 + site_bits, &tmp_bytes);
        if (e == hipSuccess && ((e = d_k0.alloc(cnt * 8)) != hipSuccess || (e = d_k1.alloc(cnt * 8)) != hipSuccess ||
                                (e = d_v0.alloc(cnt * 4)) != hipSuccess || (e = d_v1.alloc(cnt * 4)) != hipSuccess ||
                                (e = d_tmp.alloc(tmp_bytes)) != hipSuccess))
            return set_error(WEPP_ENOMEM, std::string("hipMalloc (sort of the mutations): ") + hipGetErrorString(e));
        if (e == hipSuccess) e = launch_fitch_sort_keys(d_out.as<uint2>(), cnt, d_k0.as<unsigned long long>(), d_v0.as<uint32_t>(), nullptr);
        if (e == hipSuccess)
            e = launch_sort_u64_u32(d_k0.as<unsigned long long>(), d_k1.as<unsigned long long>(), d_v0.as<uint32_t>(),
                                    d_v1.as<uint32_t>(), cnt, 28 + site_bits, d_tmp.p, tmp_bytes, nullptr);
        // decoded on the device into the caller's four arrays (the raw queue entries are dead after the key
        // kernel: its memory takes the decoded rows and node ids), then copied out (staged_copy.hpp)
        DevBuf d_pm;
        DevBuf& d_b2i = plan->d_b2i;
        uint32_t* d_site = d_out.as<uint32_t>();
        uint32_t* d_node = d_site + cnt;
        if (e == hipSuccess && (e = d_pm.alloc(cnt * 2)) != hipSuccess)
            return set_error(WEPP_ENOMEM, std::string("hipMalloc (decoding the mutations): ") + hipGetErrorString(e));
        if (e == hipSuccess && !plan->dev_b2i) {
            if ((e = d_b2i.alloc((size_t)N * 4)) != hipSuccess)
                return set_error(WEPP_ENOMEM, std::string("hipMalloc (decoding the mutations): ") + hipGetErrorString(e));
            e = hipMemcpy(d_b2i.p, f.bfs2id.data(), (size_t)N * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) plan->dev_b2i = true;
        }
        if (e == hipSuccess)
            e = launch_fitch_decode(d_k1.as<unsigned long long>(), d_v1.as<uint32_t>(), d_b2i.as<uint32_t>(), cnt, d_site,
                                    d_node, d_pm.as<uint8_t>(), d_pm.as<uint8_t>() + cnt, nullptr);
        if (e == hipSuccess) e = d2h_staged(out_site, d_site, cnt * 4, nullptr);
        if (e == hipSuccess) e = d2h_staged(out_node, d_node, cnt * 4, nullptr);
        if (e == hipSuccess) e = d2h_staged(out_par, d_pm.as<uint8_t>(), cnt, nullptr);
        if (e == hipSuccess) e = d2h_staged(out_mut, d_pm.as<uint8_t>() + cnt, cnt, nullptr);
        if (e != hipSuccess) return hipf(e, "sort / decode / D2H copy of the mutations");
        return WEPP_OK;
    }
    std::vector<uint2> raw(cnt);
    e = hipMemcpy(raw.data(), d_out.p, cnt * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hipf(e, "D2H copy of the mutations");
    // rows in order; inside a row, nodes in BFS order (the order mapper_body visits them, :115)
    std::vector<uint64_t> order(cnt);
    std::iota(order.begin(), order.end(), 0ull);
    // the level-synchronous kernels report BFS indices, the stack forms DFS indices
    auto bfs_of = [&](uint32_t y) { return levels ? (y & 0x0FFFFFFFu) : f.dfs2bfs[y & 0x0FFFFFFFu]; };
    std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
        if (raw[a].x != raw[b].x) return raw[a].x < raw[b].x;
        return bfs_of(raw[a].y) < bfs_of(raw[b].y);
    });
    for (uint64_t i = 0; i < cnt; i++) {
        const uint2 r = raw[order[i]];
        out_site[i] = r.x;
        out_node[i] = levels ? f.bfs2id[r.y & 0x0FFFFFFFu] : f.dfs2id[r.y & 0x0FFFFFFFu];
        out_par[i] = (uint8_t)(1u << ((r.y >> 28) & 3u));
        out_mut[i] = (uint8_t)(1u << ((r.y >> 30) & 3u));
    }
    return WEPP_OK;
}

extern "C" int wepp_fitch_sites(const wepp_tree_desc* tree, int device, uint32_t n_sites, const uint8_t* site_ref,
                                const uint32_t* var_off, const uint32_t* var_node, const uint8_t* var_nuc,
                                uint64_t capacity, uint64_t* n_out, uint32_t* out_site, uint32_t* out_node,
                                uint8_t* out_par, uint8_t* out_mut) {
    if (!tree || !n_out || !var_off || (n_sites && !site_ref)) return set_error(WEPP_EINVAL, "null argument");
    *n_out = 0;
    if (n_sites == 0) return WEPP_OK;
    wepp_fitch_plan_t* plan = nullptr;
    int rc = wepp_fitch_plan_create(tree, device, &plan);
    if (rc != WEPP_OK) return rc;
    rc = wepp_fitch_plan_run(plan, n_sites, site_ref, var_off, var_node, var_nuc, capacity, n_out, out_site, out_node, out_par,
                             out_mut);
    wepp_fitch_plan_destroy(plan);
    return rc;
}
