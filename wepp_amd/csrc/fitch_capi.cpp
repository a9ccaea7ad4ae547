// fitch_capi.cpp -- C-ABI of the per-site Fitch-Sankoff pass (mapper_body,
// src/usher_mapper.cpp:7-162, driven by read_vcf(create_new_mat = true),
// src/mutation_annotated_tree.cpp:1907-2031).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
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
    hipError_t alloc(size_t n) { return hipMalloc(&p, std::max<size_t>(n, 16)); }
    template <typename T> T* as() { return (T*)p; }
};
}  // namespace

extern "C" int wepp_fitch_sites(const wepp_tree_desc* tree, int device, uint32_t n_sites, const uint8_t* site_ref,
                                const uint32_t* var_off, const uint32_t* var_node, const uint8_t* var_nuc,
                                uint64_t capacity, uint64_t* n_out, uint32_t* out_site, uint32_t* out_node,
                                uint8_t* out_par, uint8_t* out_mut) {
    if (!tree || !n_out || !var_off || (n_sites && !site_ref)) return set_error(WEPP_EINVAL, "null argument");
    *n_out = 0;
    if (n_sites == 0) return WEPP_OK;
    // topology only: the mutation lists of `tree` are ignored (a new MAT is being built)
    std::vector<uint32_t> zero_off((size_t)tree->n_nodes + 1, 0);
    wepp_tree_desc topo = *tree;
    topo.mut_off = zero_off.data();
    topo.mut_pos = nullptr; topo.mut_ref = nullptr; topo.mut_par = nullptr; topo.mut_mut = nullptr;
    FlatMAT f;
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
    std::vector<uint32_t> meta(N), id2dfs(N), depth(N, 0), nchild(N, 0);
    uint32_t max_children = 0;
    for (uint32_t d = 1; d < N; d++) max_children = std::max(max_children, ++nchild[f.parent_dfs[d]]);
    // the set form of the forward pass needs non-empty allele sets (checked per row below) and 15-bit counters
    bool sets_ok = max_children <= FITCH_SETS_MAX_CHILDREN;
    if (const char* env = std::getenv("WEPP_FITCH_SCORES"))      // test hook: force the score form
        if (env[0] == '1') sets_ok = false;
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
    std::vector<uint32_t> chunk_start(C + 1), chunk_depth(C + 1), chunk_min(C), chunk_open((size_t)(C + 1) * D, 0);
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
    // rows: reference base index, tree samples sorted by node index (BFS for the level-synchronous
    // form, DFS for the two stack forms)
    // the CSR over the rows is checked before anything is sized from it or written through it
    if (var_off[0] != 0) return set_error(WEPP_EINVAL, "var_off[0] must be 0");
    for (uint32_t s = 0; s < n_sites; s++)
        if (var_off[s + 1] < var_off[s]) return set_error(WEPP_EINVAL, "var_off not monotone");
    const uint64_t nv = var_off[n_sites];
    if (nv && (!var_node || !var_nuc)) return set_error(WEPP_EINVAL, "null variant arrays");
    for (uint64_t k = 0; k < nv; k++)
        if ((var_nuc[k] & 15) == 0) sets_ok = false;     // no base allowed: the scores leave the set forms' range
    bool levels = sets_ok;
    if (const char* env = std::getenv("WEPP_FITCH_DFS"))          // test hook: force the DFS stack forms
        if (env[0] == '1') levels = false;
    std::vector<uint8_t> ref_idx(n_sites), vnuc(nv);
    std::vector<uint32_t> vdfs(nv);
    // rows are independent: host threads share them (one thread per row range; a 30 K-row VCF over 1 M
    // nodes spent 0.4 s here on one thread, most of it sorting)
    {
        const uint32_t nthr = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)std::thread::hardware_concurrency(), 16ull, nv / 200000 + 1, (uint64_t)n_sites}));
        std::vector<int> rcs(nthr, WEPP_OK);
        std::vector<std::string> msgs(nthr);
        auto work = [&](uint32_t t) {
            std::vector<std::pair<uint32_t, uint8_t>> row;
            const uint32_t s0 = (uint32_t)((uint64_t)n_sites * t / nthr), s1 = (uint32_t)((uint64_t)n_sites * (t + 1) / nthr);
            for (uint32_t s = s0; s < s1; s++) {
                const uint8_t r = site_ref[s] & 15;
                if (r == 0 || (r & (r - 1))) {
                    rcs[t] = WEPP_EINVAL;
                    msgs[t] = "site_ref must be a single nucleotide (row " + std::to_string(s) + ")";
                    return;
                }
                ref_idx[s] = (uint8_t)__builtin_ctz(r);
                if (var_off[s + 1] < var_off[s]) { rcs[t] = WEPP_EINVAL; msgs[t] = "var_off not monotone"; return; }
                row.clear();
                bool sorted = true;
                for (uint32_t k = var_off[s]; k < var_off[s + 1]; k++) {
                    if (var_node[k] >= N) { rcs[t] = WEPP_EINVAL; msgs[t] = "var_node out of range"; return; }
                    const uint32_t dd = id2dfs[var_node[k]];
                    const uint32_t key = levels ? f.dfs2bfs[dd] : dd;
                    if (!row.empty() && key <= row.back().first) sorted = false;
                    row.emplace_back(key, var_nuc[k]);
                }
                // a node named twice in a row: the later entry wins, as the later assignment does at usher_mapper.cpp:57-62
                if (!sorted)
                    std::stable_sort(row.begin(), row.end(), [](const std::pair<uint32_t, uint8_t>& a,
                                                                const std::pair<uint32_t, uint8_t>& b) { return a.first < b.first; });
                uint32_t w = var_off[s];
                for (size_t i = 0; i < row.size(); i++) {
                    if (i + 1 < row.size() && row[i + 1].first == row[i].first) continue;
                    vdfs[w] = row[i].first;
                    vnuc[w] = row[i].second;
                    w++;
                }
                for (; w < var_off[s + 1]; w++) { vdfs[w] = 0xFFFFFFFFu; vnuc[w] = 0; }   // dropped duplicates
            }
        };
        if (nthr == 1) work(0);
        else {
            std::vector<std::thread> pool;
            for (uint32_t t = 0; t < nthr; t++) pool.emplace_back(work, t);
            for (auto& th : pool) th.join();
        }
        for (uint32_t t = 0; t < nthr; t++)
            if (rcs[t] != WEPP_OK) return set_error(rcs[t], msgs[t]);
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return set_error(WEPP_EDEVICE, "no HIP device available (the Fitch-Sankoff pass has no CPU fallback)");
    if (device < 0 || device >= ndev) return set_error(WEPP_EINVAL, "device index out of range");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hipf(e, "hipSetDevice");

    DevBuf d_meta, d_ref, d_voff, d_vdfs, d_vnuc, d_tables, d_count, d_out, d_cs, d_cd, d_cm, d_co, d_inh, d_outp;
    DevBuf d_lcoff, d_lpar;
    // level-synchronous form: the topology in BFS order (levels and sibling groups are contiguous)
    std::vector<uint32_t> level_off, l_coff, l_par;
    if (levels) {
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
    const size_t budget = free_b / 2;
    const uint32_t group = (uint32_t)std::max<size_t>(1, std::min<size_t>(nbatches, budget / std::max<size_t>(per_batch, 1)));
    if ((e = d_meta.alloc((size_t)N * 4)) != hipSuccess || (e = d_ref.alloc(n_sites)) != hipSuccess ||
        (e = d_voff.alloc((size_t)(n_sites + 1) * 4)) != hipSuccess || (e = d_vdfs.alloc(nv * 4)) != hipSuccess ||
        (e = d_vnuc.alloc(nv)) != hipSuccess || (e = d_tables.alloc((size_t)N * rows_per_batch * group)) != hipSuccess ||
        (e = d_inh.alloc(part_bytes * group)) != hipSuccess || (e = d_outp.alloc(part_bytes * group)) != hipSuccess ||
        (e = d_cs.alloc((C + 1) * 4)) != hipSuccess || (e = d_cd.alloc((C + 1) * 4)) != hipSuccess ||
        (e = d_cm.alloc(C * 4)) != hipSuccess || (e = d_co.alloc(chunk_open.size() * 4)) != hipSuccess ||
        (e = d_count.alloc(8)) != hipSuccess || (e = d_out.alloc(std::max<uint64_t>(capacity, 1) * 8)) != hipSuccess ||
        (e = d_lcoff.alloc(l_coff.size() * 4)) != hipSuccess || (e = d_lpar.alloc(l_par.size() * 4)) != hipSuccess)
        return set_error(WEPP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemcpy(d_meta.p, meta.data(), (size_t)N * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_ref.p, ref_idx.data(), n_sites, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_voff.p, var_off, (size_t)(n_sites + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && nv) e = hipMemcpy(d_vdfs.p, vdfs.data(), nv * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && nv) e = hipMemcpy(d_vnuc.p, vnuc.data(), nv, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cs.p, chunk_start.data(), (C + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cd.p, chunk_depth.data(), (C + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cm.p, chunk_min.data(), C * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_co.p, chunk_open.data(), chunk_open.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_count.p, 0, 8);
    if (e == hipSuccess && levels) e = hipMemcpy(d_lcoff.p, l_coff.data(), l_coff.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && levels) e = hipMemcpy(d_lpar.p, l_par.data(), l_par.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return hipf(e, "upload");
    FitchLevels fl{N, levels ? (uint32_t)level_off.size() - 1 : 0, d_lcoff.as<uint32_t>(), d_lpar.as<uint32_t>()};
    FitchTree ft{N, f.max_depth, C, d_meta.as<uint32_t>(), d_cs.as<uint32_t>(), d_cd.as<uint32_t>(),
                 d_cm.as<uint32_t>(), d_co.as<uint32_t>()};
    FitchSites fs{n_sites, d_ref.as<uint8_t>(), d_voff.as<uint32_t>(), d_vdfs.as<uint32_t>(), d_vnuc.as<uint8_t>()};
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
        DevBuf d_b2i, d_pm;
        uint32_t* d_site = d_out.as<uint32_t>();
        uint32_t* d_node = d_site + cnt;
        if (e == hipSuccess && ((e = d_b2i.alloc((size_t)N * 4)) != hipSuccess || (e = d_pm.alloc(cnt * 2)) != hipSuccess))
            return set_error(WEPP_ENOMEM, std::string("hipMalloc (decoding the mutations): ") + hipGetErrorString(e));
        if (e == hipSuccess) e = hipMemcpy(d_b2i.p, f.bfs2id.data(), (size_t)N * 4, hipMemcpyHostToDevice);
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
