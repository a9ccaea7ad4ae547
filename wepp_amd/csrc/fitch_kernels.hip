// fitch_kernels.hip -- per-site Fitch-Sankoff (the reference's mapper_body,
// src/usher_mapper.cpp:7-162) on CDNA4.
//
// One wavefront handles 64 VCF rows (lane = site) and walks the tree once in
// DFS pre-order per pass; the control flow (open / close a node) is the same
// for every lane, only the scores differ.
//   forward  (mapper_body :87-112): a node's four scores are complete when its
//            subtree closes; its contribution min(s[j], min_k s[k] + 1) is added
//            to the parent's accumulator (top of an LDS stack, one int4 row per
//            depth).  What the backward pass needs from a node is only its
//            decision table  state(parent_state)  (:130-143: the parent's state if
//            it attains the minimum, else the lowest minimal base): 4 x 2 bits,
//            written as one byte per (node, site) -- 64 B per node per wave,
//            coalesced.
//   backward (:115-157): pre-order again, state = table[state(parent)]; a
//            mutation is emitted when it differs.
// HBM-bound integer streaming: 1 byte per (node, site) written, then read.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fitch.hpp"

namespace wepp {

namespace {

struct I4 { int a[4]; };

__device__ __forceinline__ uint32_t decision_table(const I4& s) {
    int mn = min(min(s.a[0], s.a[1]), min(s.a[2], s.a[3]));
    uint32_t first = s.a[0] == mn ? 0u : (s.a[1] == mn ? 1u : (s.a[2] == mn ? 2u : 3u));
    uint32_t t = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) t |= ((s.a[p] == mn) ? (uint32_t)p : first) << (2 * p);
    return t;
}

}  // namespace

// meta[d] = depth | leaf << 31 (DFS pre-order).  stack rows live in dynamic LDS:
// (max_depth + 1) rows of 64 int4.
__global__ __launch_bounds__(64) void k_fitch_forward(FitchTree t, FitchSites s, uint32_t batch0,
                                                      uint8_t* __restrict__ tables) {
    extern __shared__ int lds_i[];
    I4* stack = reinterpret_cast<I4*>(lds_i);           // [depth][64]
    const uint32_t lane = threadIdx.x;
    const uint32_t batch = batch0 + blockIdx.x;
    const uint32_t site = batch * 64 + lane;
    const bool have = site < s.n_sites;
    const int big = (int)t.N;                             // "num_nodes" penalty, :42,58
    const uint32_t ref = have ? s.ref_idx[site] : 0;
    uint32_t vp = have ? s.var_off[site] : 0;
    const uint32_t vend = have ? s.var_off[site + 1] : 0;
    uint32_t vnext = vp < vend ? s.var_dfs[vp] : 0xFFFFFFFFu;
    uint8_t* tbl = tables + (size_t)blockIdx.x * t.N * 64;

    I4 cur;                                               // accumulator of the deepest open node
    uint32_t sp = 0;                                      // open nodes (uniform)
    uint32_t open_node = 0;                               // deepest open node (uniform)
    // the node of every open depth, for the table write at close
    uint32_t* open_ids = reinterpret_cast<uint32_t*>(stack + (size_t)(t.max_depth + 1) * 64);

    auto close_top = [&]() {
        // scores of open_node are final: table out, contribution into the parent
        const uint32_t tb = decision_table(cur);
        tbl[(size_t)open_node * 64 + lane] = (uint8_t)tb;
        const int mn = min(min(cur.a[0], cur.a[1]), min(cur.a[2], cur.a[3]));
        sp--;
        if (sp > 0) {
            I4 par = stack[(size_t)(sp - 1) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; j++) par.a[j] += min(min(cur.a[j], mn + 1), big + 1);   // :96-108
            cur = par;
            open_node = open_ids[sp - 1];
        }
    };

    for (uint32_t d0 = 0; d0 < t.N; d0 += 64) {
        const uint32_t mv = (d0 + lane < t.N) ? t.meta[d0 + lane] : 0;
        const uint32_t cntn = min(64u, t.N - d0);
        for (uint32_t i = 0; i < cntn; i++) {
            const uint32_t mt = (uint32_t)__builtin_amdgcn_readlane((int)mv, (int)i);
            const uint32_t dep = mt & 0x7FFFFFFFu;
            const bool leaf = mt >> 31;
            const uint32_t d = d0 + i;
            while (sp > dep) close_top();
            // scores of node d before its children (:26-63)
            I4 nd;
#pragma unroll
            for (int j = 0; j < 4; j++) nd.a[j] = (leaf && (uint32_t)j != ref) ? big : 0;
            if (vnext == d) {
                const uint32_t nuc = s.var_nuc[vp];
#pragma unroll
                for (int j = 0; j < 4; j++) nd.a[j] = ((nuc >> j) & 1u) ? 0 : big;
                vp++;
                vnext = vp < vend ? s.var_dfs[vp] : 0xFFFFFFFFu;
            }
            if (leaf && dep > 0) {
                // a leaf is final at once: table out, contribution straight into the parent's
                // accumulator, which stays in registers (no stack traffic for ~half of the nodes)
                tbl[(size_t)d * 64 + lane] = (uint8_t)decision_table(nd);
                const int mn = min(min(nd.a[0], nd.a[1]), min(nd.a[2], nd.a[3]));
#pragma unroll
                for (int j = 0; j < 4; j++) cur.a[j] += min(min(nd.a[j], mn + 1), big + 1);
                continue;
            }
            if (sp > 0) stack[(size_t)(sp - 1) * 64 + lane] = cur;      // park the parent
            cur = nd;
            if (lane == 0) open_ids[sp] = d;
            open_node = d;
            sp = dep + 1;
        }
    }
    while (sp > 0) close_top();
}

__global__ __launch_bounds__(64) void k_fitch_backward(FitchTree t, FitchSites s, uint32_t batch0,
                                                       const uint8_t* __restrict__ tables,
                                                       unsigned long long* __restrict__ out_count,
                                                       uint64_t capacity, uint2* __restrict__ out) {
    extern __shared__ int lds_i[];
    uint8_t* states = reinterpret_cast<uint8_t*>(lds_i);  // [depth][64]
    const uint32_t lane = threadIdx.x;
    const uint32_t batch = batch0 + blockIdx.x;
    const uint32_t site = batch * 64 + lane;
    const bool have = site < s.n_sites;
    const uint32_t ref = have ? s.ref_idx[site] : 0;
    const uint8_t* tbl = tables + (size_t)blockIdx.x * t.N * 64;
    for (uint32_t d0 = 0; d0 < t.N; d0 += 64) {
        const uint32_t mv = (d0 + lane < t.N) ? t.meta[d0 + lane] : 0;
        const uint32_t cntn = min(64u, t.N - d0);
        for (uint32_t i = 0; i < cntn; i++) {
            const uint32_t dep = (uint32_t)__builtin_amdgcn_readlane((int)mv, (int)i) & 0x7FFFFFFFu;
            const uint32_t d = d0 + i;
            const uint32_t par_state = dep ? states[(size_t)(dep - 1) * 64 + lane] : ref;     // :119-128
            const uint32_t tb = tbl[(size_t)d * 64 + lane];
            const uint32_t state = (tb >> (2 * par_state)) & 3u;
            states[(size_t)dep * 64 + lane] = (uint8_t)state;
            if (have && state != par_state) {                                                  // :145-156
                const unsigned long long slot = atomicAdd(out_count, 1ull);
                if (slot < capacity) out[slot] = make_uint2(site, d | (par_state << 28) | (state << 30));
            }
        }
    }
}

hipError_t launch_fitch_forward(const FitchTree& t, const FitchSites& s, uint32_t batch0, uint32_t nbatches,
                                uint8_t* tables, hipStream_t stream) {
    const uint32_t lds = (t.max_depth + 1) * 64 * 16 + (t.max_depth + 2) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)k_fitch_forward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fitch_forward, dim3(nbatches), dim3(64), lds, stream, t, s, batch0, tables);
    return hipGetLastError();
}

hipError_t launch_fitch_backward(const FitchTree& t, const FitchSites& s, uint32_t batch0, uint32_t nbatches,
                                 const uint8_t* tables, unsigned long long* out_count, uint64_t capacity, uint2* out,
                                 hipStream_t stream) {
    const uint32_t lds = (t.max_depth + 1) * 64;
    hipLaunchKernelGGL(k_fitch_backward, dim3(nbatches), dim3(64), lds, stream, t, s, batch0, tables, out_count,
                       capacity, out);
    return hipGetLastError();
}

}  // namespace wepp
