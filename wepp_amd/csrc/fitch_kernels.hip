// fitch_kernels.hip -- per-site Fitch-Sankoff (the reference's mapper_body,
// src/usher_mapper.cpp:7-162) on CDNA4.
//
// One wavefront handles 64 VCF rows (lane = site) and a CHUNK of the tree in DFS
// pre-order; the control flow (open / close a node) is the same for every lane,
// only the scores differ.
//   forward  (mapper_body :87-112): a node's four scores are complete when its
//            subtree closes; its contribution min(s[j], min_k s[k] + 1) is added to
//            the parent's accumulator (top of an LDS stack, one int4 row per depth).
//            What the backward pass needs from a node is only its decision table
//            state(parent_state) (:130-143: the parent's state if it attains the
//            minimum, else the lowest minimal base): 4 x 2 bits, one byte per
//            (node, site) -- 64 B per node per wave, coalesced.
//            Nodes whose subtree crosses a chunk boundary cannot be finished by one
//            wave: k_fitch_forward leaves their partial sums in global memory
//            (inh = contributions to nodes already open when the chunk starts,
//            out = accumulators of nodes the chunk opens and leaves open) and
//            k_fitch_stitch -- one wave per 64 rows, O(chunks x depth) work --
//            closes them in order.
//   backward (:115-157): pre-order again, state = table[state(parent)]; a chunk
//            first replays the path above its first node; a mutation is emitted
//            when the state differs from the parent's.
// HBM-bound integer streaming: 1 byte per (node, site) written, then read.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fitch.hpp"

namespace wepp {

namespace {

struct I4 { int a[4]; };

constexpr uint32_t FITCH_BACK_UNROLL = 8;   // decision bytes loaded ahead in the backward pass
constexpr uint32_t FITCH_QUEUE = 256;       // mutations queued per wave before one atomic reserves their slots

__device__ __forceinline__ uint32_t decision_table(const I4& s) {
    int mn = min(min(s.a[0], s.a[1]), min(s.a[2], s.a[3]));
    uint32_t first = s.a[0] == mn ? 0u : (s.a[1] == mn ? 1u : (s.a[2] == mn ? 2u : 3u));
    uint32_t t = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) t |= ((s.a[p] == mn) ? (uint32_t)p : first) << (2 * p);
    return t;
}

__device__ __forceinline__ void add_contribution(I4& par, const I4& child, int big) {   // :96-108
    const int mn = min(min(child.a[0], child.a[1]), min(child.a[2], child.a[3]));
#pragma unroll
    for (int j = 0; j < 4; j++) par.a[j] += min(min(child.a[j], mn + 1), big + 1);
}

// state of a node given its byte and the parent's state (:115-143)
template <bool MASKS>
__device__ __forceinline__ uint32_t next_state(uint32_t tb, uint32_t ps) {
    if (!MASKS) return (tb >> (2 * ps)) & 3u;
    // keep the parent's state if it is optimal, else the lowest optimal base
    return ((tb >> ps) & 1u) ? (uint32_t)__builtin_ctz(~tb & 15u) : ps;
}

}  // namespace

// grid = nbatches * C, one wave per (batch, chunk).
__global__ __launch_bounds__(64) void k_fitch_forward(FitchTree t, FitchSites s, uint32_t batch0,
                                                      uint8_t* __restrict__ tables, int4* __restrict__ inh_part,
                                                      int4* __restrict__ out_part) {
    extern __shared__ int lds_i[];
    I4* stack = reinterpret_cast<I4*>(lds_i);                                  // [depth][64]
    uint32_t* open_ids = reinterpret_cast<uint32_t*>(stack + (size_t)(t.max_depth + 1) * 64);
    const uint32_t lane = threadIdx.x;
    const uint32_t bl = blockIdx.x / t.C;                                       // batch within the group
    const uint32_t ch = blockIdx.x % t.C;
    const uint32_t site = (batch0 + bl) * 64 + lane;
    const bool have = site < s.n_sites;
    const int big = (int)t.N;                                                   // "num_nodes" penalty, :42,58
    const uint32_t ref = have ? s.ref_idx[site] : 0;
    const uint32_t a = t.chunk_start[ch], b = t.chunk_start[ch + 1];
    const uint32_t D = t.max_depth + 1;
    uint8_t* tbl = tables + (size_t)bl * t.N * 64;
    I4* inh = reinterpret_cast<I4*>(inh_part) + ((size_t)bl * t.C + ch) * D * 64;
    I4* outp = reinterpret_cast<I4*>(out_part) + ((size_t)bl * t.C + ch) * D * 64;

    // variants of this row inside the chunk: first one with DFS index >= a
    uint32_t vp = have ? s.var_off[site] : 0;
    const uint32_t vend = have ? s.var_off[site + 1] : 0;
    {
        uint32_t lo = vp, hi = vend;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s.var_dfs[mid] < a) lo = mid + 1; else hi = mid;
        }
        vp = lo;
    }
    // two-deep queue of this row's next variants: the load that refills it has a whole
    // inter-variant gap to land
    uint32_t vnext = vp < vend ? s.var_dfs[vp] : 0xFFFFFFFFu;
    uint32_t vnuc = vp < vend ? s.var_nuc[vp] : 0;
    uint32_t vnext2 = vp + 1 < vend ? s.var_dfs[vp + 1] : 0xFFFFFFFFu;
    uint32_t vnuc2 = vp + 1 < vend ? s.var_nuc[vp + 1] : 0;

    // nodes open when the chunk starts: their partial sums start from zero here
    const uint32_t dep_a = t.chunk_depth[ch];
    uint32_t inh_top = dep_a;                       // inherited levels still open (uniform)
    I4 zero = {{0, 0, 0, 0}};
    for (uint32_t k = 0; k < dep_a; k++) stack[(size_t)k * 64 + lane] = zero;
    I4 cur = zero;                                  // accumulator of the deepest open node
    uint32_t sp = dep_a;                            // open nodes (uniform)
    uint32_t open_node = 0;

    auto close_top = [&]() {
        const uint32_t k = sp - 1;
        if (k < inh_top) {
            // a node opened by an earlier chunk closes here: only its partial sum is known
            inh[(size_t)k * 64 + lane] = cur;
            inh_top = k;
            sp--;
            if (sp > 0) cur = stack[(size_t)(sp - 1) * 64 + lane];
            return;
        }
        tbl[(size_t)open_node * 64 + lane] = (uint8_t)decision_table(cur);
        sp--;
        if (sp > 0) {
            I4 par = stack[(size_t)(sp - 1) * 64 + lane];
            add_contribution(par, cur, big);
            cur = par;
            if (sp - 1 >= inh_top) open_node = open_ids[sp - 1];
        }
    };

    for (uint32_t d0 = a; d0 < b; d0 += 64) {
        const uint32_t mv = (d0 + lane < b) ? t.meta[d0 + lane] : 0;
        // wait for the 64 meta words HERE: left to the compiler the wait sits in front of the readlane
        // of every node, where vmcnt(0) also drains the table store of the previous node (~500 cycles)
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), lgkmcnt/expcnt untouched
        const uint32_t cntn = min(64u, b - d0);
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
#pragma unroll
                for (int j = 0; j < 4; j++) nd.a[j] = ((vnuc >> j) & 1u) ? 0 : big;
                vp++;
                vnext = vnext2;
                vnuc = vnuc2;
                vnext2 = vp + 1 < vend ? s.var_dfs[vp + 1] : 0xFFFFFFFFu;
                vnuc2 = vp + 1 < vend ? s.var_nuc[vp + 1] : 0;
            }
            if (leaf && dep > 0) {
                // a leaf is final at once: table out, contribution straight into the parent's
                // accumulator, which stays in registers (no stack traffic for ~half of the nodes)
                tbl[(size_t)d * 64 + lane] = (uint8_t)decision_table(nd);
                add_contribution(cur, nd, big);
                continue;
            }
            if (sp > 0) stack[(size_t)(sp - 1) * 64 + lane] = cur;      // park the parent
            cur = nd;
            if (lane == 0) open_ids[sp] = d;
            open_node = d;
            sp = dep + 1;
        }
    }
    // subtrees that end exactly at the chunk boundary are complete: close them as the first
    // node of the next chunk would (only its strict ancestors stay open)
    {
        const uint32_t dep_next = t.chunk_depth[ch + 1];
        while (sp > dep_next) close_top();
    }
    // what is still open belongs to the stitch pass: partial sums of inherited nodes,
    // accumulators of the nodes this chunk opened
    if (sp > 0) stack[(size_t)(sp - 1) * 64 + lane] = cur;
    for (uint32_t k = 0; k < sp; k++) {
        const I4 v = stack[(size_t)k * 64 + lane];
        if (k < inh_top) inh[(size_t)k * 64 + lane] = v;
        else outp[(size_t)k * 64 + lane] = v;
    }
}

// grid = nbatches, one wave per batch: closes the nodes that span chunk boundaries.
// chunk_open[c * D + k] = node open at level k when chunk c starts (c = C: after the
// last node); chunk_min[c] = smallest depth of a node of chunk c.
__global__ __launch_bounds__(64) void k_fitch_stitch(FitchTree t, FitchSites s, uint32_t batch0,
                                                     uint8_t* __restrict__ tables, const int4* __restrict__ inh_part,
                                                     const int4* __restrict__ out_part) {
    extern __shared__ int lds_i[];
    I4* acc = reinterpret_cast<I4*>(lds_i);                                    // [depth][64]
    const uint32_t lane = threadIdx.x;
    const uint32_t bl = blockIdx.x;
    const int big = (int)t.N;
    const uint32_t D = t.max_depth + 1;
    uint8_t* tbl = tables + (size_t)bl * t.N * 64;
    for (uint32_t c = 0; c <= t.C; c++) {
        const uint32_t dep_a = t.chunk_depth[c];                 // nodes open when chunk c starts
        // levels >= lo close inside chunk c: when a node of that depth opens, or at the chunk's
        // end because the next chunk starts shallower
        const uint32_t lo = (c < t.C) ? min(min(t.chunk_min[c], dep_a), t.chunk_depth[c + 1]) : 0;
        if (c < t.C) {
            const I4* inh = reinterpret_cast<const I4*>(inh_part) + ((size_t)bl * t.C + c) * D * 64;
            for (uint32_t k = 0; k < dep_a; k++) {
                I4 v = acc[(size_t)k * 64 + lane];
                const I4 p = inh[(size_t)k * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; j++) v.a[j] += p.a[j];
                acc[(size_t)k * 64 + lane] = v;
            }
        }
        for (uint32_t k = dep_a; k-- > lo;) {                    // deepest first
            const I4 v = acc[(size_t)k * 64 + lane];
            tbl[(size_t)t.chunk_open[(size_t)c * D + k] * 64 + lane] = (uint8_t)decision_table(v);
            if (k > 0) {
                I4 par = acc[(size_t)(k - 1) * 64 + lane];
                add_contribution(par, v, big);
                acc[(size_t)(k - 1) * 64 + lane] = par;
            }
        }
        if (c < t.C) {
            const I4* outp = reinterpret_cast<const I4*>(out_part) + ((size_t)bl * t.C + c) * D * 64;
            const uint32_t dep_b = t.chunk_depth[c + 1];
            for (uint32_t k = lo; k < dep_b; k++) acc[(size_t)k * 64 + lane] = outp[(size_t)k * 64 + lane];
        }
    }
}

// -----------------------------------------------------------------------------
// Set form of the forward pass.  With unit substitution costs the four scores of a node are
//   s[j] = K + N * [j not allowed] + e[j],   e[j] = number of children for which j is not optimal,
// (K the same for all j; "allowed" = the node's observed alleles, all four when it has none): a
// child contributes min(s[j], min_k s[k] + 1) = min + [j not optimal] (:96-108), so only its
// optimal SET travels up, and the decision table (:130-143) is "keep the parent's state if it is
// optimal, else the lowest optimal base".  Valid while every observed allele set is non-empty
// (then every minimum stays below N and the clamp at N + 1 of :98 never binds) -- the host checks
// that, and that no node has more than 32767 children, and falls back to the score form otherwise.
// A node's state is four 16-bit counters in two registers, x[j] = e[j] + 0x8000 * [j not allowed]:
// the minimum over the packed fields is the minimum over the allowed bases, `x - min` clamped to 1
// is the packed contribution to the parent, and the byte kept for the backward pass is the
// 4-bit "not optimal" mask.  ~20 instructions per node instead of ~100, half the LDS stack.
// -----------------------------------------------------------------------------
namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
struct X2 { uint32_t a, b; };                      // fields 0,1 | 2,3

__device__ __forceinline__ uint32_t pk_min(uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, y)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, x) - __builtin_bit_cast(u16x2, y));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, x) + __builtin_bit_cast(u16x2, y));
}
// counters of a node before its children: 0, with the "not allowed" flag where the mask says so
__device__ __forceinline__ X2 x_init(uint32_t allowed) {
    const uint32_t n = ~allowed;
    return {((n & 1u) << 15) | ((n & 2u) << 30), ((n & 4u) << 13) | ((n & 8u) << 28)};
}
// packed [j not in mask] for the four bases
__device__ __forceinline__ X2 x_delta_of_notopt(uint32_t n) {
    return {(n & 1u) | ((n & 2u) << 15), ((n >> 2) & 1u) | ((n & 8u) << 13)};
}
// closes a node: its "not optimal" mask and its contribution to the parent
__device__ __forceinline__ uint32_t x_close(const X2& x, X2& delta) {
    const uint32_t m2 = pk_min(x.a, x.b);
    const uint32_t m = min(m2 & 0xFFFFu, m2 >> 16);
    const uint32_t mm = m | (m << 16);
    delta.a = pk_min(pk_sub(x.a, mm), 0x00010001u);
    delta.b = pk_min(pk_sub(x.b, mm), 0x00010001u);
    const uint32_t t = delta.a | (delta.b << 2);
    return (t | (t >> 15)) & 15u;
}

}  // namespace

// grid = nbatches * C, one wave per (batch, chunk); same chunking and scratch roles as k_fitch_forward
__global__ __launch_bounds__(64) void k_fitch_forward_sets(FitchTree t, FitchSites s, uint32_t batch0,
                                                           uint8_t* __restrict__ tables, uint2* __restrict__ inh_part,
                                                           uint2* __restrict__ out_part) {
    extern __shared__ int lds_i[];
    X2* stack = reinterpret_cast<X2*>(lds_i);                                  // [depth][64]
    uint32_t* open_ids = reinterpret_cast<uint32_t*>(stack + (size_t)(t.max_depth + 1) * 64);
    const uint32_t lane = threadIdx.x;
    const uint32_t bl = blockIdx.x / t.C;
    const uint32_t ch = blockIdx.x % t.C;
    const uint32_t site = (batch0 + bl) * 64 + lane;
    const bool have = site < s.n_sites;
    const uint32_t ref = have ? s.ref_idx[site] : 0;
    const uint32_t a = t.chunk_start[ch], b = t.chunk_start[ch + 1];
    const uint32_t D = t.max_depth + 1;
    uint8_t* tbl = tables + (size_t)bl * t.N * 64;
    X2* inh = reinterpret_cast<X2*>(inh_part) + ((size_t)bl * t.C + ch) * D * 64;
    X2* outp = reinterpret_cast<X2*>(out_part) + ((size_t)bl * t.C + ch) * D * 64;

    uint32_t vp = have ? s.var_off[site] : 0;
    const uint32_t vend = have ? s.var_off[site + 1] : 0;
    {
        uint32_t lo = vp, hi = vend;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s.var_dfs[mid] < a) lo = mid + 1; else hi = mid;
        }
        vp = lo;
    }
    uint32_t vnext = vp < vend ? s.var_dfs[vp] : 0xFFFFFFFFu;
    uint32_t vnuc = vp < vend ? s.var_nuc[vp] : 0;
    uint32_t vnext2 = vp + 1 < vend ? s.var_dfs[vp + 1] : 0xFFFFFFFFu;
    uint32_t vnuc2 = vp + 1 < vend ? s.var_nuc[vp + 1] : 0;

    // a leaf without an observation allows the reference base only (:36-45)
    const uint32_t leaf_notopt = ~(1u << ref) & 15u;
    const X2 leaf_delta = x_delta_of_notopt(leaf_notopt);

    // Open nodes: level sp-1 in `cur`, level sp-2 in `par` (registers), levels below in the LDS stack.
    // A close needs the parent at once; with it in a register the LDS read that refills `par`
    // has until the NEXT close to land, and a value popped from the stack is not written back
    // (par_dirty = the stack slot of `par` is stale).
    const uint32_t dep_a = t.chunk_depth[ch];
    uint32_t inh_top = dep_a;
    const X2 zero = {0u, 0u};
    for (uint32_t k = 0; k < dep_a; k++) stack[(size_t)k * 64 + lane] = zero;
    X2 cur = zero, par = zero;
    bool par_dirty = false;
    uint32_t sp = dep_a;
    uint32_t open_node = 0;

    auto pop_par = [&]() {                   // after sp was decremented: refill par = level sp-2
        par_dirty = false;
        if (sp > 1) par = stack[(size_t)(sp - 2) * 64 + lane];
    };
    auto close_top = [&]() {
        const uint32_t k = sp - 1;
        if (k < inh_top) {
            inh[(size_t)k * 64 + lane] = cur;
            inh_top = k;
            sp--;
            cur = par;
            pop_par();
            return;
        }
        X2 dl;
        tbl[(size_t)open_node * 64 + lane] = (uint8_t)x_close(cur, dl);
        sp--;
        if (sp > 0) {
            cur.a = pk_add(par.a, dl.a);
            cur.b = pk_add(par.b, dl.b);
            if (sp - 1 >= inh_top) open_node = open_ids[sp - 1];
            pop_par();
        }
    };

    for (uint32_t d0 = a; d0 < b; d0 += 64) {
        const uint32_t mv = (d0 + lane < b) ? t.meta[d0 + lane] : 0;
        // wait for the 64 meta words HERE: left to the compiler the wait sits in front of the readlane
        // of every node, where vmcnt(0) also drains the table store of the previous node (~500 cycles)
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), lgkmcnt/expcnt untouched
        const uint32_t cntn = min(64u, b - d0);
        for (uint32_t i = 0; i < cntn; i++) {
            const uint32_t mt = (uint32_t)__builtin_amdgcn_readlane((int)mv, (int)i);
            const uint32_t dep = mt & 0x7FFFFFFFu;
            const bool leaf = mt >> 31;
            const uint32_t d = d0 + i;
            while (sp > dep) close_top();
            const bool is_var = vnext == d;
            const uint32_t allowed = vnuc & 15u;
            if (is_var) {
                vp++;
                vnext = vnext2;
                vnuc = vnuc2;
                vnext2 = vp + 1 < vend ? s.var_dfs[vp + 1] : 0xFFFFFFFFu;
                vnuc2 = vp + 1 < vend ? s.var_nuc[vp + 1] : 0;
            }
            if (leaf && dep > 0) {
                // a leaf is final at once: its allowed set is its optimal set
                const uint32_t no = is_var ? (~allowed & 15u) : leaf_notopt;
                const X2 dl = is_var ? x_delta_of_notopt(no) : leaf_delta;
                tbl[(size_t)d * 64 + lane] = (uint8_t)no;
                cur.a = pk_add(cur.a, dl.a);
                cur.b = pk_add(cur.b, dl.b);
                continue;
            }
            // open node d: the old parent goes to the stack if its slot is stale, cur becomes the parent
            if (sp > 1 && par_dirty) stack[(size_t)(sp - 2) * 64 + lane] = par;
            if (sp > 0) { par = cur; par_dirty = true; }
            cur = is_var ? x_init(allowed) : (leaf ? x_init(1u << ref) : zero);
            if (lane == 0) open_ids[sp] = d;
            open_node = d;
            sp = dep + 1;
        }
    }
    {
        const uint32_t dep_next = t.chunk_depth[ch + 1];
        while (sp > dep_next) close_top();
    }
    if (sp > 0) stack[(size_t)(sp - 1) * 64 + lane] = cur;
    if (sp > 1) stack[(size_t)(sp - 2) * 64 + lane] = par;
    for (uint32_t k = 0; k < sp; k++) {
        const X2 v = stack[(size_t)k * 64 + lane];
        if (k < inh_top) inh[(size_t)k * 64 + lane] = v;
        else outp[(size_t)k * 64 + lane] = v;
    }
}

__global__ __launch_bounds__(64) void k_fitch_stitch_sets(FitchTree t, FitchSites s, uint32_t batch0,
                                                          uint8_t* __restrict__ tables, const uint2* __restrict__ inh_part,
                                                          const uint2* __restrict__ out_part) {
    extern __shared__ int lds_i[];
    X2* acc = reinterpret_cast<X2*>(lds_i);                                    // [depth][64]
    const uint32_t lane = threadIdx.x;
    const uint32_t bl = blockIdx.x;
    const uint32_t D = t.max_depth + 1;
    uint8_t* tbl = tables + (size_t)bl * t.N * 64;
    for (uint32_t c = 0; c <= t.C; c++) {
        const uint32_t dep_a = t.chunk_depth[c];
        const uint32_t lo = (c < t.C) ? min(min(t.chunk_min[c], dep_a), t.chunk_depth[c + 1]) : 0;
        if (c < t.C) {
            const X2* inh = reinterpret_cast<const X2*>(inh_part) + ((size_t)bl * t.C + c) * D * 64;
            for (uint32_t k = 0; k < dep_a; k++) {
                X2 v = acc[(size_t)k * 64 + lane];
                const X2 p = inh[(size_t)k * 64 + lane];
                v.a = pk_add(v.a, p.a);
                v.b = pk_add(v.b, p.b);
                acc[(size_t)k * 64 + lane] = v;
            }
        }
        for (uint32_t k = dep_a; k-- > lo;) {                    // deepest first
            const X2 v = acc[(size_t)k * 64 + lane];
            X2 dl;
            tbl[(size_t)t.chunk_open[(size_t)c * D + k] * 64 + lane] = (uint8_t)x_close(v, dl);
            if (k > 0) {
                X2 par = acc[(size_t)(k - 1) * 64 + lane];
                par.a = pk_add(par.a, dl.a);
                par.b = pk_add(par.b, dl.b);
                acc[(size_t)(k - 1) * 64 + lane] = par;
            }
        }
        if (c < t.C) {
            const X2* outp = reinterpret_cast<const X2*>(out_part) + ((size_t)bl * t.C + c) * D * 64;
            const uint32_t dep_b = t.chunk_depth[c + 1];
            for (uint32_t k = lo; k < dep_b; k++) acc[(size_t)k * 64 + lane] = outp[(size_t)k * 64 + lane];
        }
    }
}

// grid = nbatches * C.  MASKS: the byte of a node is its 4-bit "not optimal" mask (set form of the
// forward pass) instead of the decision table.
template <bool MASKS>
__global__ __launch_bounds__(64) void k_fitch_backward(FitchTree t, FitchSites s, uint32_t batch0,
                                                       const uint8_t* __restrict__ tables,
                                                       unsigned long long* __restrict__ out_count,
                                                       uint64_t capacity, uint2* __restrict__ out) {
    extern __shared__ int lds_i[];
    uint8_t* states = reinterpret_cast<uint8_t*>(lds_i);  // [depth][64]
    const uint32_t lane = threadIdx.x;
    const uint32_t bl = blockIdx.x / t.C;
    const uint32_t ch = blockIdx.x % t.C;
    const uint32_t site = (batch0 + bl) * 64 + lane;
    const bool have = site < s.n_sites;
    const uint32_t ref = have ? s.ref_idx[site] : 0;
    const uint8_t* tbl = tables + (size_t)bl * t.N * 64;
    const uint32_t a = t.chunk_start[ch], b = t.chunk_start[ch + 1];
    const uint32_t D = t.max_depth + 1;
    // states of the path above the chunk's first node (:119-143 replayed from the root)
    {
        uint32_t ps = ref;
        const uint32_t dep_a = t.chunk_depth[ch];
        for (uint32_t k = 0; k < dep_a; k++) {
            const uint32_t tb = tbl[(size_t)t.chunk_open[(size_t)ch * D + k] * 64 + lane];
            ps = next_state<MASKS>(tb, ps);
            states[(size_t)k * 64 + lane] = (uint8_t)ps;
        }
    }
    // emitted mutations are queued per wave and flushed with one atomic per ~200 entries
    uint2* queue = reinterpret_cast<uint2*>(states + (size_t)D * 64);
    uint32_t qn = 0;                                        // uniform
    auto flush = [&]() {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(out_count, (unsigned long long)qn);
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        for (uint32_t i = lane; i < qn; i += 64)
            if (base + i < capacity) out[base + i] = queue[i];
        qn = 0;
    };
    for (uint32_t d0 = a; d0 < b; d0 += 64) {
        const uint32_t mv = (d0 + lane < b) ? t.meta[d0 + lane] : 0;
        const uint32_t cntn = min(64u, b - d0);
        for (uint32_t i0 = 0; i0 < cntn; i0 += FITCH_BACK_UNROLL) {
            // decision bytes of the next few nodes in flight together
            uint32_t tbv[FITCH_BACK_UNROLL];
#pragma unroll
            for (uint32_t u = 0; u < FITCH_BACK_UNROLL; u++)
                tbv[u] = tbl[(size_t)min(d0 + i0 + u, b - 1) * 64 + lane];
#pragma unroll
            for (uint32_t u = 0; u < FITCH_BACK_UNROLL; u++) {
                const uint32_t i = i0 + u;
                if (i >= cntn) break;
                const uint32_t dep = (uint32_t)__builtin_amdgcn_readlane((int)mv, (int)i) & 0x7FFFFFFFu;
                const uint32_t d = d0 + i;
                const uint32_t par_state = dep ? states[(size_t)(dep - 1) * 64 + lane] : ref;     // :119-128
                const uint32_t state = next_state<MASKS>(tbv[u], par_state);
                states[(size_t)dep * 64 + lane] = (uint8_t)state;
                const bool emit = have && state != par_state;                                      // :145-156
                const unsigned long long mask = __ballot(emit);
                if (mask) {
                    if (emit) {
                        const uint32_t at = qn + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                        queue[at] = make_uint2(site, d | (par_state << 28) | (state << 30));
                    }
                    qn += (uint32_t)__popcll(mask);
                    if (qn > FITCH_QUEUE - 64) flush();
                }
            }
        }
    }
    if (qn) flush();
}

// -----------------------------------------------------------------------------
// Level-synchronous form (the default).  Nodes in BFS order: a level is a contiguous range, the
// children of a node are contiguous, and byte [node][row] of a batch of 64 rows holds the node's
// "not optimal" mask (up pass) and later its state (down pass).  One launch per tree level, each
// wave takes FITCH_LEVEL_CHUNK consecutive nodes of the level for one batch of rows (lane = row):
//   up   (deepest level first): x = counters of the node's allowed set, + the packed
//        [j not optimal] of every child (64-byte loads, consecutive for consecutive children),
//        close -> mask byte.  No stack, no LDS: the waves of a level are independent.
//   down (root first): state = keep the parent's state if it is optimal, else the lowest optimal
//        base (:130-143); the byte is overwritten with the state; a mutation is queued where the
//        state differs from the parent's (:145-156).
// Same validity range as the set form above (non-empty allele sets, <= 32767 children).
// -----------------------------------------------------------------------------
// Four rows per lane (a batch = 256 VCF rows): the byte row of a node is 256 B, every wave load or
// store moves a whole dword per lane -- 256-byte requests instead of 64-byte ones (one byte per lane
// left the memory system request-bound at ~2.3 TB/s) -- and the wave-uniform part of a node (offsets,
// branches, `readlane`s) is shared by four times as many rows.
constexpr uint32_t FR = FITCH_ROWS_PER_LANE;
static_assert(FR == 4, "one dword of mask / state bytes per lane");

__global__ __launch_bounds__(64 * FITCH_LEVEL_WAVES) void k_fitch_up(FitchLevels t, FitchSites s, uint32_t batch0,
                                                                     uint32_t lev_a, uint32_t lev_b, uint32_t nchunks,
                                                                     uint32_t nunits, uint8_t* __restrict__ bytes) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t unit = blockIdx.x * FITCH_LEVEL_WAVES + wv;
    if (unit >= nunits) return;
    const uint32_t bl = unit / nchunks;
    const uint32_t ch = unit % nchunks;
    const uint32_t a = lev_a + ch * FITCH_LEVEL_CHUNK, b = min(lev_b, a + FITCH_LEVEL_CHUNK);
    uint32_t* by = reinterpret_cast<uint32_t*>(bytes + (size_t)bl * t.N * (64 * FR)) + lane;   // + node * 64: this lane's 4 rows

    // the observations of this lane's rows inside [a, b): sorted by BFS index; two-deep queues, so that
    // the load that refills one has a whole inter-observation gap to land
    uint32_t vp[FR], vend[FR], vnext[FR], vnuc[FR], vnext2[FR], vnuc2[FR], leaf_no[FR];
#pragma unroll
    for (uint32_t q = 0; q < FR; q++) {
        const uint32_t site = ((batch0 + bl) * 64 + lane) * FR + q;
        const bool have = site < s.n_sites;
        leaf_no[q] = ~(1u << (have ? s.ref_idx[site] : 0)) & 15u;     // a leaf without an observation: reference base only (:36-45)
        vp[q] = have ? s.var_off[site] : 0;
        vend[q] = have ? s.var_off[site + 1] : 0;
        uint32_t lo = vp[q], hi = vend[q];
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s.var_dfs[mid] < a) lo = mid + 1; else hi = mid;
        }
        vp[q] = lo;
        vnext[q] = vp[q] < vend[q] ? s.var_dfs[vp[q]] : 0xFFFFFFFFu;
        vnuc[q] = vp[q] < vend[q] ? s.var_nuc[vp[q]] : 0;
        vnext2[q] = vp[q] + 1 < vend[q] ? s.var_dfs[vp[q] + 1] : 0xFFFFFFFFu;
        vnuc2[q] = vp[q] + 1 < vend[q] ? s.var_nuc[vp[q] + 1] : 0;
    }
    // observed allele sets of node d for this lane's rows (nodes are visited in increasing BFS index):
    // bit q of the result = row q observes the node, allowed[q] = its set.  vmin = the next observed
    // node over the lane's four rows: most nodes are observed by no row of the wave, which one compare
    // and a ballot establish (`any` = false, nothing else is touched).
    uint32_t vmin = min(min(vnext[0], vnext[1]), min(vnext[2], vnext[3]));
    auto observed = [&](uint32_t d, uint32_t (&allowed)[FR], bool& any) -> uint32_t {
        any = __ballot(vmin == d) != 0;
        if (!any) return 0;
        __builtin_amdgcn_sched_barrier(0);
        uint32_t is = 0;
#pragma unroll
        for (uint32_t q = 0; q < FR; q++) {
            allowed[q] = vnuc[q] & 15u;
            is |= (vnext[q] == d ? 1u : 0u) << q;
        }
#pragma unroll
        for (uint32_t q = 0; q < FR; q++) {
            if ((is >> q) & 1u) {
                vp[q]++;
                vnext[q] = vnext2[q];
                vnuc[q] = vnuc2[q];
                vnext2[q] = vp[q] + 1 < vend[q] ? s.var_dfs[vp[q] + 1] : 0xFFFFFFFFu;
                vnuc2[q] = vp[q] + 1 < vend[q] ? s.var_nuc[vp[q] + 1] : 0;
            }
        }
        vmin = min(min(vnext[0], vnext[1]), min(vnext[2], vnext[3]));
        return is;
    };
    uint32_t leaf_w = 0;                                       // byte row of a leaf no row observes
#pragma unroll
    for (uint32_t q = 0; q < FR; q++) leaf_w |= leaf_no[q] << (8 * q);
    uint32_t next_node = a;                                    // nodes below it are done
    auto leaves_until = [&](uint32_t p) {                      // the nodes in [next_node, p) have no child
        for (uint32_t d = next_node; d < p; d++) {
            uint32_t allowed[FR];
            bool any;
            const uint32_t is = observed(d, allowed, any);
            uint32_t w = leaf_w;
            if (any) {
                w = 0;
#pragma unroll
                for (uint32_t q = 0; q < FR; q++) w |= (((is >> q) & 1u) ? (~allowed[q] & 15u) : leaf_no[q]) << (8 * q);
            }
            by[(size_t)d * 64] = w;
        }
    };
    // The children of the nodes [a, b) are the contiguous range [coff[a], coff[b]) of the level
    // below: one stream of byte rows cut into sibling groups by their parent index.  Two groups of
    // FITCH_BACK_UNROLL rows are in flight: the next group is requested before the current one is used.
    const uint32_t c_begin = t.coff[a], c_end = t.coff[b];
    uint32_t cur_par = 0xFFFFFFFFu;
    X2 x[FR];
#pragma unroll
    for (uint32_t q = 0; q < FR; q++) x[q] = X2{0u, 0u};
    auto close_par = [&]() {
        uint32_t w = 0;
#pragma unroll
        for (uint32_t q = 0; q < FR; q++) {
            X2 dl;
            w |= x_close(x[q], dl) << (8 * q);
        }
        by[(size_t)cur_par * 64] = w;
    };
    auto load_group = [&](uint32_t c, uint32_t (&mb)[FITCH_BACK_UNROLL]) {
#pragma unroll
        for (uint32_t u = 0; u < FITCH_BACK_UNROLL; u++) mb[u] = by[(size_t)min(c + u, c_end - 1) * 64];
    };
    uint32_t pv = 0;                                           // parent indices of 64 consecutive children
    auto use_group = [&](uint32_t c, const uint32_t (&mb)[FITCH_BACK_UNROLL]) {
#pragma unroll
        for (uint32_t u = 0; u < FITCH_BACK_UNROLL; u++) {
            if (c + u >= c_end) break;
            const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)pv, (int)((c + u - c_begin) & 63u));
            if (p != cur_par) {                                    // uniform: a new sibling group
                if (cur_par != 0xFFFFFFFFu) close_par();
                leaves_until(p);
                uint32_t allowed[FR];
                bool any;
                const uint32_t is = observed(p, allowed, any);
#pragma unroll
                for (uint32_t q = 0; q < FR; q++) x[q] = X2{0u, 0u};
                if (any) {
#pragma unroll
                    for (uint32_t q = 0; q < FR; q++)
                        if ((is >> q) & 1u) x[q] = x_init(allowed[q]);
                }
                cur_par = p;
                next_node = p + 1;
            }
#pragma unroll
            for (uint32_t q = 0; q < FR; q++) {
                const X2 dx = x_delta_of_notopt((mb[u] >> (8 * q)) & 15u);
                x[q].a = pk_add(x[q].a, dx.a);
                x[q].b = pk_add(x[q].b, dx.b);
            }
        }
    };
    static_assert(64 % (2 * FITCH_BACK_UNROLL) == 0, "two groups per step, whole steps per 64 children");
    if (c_begin < c_end) {
        uint32_t mbA[FITCH_BACK_UNROLL], mbB[FITCH_BACK_UNROLL];
        load_group(c_begin, mbA);
        for (uint32_t c = c_begin; c < c_end; c += 2 * FITCH_BACK_UNROLL) {
            if (((c - c_begin) & 63u) == 0) {                      // the parent indices of the next 64 children
                pv = (c + lane < c_end) ? t.parent[c + lane] : 0;
            }
            load_group(c + FITCH_BACK_UNROLL, mbB);
            use_group(c, mbA);
            load_group(c + 2 * FITCH_BACK_UNROLL, mbA);
            use_group(c + FITCH_BACK_UNROLL, mbB);
        }
    }
    if (cur_par != 0xFFFFFFFFu) close_par();
    leaves_until(b);
}

__global__ __launch_bounds__(64 * FITCH_LEVEL_WAVES) void k_fitch_down(FitchLevels t, FitchSites s, uint32_t batch0,
                                                                       uint32_t lev_a, uint32_t lev_b, uint32_t nchunks,
                                                                       uint32_t nunits, uint8_t* __restrict__ bytes,
                                                                       unsigned long long* __restrict__ out_count,
                                                                       uint64_t capacity, uint2* __restrict__ out) {
    __shared__ uint2 queue_all[FITCH_LEVEL_WAVES][FITCH_QUEUE];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t unit = blockIdx.x * FITCH_LEVEL_WAVES + wv;
    if (unit >= nunits) return;
    uint2* queue = queue_all[wv];
    const uint32_t bl = unit / nchunks;
    const uint32_t ch = unit % nchunks;
    const uint32_t a = lev_a + ch * FITCH_LEVEL_CHUNK, b = min(lev_b, a + FITCH_LEVEL_CHUNK);
    uint32_t* by = reinterpret_cast<uint32_t*>(bytes + (size_t)bl * t.N * (64 * FR)) + lane;
    const uint32_t site0 = ((batch0 + bl) * 64 + lane) * FR;
    uint32_t refw = 0;                                      // the four rows' reference bases, one per byte
    uint32_t have_mask = 0;
#pragma unroll
    for (uint32_t q = 0; q < FR; q++) {
        const bool have = site0 + q < s.n_sites;
        have_mask |= (have ? 1u : 0u) << q;
        refw |= (have ? (uint32_t)s.ref_idx[site0 + q] : 0u) << (8 * q);
    }
    uint32_t qn = 0;                                        // uniform
    auto flush = [&]() {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(out_count, (unsigned long long)qn);
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        for (uint32_t i = lane; i < qn; i += 64)
            if (base + i < capacity) out[base + i] = queue[i];
        qn = 0;
    };
    // mask rows of the next nodes and the state rows of their parents (the level above, final): two
    // groups in flight, the next one requested before the current one is used; siblings read the same line
    uint32_t pv = 0;
    auto load_group = [&](uint32_t d, uint32_t (&tb)[FITCH_BACK_UNROLL], uint32_t (&ps)[FITCH_BACK_UNROLL]) {
#pragma unroll
        for (uint32_t u = 0; u < FITCH_BACK_UNROLL; u++) {
            const uint32_t dd = min(d + u, b - 1);
            tb[u] = by[(size_t)dd * 64];
            ps[u] = by[(size_t)(uint32_t)__builtin_amdgcn_readlane((int)pv, (int)((dd - a) & 63u)) * 64];
        }
    };
    auto use_group = [&](uint32_t d0, const uint32_t (&tb)[FITCH_BACK_UNROLL], const uint32_t (&ps)[FITCH_BACK_UNROLL]) {
#pragma unroll
        for (uint32_t u = 0; u < FITCH_BACK_UNROLL; u++) {
            const uint32_t d = d0 + u;
            if (d >= b) break;
            const uint32_t psw = (d == 0) ? refw : ps[u];                              // :119-128
            uint32_t stw = 0, emit_bits = 0;
#pragma unroll
            for (uint32_t q = 0; q < FR; q++) {
                const uint32_t par_state = (psw >> (8 * q)) & 3u;
                const uint32_t state = next_state<true>((tb[u] >> (8 * q)) & 15u, par_state);
                stw |= state << (8 * q);
                emit_bits |= (state != par_state ? 1u : 0u) << q;                      // :145-156
            }
            by[(size_t)d * 64] = stw;
            emit_bits &= have_mask;
            if (__ballot(emit_bits != 0)) {
#pragma unroll
                for (uint32_t q = 0; q < FR; q++) {
                    const bool emit = (emit_bits >> q) & 1u;
                    const unsigned long long mask = __ballot(emit);
                    if (mask) {
                        if (emit) {
                            const uint32_t at = qn + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                            queue[at] = make_uint2(site0 + q, d | (((psw >> (8 * q)) & 3u) << 28) | (((stw >> (8 * q)) & 3u) << 30));
                        }
                        qn += (uint32_t)__popcll(mask);
                        if (qn > FITCH_QUEUE - 64) flush();
                    }
                }
            }
        }
    };
    // windows of 64 nodes (one vector of parent indices each), 64 / FITCH_BACK_UNROLL groups per window
    uint32_t tbA[FITCH_BACK_UNROLL], psA[FITCH_BACK_UNROLL], tbB[FITCH_BACK_UNROLL], psB[FITCH_BACK_UNROLL];
    for (uint32_t w0 = a; w0 < b; w0 += 64) {
        pv = (w0 + lane < b) ? t.parent[w0 + lane] : 0;
        load_group(w0, tbA, psA);
#pragma unroll 1
        for (uint32_t g = 0; g < 64; g += 2 * FITCH_BACK_UNROLL) {
            if (w0 + g >= b) break;
            load_group(w0 + g + FITCH_BACK_UNROLL, tbB, psB);
            use_group(w0 + g, tbA, psA);
            if (g + 2 * FITCH_BACK_UNROLL < 64) load_group(w0 + g + 2 * FITCH_BACK_UNROLL, tbA, psA);
            use_group(w0 + g + FITCH_BACK_UNROLL, tbB, psB);
        }
    }
    if (qn) flush();
}

__global__ void k_fitch_sort_keys(const uint2* __restrict__ out, uint64_t n, unsigned long long* __restrict__ keys,
                                  uint32_t* __restrict__ vals) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint2 r = out[i];
    keys[i] = ((unsigned long long)r.x << 28) | (r.y & 0x0FFFFFFFu);
    vals[i] = r.y;
}

hipError_t launch_fitch_sort_keys(const uint2* out, uint64_t n, unsigned long long* keys, uint32_t* vals, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fitch_sort_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, n, keys, vals);
    return hipGetLastError();
}

// the sorted mutations in the caller's layout: row, node id, parent and new allele as one-hot masks
__global__ void k_fitch_decode(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                               const uint32_t* __restrict__ bfs2id, uint64_t n, uint32_t* __restrict__ out_site,
                               uint32_t* __restrict__ out_node, uint8_t* __restrict__ out_par,
                               uint8_t* __restrict__ out_mut) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = vals[i];
    out_site[i] = (uint32_t)(keys[i] >> 28);
    out_node[i] = bfs2id[v & 0x0FFFFFFFu];
    out_par[i] = (uint8_t)(1u << ((v >> 28) & 3u));
    out_mut[i] = (uint8_t)(1u << ((v >> 30) & 3u));
}

hipError_t launch_fitch_decode(const unsigned long long* keys, const uint32_t* vals, const uint32_t* bfs2id, uint64_t n,
                               uint32_t* out_site, uint32_t* out_node, uint8_t* out_par, uint8_t* out_mut,
                               hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fitch_decode, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, keys, vals, bfs2id, n,
                       out_site, out_node, out_par, out_mut);
    return hipGetLastError();
}

hipError_t launch_fitch_levels(const FitchLevels& t, const uint32_t* h_level_off, const FitchSites& s, uint32_t batch0,
                               uint32_t nbatches, uint8_t* bytes, unsigned long long* out_count, uint64_t capacity,
                               uint2* out, hipStream_t stream) {
    for (uint32_t l = t.n_levels; l-- > 0;) {                 // up: deepest level first
        const uint32_t a = h_level_off[l], b = h_level_off[l + 1];
        const uint32_t nch = (b - a + FITCH_LEVEL_CHUNK - 1) / FITCH_LEVEL_CHUNK;
        const uint32_t units = nch * nbatches;
        hipLaunchKernelGGL(k_fitch_up, dim3((units + FITCH_LEVEL_WAVES - 1) / FITCH_LEVEL_WAVES), dim3(64 * FITCH_LEVEL_WAVES),
                           0, stream, t, s, batch0, a, b, nch, units, bytes);
    }
    for (uint32_t l = 0; l < t.n_levels; l++) {               // down: root first
        const uint32_t a = h_level_off[l], b = h_level_off[l + 1];
        const uint32_t nch = (b - a + FITCH_LEVEL_CHUNK - 1) / FITCH_LEVEL_CHUNK;
        const uint32_t units = nch * nbatches;
        hipLaunchKernelGGL(k_fitch_down, dim3((units + FITCH_LEVEL_WAVES - 1) / FITCH_LEVEL_WAVES),
                           dim3(64 * FITCH_LEVEL_WAVES), 0, stream, t, s, batch0, a, b, nch, units, bytes, out_count, capacity,
                           out);
    }
    return hipGetLastError();
}

hipError_t launch_fitch_forward_sets(const FitchTree& t, const FitchSites& s, uint32_t batch0, uint32_t nbatches,
                                     uint8_t* tables, uint2* inh_part, uint2* out_part, hipStream_t stream) {
    const uint32_t lds = (t.max_depth + 1) * 64 * 8 + (t.max_depth + 2) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)k_fitch_forward_sets, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fitch_forward_sets, dim3(nbatches * t.C), dim3(64), lds, stream, t, s, batch0, tables, inh_part,
                       out_part);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t lds2 = (t.max_depth + 1) * 64 * 8;
    hipLaunchKernelGGL(k_fitch_stitch_sets, dim3(nbatches), dim3(64), lds2, stream, t, s, batch0, tables, inh_part, out_part);
    return hipGetLastError();
}

hipError_t launch_fitch_forward(const FitchTree& t, const FitchSites& s, uint32_t batch0, uint32_t nbatches,
                                uint8_t* tables, int4* inh_part, int4* out_part, hipStream_t stream) {
    const uint32_t lds = (t.max_depth + 1) * 64 * 16 + (t.max_depth + 2) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)k_fitch_forward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fitch_forward, dim3(nbatches * t.C), dim3(64), lds, stream, t, s, batch0, tables, inh_part,
                       out_part);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t lds2 = (t.max_depth + 1) * 64 * 16;
    e = hipFuncSetAttribute((const void*)k_fitch_stitch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fitch_stitch, dim3(nbatches), dim3(64), lds2, stream, t, s, batch0, tables, inh_part, out_part);
    return hipGetLastError();
}

hipError_t launch_fitch_backward(const FitchTree& t, const FitchSites& s, uint32_t batch0, uint32_t nbatches,
                                 const uint8_t* tables, bool masks, unsigned long long* out_count, uint64_t capacity,
                                 uint2* out, hipStream_t stream) {
    const uint32_t lds = (t.max_depth + 1) * 64 + FITCH_QUEUE * 8;
    if (masks)
        hipLaunchKernelGGL(k_fitch_backward<true>, dim3(nbatches * t.C), dim3(64), lds, stream, t, s, batch0, tables,
                           out_count, capacity, out);
    else
        hipLaunchKernelGGL(k_fitch_backward<false>, dim3(nbatches * t.C), dim3(64), lds, stream, t, s, batch0, tables,
                           out_count, capacity, out);
    return hipGetLastError();
}

}  // namespace wepp
