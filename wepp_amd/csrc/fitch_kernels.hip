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
