// gen.cpp -- deterministic synthetic MATs and reads (host only).
//
// The reference bundles no MAT and no reads (both README
// examples start with wget) and the GPU box has no network, so every config
// of BASELINE.json runs on data generated here from seeds.  Only splitmix64 is
// used (never std::*_distribution) so that the same seed gives the same bytes
// on every box.  Shapes follow SURVEY.md section 8(d): ~1.0-1.1 mutations per
// node (27 % none, 55 % one, 18 % two-four), Zipf-like site weights for
// homoplasy, a few back-mutations, optional ambiguous / masked mutations;
// reads are windows of a tiled amplicon scheme (ARTIC-like 400 bp amplicons
// for 150 bp reads, midnight-like 1200 bp amplicons for long reads) drawn from
// random leaf genotypes with substitution errors and Ns.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/wepp_place.h"
#include "errors.hpp"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // uniform in [0, n)
    uint64_t below(uint64_t n) { return (uint64_t)(((__uint128_t)next() * n) >> 64); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    bool chance(double p) { return p > 0.0 && unit() < p; }
};

inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace

struct wepp_gen_tree {
    wepp_gen_tree_params p;
    std::vector<int32_t> parent;
    std::vector<uint32_t> mut_off;
    std::vector<int32_t> mut_pos;
    std::vector<uint8_t> mut_ref, mut_par, mut_mut;
    std::vector<uint32_t> leaves;
    uint64_t ref_seed;
    // reference base (one-hot mask) at 1-based position
    uint8_t ref_base(uint32_t pos) const { return (uint8_t)(1u << (mix64(ref_seed ^ (0x51ED27ull * pos)) & 3)); }
    // allele state of node `n`'s genotype at `pos`: 0 = no mutation on the path
    uint8_t state_at(int32_t n, int32_t pos) const {
        while (n >= 0) {
            uint32_t a = mut_off[n], b = mut_off[n + 1];
            for (uint32_t k = a; k < b; k++)
                if (mut_pos[k] == pos) return mut_mut[k];
            n = parent[n];
        }
        return 0;
    }
};

struct wepp_gen_reads {
    std::vector<uint32_t> read_off, read_word;
    std::vector<int32_t> win_start, win_end;   // genome window every read was drawn from
};

extern "C" int wepp_gen_tree_create(const wepp_gen_tree_params* pp, wepp_gen_tree_t** out) {
    if (!pp || !out) return wepp::set_error(WEPP_EINVAL, "null argument");
    if (pp->n_nodes == 0 || pp->genome_len == 0 || pp->genome_len > WEPP_MAX_POSITION)
        return wepp::set_error(WEPP_EINVAL, "n_nodes and genome_len must be positive (genome_len < 2^20)");
    wepp_gen_tree* t = new (std::nothrow) wepp_gen_tree();
    if (!t) return wepp::set_error(WEPP_ENOMEM, "out of host memory");
    try {
        t->p = *pp;
        const uint32_t N = pp->n_nodes, L = pp->genome_len;
        Rng rng(pp->seed);
        t->ref_seed = mix64(pp->seed ^ 0xA5A5A5A5DEADBEEFull);
        // site weights: Zipf over a seeded permutation of the sites
        std::vector<uint32_t> site(L);
        for (uint32_t i = 0; i < L; i++) site[i] = i + 1;
        for (uint32_t i = L; i > 1; i--) std::swap(site[i - 1], site[rng.below(i)]);
        std::vector<double> cdf(L);
        double acc = 0;
        for (uint32_t i = 0; i < L; i++) { acc += 1.0 / std::pow((double)(i + 1), pp->zipf_s); cdf[i] = acc; }
        auto draw_site = [&]() -> uint32_t {
            double u = rng.unit() * acc;
            size_t k = std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin();
            if (k >= L) k = L - 1;
            return site[k];
        };
        t->parent.assign(N, -1);
        t->mut_off.assign(N + 1, 0);
        t->mut_pos.reserve((size_t)N + N / 8);
        t->mut_ref.reserve((size_t)N + N / 8);
        t->mut_par.reserve((size_t)N + N / 8);
        t->mut_mut.reserve((size_t)N + N / 8);
        std::vector<uint32_t> nchild(N, 0);
        std::vector<std::pair<int32_t, uint8_t>> mine;  // (pos, mut)
        // tree shape knobs (wepp_place.h): with the defaults no extra random number is drawn -- the default tree of a
        // seed is the tree it has always been
        const uint32_t choices = std::max<uint32_t>(1, pp->depth_choices);
        const uint32_t hubs = pp->p_hub > 0.0 ? (pp->n_hubs ? std::min(pp->n_hubs, N) : N / 4096 + 1) : 0;
        std::vector<uint32_t> depth(choices > 1 ? N : 0, 0);
        for (uint32_t i = 0; i < N; i++) {
            if (i > 0) {
                uint32_t par;
                if (hubs && i > hubs && rng.chance(pp->p_hub)) par = (uint32_t)rng.below(hubs);
                else if (i > 8 && rng.chance(pp->p_recent_parent)) par = i - 1 - (uint32_t)rng.below(8);
                else {
                    par = (uint32_t)rng.below(i);
                    for (uint32_t c = 1; c < choices; c++) {
                        const uint32_t alt = (uint32_t)rng.below(i);
                        if (depth[alt] > depth[par]) par = alt;
                    }
                }
                if (choices > 1) depth[i] = depth[par] + 1;
                t->parent[i] = (int32_t)par;
                nchild[par]++;
            }
            uint32_t nm;
            if (i == 0) nm = pp->root_mutations;
            else {
                double u = rng.unit();
                nm = u < 0.27 ? 0 : (u < 0.82 ? 1 : 2 + (uint32_t)rng.below(3));
            }
            mine.clear();
            bool masked = (i > 0) && rng.chance(pp->p_masked_node);
            for (uint32_t k = 0; k < nm; k++) {
                uint32_t pos = draw_site();
                bool dup = false;
                for (auto& q : mine) dup |= (q.first == (int32_t)pos);
                if (dup) continue;
                mine.emplace_back((int32_t)pos, 0);
            }
            std::sort(mine.begin(), mine.end());
            if (masked) {
                t->mut_pos.push_back(-1);
                t->mut_ref.push_back(0);
                t->mut_par.push_back(0);
                t->mut_mut.push_back(0);
            }
            for (auto& q : mine) {
                uint8_t ref = t->ref_base((uint32_t)q.first);
                uint8_t st = (i == 0) ? 0 : t->state_at(t->parent[i], q.first);
                uint8_t cur = st ? st : ref;  // parent genotype allele
                uint8_t mut;
                if (cur != ref && rng.chance(pp->p_back_mutation)) mut = ref;
                else {
                    // a one-hot allele not contained in the parent allele
                    do { mut = (uint8_t)(1u << rng.below(4)); } while (mut & cur);
                }
                if (rng.chance(pp->p_ambiguous)) {
                    uint8_t extra;
                    do { extra = (uint8_t)(1u << rng.below(4)); } while (extra == mut);
                    if ((uint8_t)(mut | extra) != cur) mut |= extra;
                }
                t->mut_pos.push_back(q.first);
                t->mut_ref.push_back(ref);
                t->mut_par.push_back(cur);
                t->mut_mut.push_back(mut);
            }
            t->mut_off[i + 1] = (uint32_t)t->mut_pos.size();
        }
        for (uint32_t i = 0; i < N; i++)
            if (nchild[i] == 0) t->leaves.push_back(i);
    } catch (const std::bad_alloc&) {
        delete t;
        return wepp::set_error(WEPP_ENOMEM, "out of host memory");
    }
    *out = t;
    return WEPP_OK;
}

extern "C" int wepp_gen_tree_desc(const wepp_gen_tree_t* t, wepp_tree_desc* out) {
    if (!t || !out) return wepp::set_error(WEPP_EINVAL, "null argument");
    out->n_nodes = (uint32_t)t->parent.size();
    out->parent = t->parent.data();
    out->mut_off = t->mut_off.data();
    out->mut_pos = t->mut_pos.data();
    out->mut_ref = t->mut_ref.data();
    out->mut_par = t->mut_par.data();
    out->mut_mut = t->mut_mut.data();
    return WEPP_OK;
}

extern "C" int wepp_gen_tree_get_shape(const wepp_gen_tree_t* t, wepp_gen_tree_shape* out) {
    if (!t || !out) return wepp::set_error(WEPP_EINVAL, "null argument");
    const uint32_t N = (uint32_t)t->parent.size();
    std::vector<uint32_t> pm(N, 0), dep(N, 0), nch(N, 0);
    uint32_t max_depth = 0, max_children = 0;
    for (uint32_t i = 0; i < N; i++) {          // (a parent's index is below its children's)
        uint32_t own = 0;
        for (uint32_t k = t->mut_off[i]; k < t->mut_off[i + 1]; k++) own += t->mut_pos[k] >= 0 ? 1u : 0u;
        if (i) {
            const uint32_t p = (uint32_t)t->parent[i];
            pm[i] = pm[p] + own;
            dep[i] = dep[p] + 1;
            max_children = std::max(max_children, ++nch[p]);
        } else pm[i] = own;
        max_depth = std::max(max_depth, dep[i]);
    }
    std::vector<uint32_t> leaf_pm;
    leaf_pm.reserve(t->leaves.size());
    double sum = 0;
    for (uint32_t l : t->leaves) { leaf_pm.push_back(pm[l]); sum += pm[l]; }
    std::sort(leaf_pm.begin(), leaf_pm.end());
    const size_t nl = leaf_pm.size();
    out->n_nodes = N;
    out->n_leaves = (uint32_t)nl;
    out->max_depth = max_depth;
    out->max_children = max_children;
    out->path_mutations_median = nl ? leaf_pm[nl / 2] : 0;
    out->path_mutations_p95 = nl ? leaf_pm[std::min(nl - 1, nl * 95 / 100)] : 0;
    out->path_mutations_max = nl ? leaf_pm.back() : 0;
    out->path_mutations_mean = nl ? sum / (double)nl : 0.0;
    out->mutations_per_node = N ? (double)t->mut_pos.size() / N : 0.0;
    return WEPP_OK;
}

extern "C" int wepp_gen_tree_destroy(wepp_gen_tree_t* t) {
    delete t;
    return WEPP_OK;
}

extern "C" int wepp_gen_reads_create(const wepp_gen_tree_t* t, const wepp_gen_reads_params* pp,
                                     wepp_gen_reads_t** out) {
    if (!t || !pp || !out) return wepp::set_error(WEPP_EINVAL, "null argument");
    if (pp->read_len == 0 || pp->amplicon_len == 0 || pp->amplicon_step == 0)
        return wepp::set_error(WEPP_EINVAL, "read_len, amplicon_len and amplicon_step must be positive");
    wepp_gen_reads* r = new (std::nothrow) wepp_gen_reads();
    if (!r) return wepp::set_error(WEPP_ENOMEM, "out of host memory");
    try {
        const uint32_t L = t->p.genome_len;
        Rng rng(pp->seed);
        const uint32_t n_amp = std::max<uint32_t>(1, (L + pp->amplicon_step - 1) / pp->amplicon_step);
        r->read_off.assign((size_t)pp->n_reads + 1, 0);
        r->read_word.reserve((size_t)pp->n_reads * 2);
        struct Ent { uint32_t pos; uint8_t ref, mut, missing; };
        std::vector<Ent> ents;
        std::vector<int32_t> seen;
        auto geo = [&](double p) -> uint32_t {  // gap to the next event, >= 1
            if (p <= 0.0) return 0xFFFFFFFFu;
            if (p >= 1.0) return 1;
            double u = rng.unit();
            double g = std::floor(std::log1p(-u) / std::log1p(-p)) + 1.0;
            return g > 4e9 ? 0xFFFFFFFFu : (uint32_t)g;
        };
        for (uint32_t q = 0; q < pp->n_reads; q++) {
            uint32_t leaf = t->leaves[rng.below(t->leaves.size())];
            uint32_t amp = (uint32_t)rng.below(n_amp);
            uint32_t a0 = 1 + amp * pp->amplicon_step;
            uint32_t a1 = std::min<uint32_t>(L, a0 + pp->amplicon_len - 1);
            uint32_t ws = a0, we = a1;
            if (a1 - a0 + 1 > pp->read_len) {
                ws = a0 + (uint32_t)rng.below(a1 - a0 + 2 - pp->read_len);
                we = ws + pp->read_len - 1;
            }
            r->win_start.push_back((int32_t)ws);
            r->win_end.push_back((int32_t)we);
            // leaf genotype inside the window: most recent mutation per position
            ents.clear();
            seen.clear();
            for (int32_t n = (int32_t)leaf; n >= 0; n = t->parent[n]) {
                for (uint32_t k = t->mut_off[n]; k < t->mut_off[n + 1]; k++) {
                    int32_t p = t->mut_pos[k];
                    if (p < (int32_t)ws || p > (int32_t)we) continue;
                    if (std::find(seen.begin(), seen.end(), p) != seen.end()) continue;
                    seen.push_back(p);
                    uint8_t ref = t->mut_ref[k], m = t->mut_mut[k];
                    uint8_t base = (uint8_t)(m & (uint8_t)(-(int8_t)m));  // a read shows one concrete base
                    if (base != ref) ents.push_back({(uint32_t)p, ref, base, 0});
                }
            }
            auto set_entry = [&](uint32_t pos, uint8_t ref, uint8_t mut, uint8_t missing, bool remove) {
                for (size_t i = 0; i < ents.size(); i++)
                    if (ents[i].pos == pos) {
                        if (remove) ents.erase(ents.begin() + (long)i);
                        else { ents[i].mut = mut; ents[i].missing = missing; }
                        return;
                    }
                if (!remove) ents.push_back({pos, ref, mut, missing});
            };
            // substitution errors
            for (uint64_t pos = (uint64_t)ws - 1 + geo(pp->p_substitution); pos <= we; pos += geo(pp->p_substitution)) {
                uint8_t ref = t->ref_base((uint32_t)pos);
                uint8_t cur = ref;
                for (auto& e : ents) if (e.pos == pos) cur = e.mut;
                uint8_t nb;
                do { nb = (uint8_t)(1u << rng.below(4)); } while (nb == cur);
                if (rng.chance(pp->p_iupac)) {
                    uint8_t amb = (uint8_t)(nb | (1u << rng.below(4)));
                    if (amb != 15) { set_entry((uint32_t)pos, ref, amb, 0, false); continue; }
                }
                set_entry((uint32_t)pos, ref, nb, 0, nb == ref);
            }
            // Ns
            for (uint64_t pos = (uint64_t)ws - 1 + geo(pp->p_n); pos <= we; pos += geo(pp->p_n))
                set_entry((uint32_t)pos, t->ref_base((uint32_t)pos), 15, 1, false);
            std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.pos < b.pos; });
            for (auto& e : ents) r->read_word.push_back(wepp_pack_read_word(e.pos, e.ref, e.mut, e.missing));
            if (r->read_word.size() >= 0xFFFFFFF0ull) { delete r; return wepp::set_error(WEPP_ELIMIT, "too many read words"); }
            r->read_off[q + 1] = (uint32_t)r->read_word.size();
        }
    } catch (const std::bad_alloc&) {
        delete r;
        return wepp::set_error(WEPP_ENOMEM, "out of host memory");
    }
    *out = r;
    return WEPP_OK;
}

extern "C" int wepp_gen_reads_windows(const wepp_gen_reads_t* r, const int32_t** start, const int32_t** end) {
    if (!r) return wepp::set_error(WEPP_EINVAL, "null argument");
    if (start) *start = r->win_start.data();
    if (end) *end = r->win_end.data();
    return WEPP_OK;
}

extern "C" int wepp_gen_reads_get(const wepp_gen_reads_t* r, uint32_t* n_reads, const uint32_t** read_off,
                                  const uint32_t** read_word) {
    if (!r) return wepp::set_error(WEPP_EINVAL, "null argument");
    if (n_reads) *n_reads = (uint32_t)(r->read_off.size() - 1);
    if (read_off) *read_off = r->read_off.data();
    if (read_word) *read_word = r->read_word.data();
    return WEPP_OK;
}

extern "C" int wepp_gen_reads_destroy(wepp_gen_reads_t* r) {
    delete r;
    return WEPP_OK;
}
