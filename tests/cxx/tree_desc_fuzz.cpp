// The C-ABI's tree check under AddressSanitizer + UBSan: a generated wepp_tree_desc is damaged (parents, mutation
// offsets, positions, allele masks) and flattened; a damaged description must be flattened correctly or rejected
// with an error code -- no crash, no sanitizer report.  Built and run by tests/test_pb_fuzz.py.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../include/wepp_place.h"

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 500;
    wepp_gen_tree_params p{};
    p.seed = 5; p.n_nodes = 400; p.genome_len = 300; p.p_recent_parent = 0.25; p.zipf_s = 0.6; p.p_back_mutation = 0.05;
    p.p_ambiguous = 0.02; p.p_masked_node = 0.02; p.root_mutations = 1;
    wepp_gen_tree_t* g = nullptr;
    if (wepp_gen_tree_create(&p, &g) != WEPP_OK) return 2;
    wepp_tree_desc d{};
    if (wepp_gen_tree_desc(g, &d) != WEPP_OK) return 2;
    const uint32_t N = d.n_nodes, M = d.mut_off[N];
    std::mt19937 rng(99);
    int ok = 0, rejected = 0;
    for (int r = 0; r < rounds; r++) {
        std::vector<int32_t> parent(d.parent, d.parent + N), pos(d.mut_pos, d.mut_pos + M);
        std::vector<uint32_t> off(d.mut_off, d.mut_off + N + 1);
        std::vector<uint8_t> ref(d.mut_ref, d.mut_ref + M), mut(d.mut_mut, d.mut_mut + M), par;
        if (d.mut_par) par.assign(d.mut_par, d.mut_par + M);
        const int hits = 1 + (int)(rng() % 3);
        for (int h = 0; h < hits; h++) {
            switch (rng() % 7) {
                case 0: parent[rng() % N] = (int32_t)(rng() % (N + 3)) - 1; break;                 // cycles, forests, out of range
                case 1: parent[rng() % N] = (int32_t)rng(); break;
                case 2: off[rng() % (N + 1)] = (uint32_t)(rng() % (M + 5)); break;                 // non-monotone / beyond the arrays
                case 3: if (M) pos[rng() % M] = (int32_t)(rng() % 2 ? rng() : (rng() % 400)); break;  // unsorted, huge, negative
                case 4: if (M) ref[rng() % M] = (uint8_t)(rng() % 20); break;
                case 5: if (M) mut[rng() % M] = (uint8_t)(rng() % 20); break;
                default: if (M && !par.empty()) par[rng() % M] = (uint8_t)(rng() % 20); break;
            }
        }
        // the arrays must be as long as the offsets claim for the check to be able to reject them: pad generously
        const size_t pad = 64;
        pos.resize(M + pad, 1); ref.resize(M + pad, 1); mut.resize(M + pad, 2);
        if (!par.empty()) par.resize(M + pad, 1);
        wepp_tree_desc x = d;
        x.parent = parent.data(); x.mut_off = off.data(); x.mut_pos = pos.data(); x.mut_ref = ref.data(); x.mut_mut = mut.data();
        x.mut_par = par.empty() ? nullptr : par.data();
        wepp_flat_t* f = nullptr;
        const int rc = wepp_flat_create(&x, &f);
        if (rc == WEPP_OK) { ok++; wepp_flat_destroy(f); }
        else rejected++;
    }
    wepp_gen_tree_destroy(g);
    std::printf("ok %d flattened %d rejected\n", ok, rejected);
    return 0;
}
