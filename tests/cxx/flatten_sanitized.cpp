// The flattener (tree -> sweep streams, window streams, position index, range-query tables, EPP stream) built with
// AddressSanitizer + UndefinedBehaviorSanitizer and run over generated trees of several shapes, with and without the
// per-entry pre-test bytes.  Built and run by tests/test_flatten_sanitized.py (CPU build only).
#include <cstdio>
#include <cstdlib>

#include "../../include/wepp_place.h"

int main() {
    struct Case { uint64_t seed; uint32_t nodes, genome; double p_recent, zipf, p_back, p_amb, p_masked; uint32_t root_muts; };
    const Case cases[] = {
        {1, 300, 60, 0.25, 0.6, 0.05, 0.1, 0.05, 1},      {2, 5000, 800, 0.25, 0.6, 0.02, 0.0, 0.0, 0},
        {3, 20000, 29903, 0.25, 0.6, 0.02, 0.003, 0.01, 0}, {4, 60000, 29903, 0.4, 0.9, 0.1, 0.0, 0.0, 2},
        {5, 9000, 3000, 0.1, 0.3, 0.3, 0.05, 0.02, 0},     {6, 2, 10, 0.25, 0.6, 0.0, 0.0, 0.0, 0},
        {7, 120000, 29903, 0.25, 0.6, 0.02, 0.0, 0.0, 0},
    };
    for (const Case& c : cases) {
        wepp_gen_tree_params p{};
        p.seed = c.seed; p.n_nodes = c.nodes; p.genome_len = c.genome; p.p_recent_parent = c.p_recent; p.zipf_s = c.zipf;
        p.p_back_mutation = c.p_back; p.p_ambiguous = c.p_amb; p.p_masked_node = c.p_masked; p.root_mutations = c.root_muts;
        wepp_gen_tree_t* g = nullptr;
        if (wepp_gen_tree_create(&p, &g) != WEPP_OK) { std::printf("gen failed: %s\n", wepp_last_error()); return 1; }
        wepp_tree_desc d{};
        if (wepp_gen_tree_desc(g, &d) != WEPP_OK) return 1;
        wepp_flat_t* f = nullptr;
        if (wepp_flat_create(&d, &f) != WEPP_OK) { std::printf("flatten failed: %s\n", wepp_last_error()); return 1; }
        uint64_t cnt = 0; uint32_t eb = 0; const void* data = nullptr;
        if (wepp_flat_get(f, "rank2bfs", &data, &cnt, &eb) != WEPP_OK || cnt != c.nodes) { std::printf("bad rank2bfs\n"); return 1; }
        wepp_flat_destroy(f);
        wepp_gen_tree_destroy(g);
    }
    std::printf("ok %zu trees\n", sizeof(cases) / sizeof(cases[0]));
    return 0;
}
