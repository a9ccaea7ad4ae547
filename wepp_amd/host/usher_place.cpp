// usher_place.cpp -- see usher_place.hpp (citations: /root/reference/src/usher_common.cpp).
#include "usher_place.hpp"

#include <algorithm>
#include <cstdio>
#include <unordered_map>

#include "../../include/wepp_place.h"

namespace {
struct FileCloser {
    FILE* f = nullptr;
    ~FileCloser() { if (f) fclose(f); }
};
}  // namespace

int usher_place_samples(std::string outdir, uint32_t max_uncertainty, uint32_t max_parsimony,
                        bool print_parsimony_scores, std::vector<Missing_Sample>& missing_samples,
                        std::vector<std::string>& low_confidence_samples, MAT::Tree* T,
                        std::vector<usher_place_result>* results, int device, bool sort_before_placement_1,
                        bool sort_before_placement_2, bool sort_before_placement_3, bool reverse_sort) {
    if (!T || !T->root) {
        fprintf(stderr, "ERROR: empty tree!\n");
        return 1;
    }
    if (sort_before_placement_3) {                                   // usher_common.cpp:140-155
        std::stable_sort(missing_samples.begin(), missing_samples.end());
        if (reverse_sort) std::reverse(missing_samples.begin(), missing_samples.end());
    }
    // BFS order once (the reference re-expands per sample, :339); node id = BFS
    // index, so ascending id under a parent is the stored child order.
    std::vector<MAT::Node*> bfs = T->breadth_first_expansion();
    const size_t total_nodes = bfs.size();
    std::unordered_map<const MAT::Node*, int32_t> id;
    id.reserve(total_nodes * 2);
    for (size_t k = 0; k < total_nodes; k++) id[bfs[k]] = (int32_t)k;
    std::vector<int32_t> parent(total_nodes), mut_pos;
    std::vector<uint32_t> mut_off(total_nodes + 1, 0);
    std::vector<uint8_t> mut_ref, mut_par, mut_mut;
    for (size_t k = 0; k < total_nodes; k++) {
        parent[k] = bfs[k]->parent ? id[bfs[k]->parent] : -1;
        for (auto& m : bfs[k]->mutations) {
            mut_pos.push_back(m.position);
            mut_ref.push_back((uint8_t)m.ref_nuc);
            mut_par.push_back((uint8_t)m.par_nuc);
            mut_mut.push_back((uint8_t)m.mut_nuc);
        }
        mut_off[k + 1] = (uint32_t)mut_pos.size();
    }
    wepp_tree_desc desc{(uint32_t)total_nodes, parent.data(), mut_off.data(), mut_pos.data(),
                        mut_ref.data(), mut_par.data(), mut_mut.data()};
    wepp_mat_t* mat = nullptr;
    if (wepp_mat_create(&desc, device, &mat) != WEPP_OK) {
        fprintf(stderr, "ERROR: %s\n", wepp_last_error());
        return 1;
    }

    // samples already in the tree are skipped with the reference's warning (:323-326)
    std::vector<size_t> todo;
    for (size_t s = 0; s < missing_samples.size(); s++) {
        if (T->get_node(missing_samples[s].name) != nullptr)
            fprintf(stderr, "WARNING: Sample %s already in the tree! Ignoring.\n\n", missing_samples[s].name.c_str());
        else todo.push_back(s);
    }
    std::vector<uint32_t> read_off(1, 0), read_word;
    for (size_t s : todo) {
        auto& muts = missing_samples[s].mutations;
        std::sort(muts.begin(), muts.end());                      // :200
        for (auto& m : muts)
            read_word.push_back(wepp_pack_read_word((uint32_t)m.position, (uint32_t)m.ref_nuc, (uint32_t)m.mut_nuc,
                                                    m.is_missing ? 1u : 0u));
        read_off.push_back((uint32_t)read_word.size());
    }
    const uint32_t R = (uint32_t)todo.size();
    std::vector<uint32_t> best_j(R), num_best(R), flags(R);
    std::vector<int32_t> best_sd(R), node_sd;
    if (print_parsimony_scores) node_sd.resize((size_t)R * total_nodes);
    if (wepp_place_batch(mat, read_off.data(), read_word.data(), R, best_j.data(), best_sd.data(), num_best.data(),
                         flags.data(), print_parsimony_scores ? node_sd.data() : nullptr) != WEPP_OK) {
        fprintf(stderr, "ERROR: %s\n", wepp_last_error());
        wepp_mat_destroy(mat);
        return 1;
    }
    // imputed mutations of the chosen node (column 4 of placement_stats.tsv, :764-781)
    std::vector<uint32_t> imp_off(R + 1, 0);
    std::vector<int32_t> imp_pos;
    std::vector<uint8_t> imp_nuc;
    if (!print_parsimony_scores) {
        size_t amb = 0;
        for (uint32_t w : read_word) amb += (!((w >> 28) & 1u) && (((w >> 24) & 15u) & (((w >> 24) & 15u) - 1))) ? 1 : 0;
        imp_pos.resize(amb + 1);
        imp_nuc.resize(amb + 1);
        if (wepp_imputed_mutations(mat, read_off.data(), read_word.data(), R, best_j.data(), imp_off.data(),
                                   imp_pos.data(), imp_nuc.data(), amb) != WEPP_OK) {
            fprintf(stderr, "ERROR: %s\n", wepp_last_error());
            wepp_mat_destroy(mat);
            return 1;
        }
    }
    // excess mutations of the optimal nodes (last column of parsimony-scores.tsv, :555-574)
    std::vector<uint64_t> exc_off(1, 0);
    std::vector<int32_t> exc_pos;
    std::vector<uint8_t> exc_ref, exc_par, exc_mut;
    std::vector<size_t> exc_first(R + 1, 0);      // first pair of sample q (its optimal nodes in BFS order)
    if (print_parsimony_scores) {
        std::vector<uint32_t> pair_read, pair_node;
        for (uint32_t q = 0; q < R; q++) {
            exc_first[q] = pair_read.size();
            const int32_t* nsd = node_sd.data() + (size_t)q * total_nodes;
            for (size_t k = 0; k < total_nodes; k++)
                if (nsd[k] == best_sd[q] && nsd[k] != 0) { pair_read.push_back(q); pair_node.push_back((uint32_t)k); }
        }
        exc_first[R] = pair_read.size();
        exc_off.assign(pair_read.size() + 1, 0);
        uint64_t cap = 0;
        for (int attempt = 0; attempt < 2; attempt++) {
            exc_pos.resize(cap + 1); exc_ref.resize(cap + 1); exc_par.resize(cap + 1); exc_mut.resize(cap + 1);
            int rc = wepp_excess_mutations(mat, read_off.data(), read_word.data(), R, (uint32_t)pair_read.size(),
                                           pair_read.data(), pair_node.data(), exc_off.data(), exc_pos.data(),
                                           exc_ref.data(), exc_par.data(), exc_mut.data(), cap);
            if (rc == WEPP_ELIMIT && attempt == 0 && exc_off.back() > cap) { cap = exc_off.back(); continue; }
            if (rc != WEPP_OK) {
                fprintf(stderr, "ERROR: %s\n", wepp_last_error());
                wepp_mat_destroy(mat);
                return 1;
            }
            break;
        }
    }
    wepp_mat_destroy(mat);

    FileCloser stats, scores;
    if (!outdir.empty()) {
        stats.f = fopen((outdir + "/placement_stats.tsv").c_str(), "w");                      // :303-304
        if (!stats.f) {
            fprintf(stderr, "ERROR: cannot write to %s\n", outdir.c_str());
            return 1;
        }
    }
    if (results) results->clear();
    // order in which the rows are written (usher_common.cpp:164-169, :276-295)
    std::vector<uint32_t> indexes(R);
    for (uint32_t q = 0; q < R; q++) indexes[q] = q;
    if ((sort_before_placement_1 || sort_before_placement_2) && missing_samples.size() > 1) {
        if (sort_before_placement_1)
            std::stable_sort(indexes.begin(), indexes.end(), [&](uint32_t a, uint32_t b) {
                return best_sd[a] < best_sd[b] || (best_sd[a] == best_sd[b] && num_best[a] < num_best[b]);
            });
        else
            std::stable_sort(indexes.begin(), indexes.end(), [&](uint32_t a, uint32_t b) {
                return num_best[a] < num_best[b] || (num_best[a] == num_best[b] && best_sd[a] < best_sd[b]);
            });
        if (reverse_sort) std::reverse(indexes.begin(), indexes.end());
    }
    for (uint32_t qi = 0; qi < R; qi++) {
        const uint32_t q = indexes[qi];
        const std::string& sample = missing_samples[todo[q]].name;
        if (print_parsimony_scores && !outdir.empty() && qi == 0) {
            std::string fn = outdir + "/parsimony-scores.tsv";                                 // :329-336
            fprintf(stderr, "\nNow computing branch parsimony scores for adding the missing samples at each of the %zu nodes in the existing tree without modifying the tree.\n", total_nodes);
            fprintf(stderr, "The branch parsimony scores will be written to file %s\n\n", fn.c_str());
            scores.f = fopen(fn.c_str(), "w");
            if (scores.f)
                fprintf(scores.f, "#Sample\tTree node\tParsimony score\tOptimal (y/n)\tParsimony-increasing mutations (for optimal nodes)\n");
        }
        const int best_set_difference = best_sd[q];
        const size_t nb = num_best[q];
        if (!print_parsimony_scores) {
            fprintf(stderr, "Current tree size (#nodes): %zu\tSample name: %s\tParsimony score: %d\tNumber of parsimony-optimal placements: %zu\n",
                    total_nodes, sample.c_str(), best_set_difference, nb);                    // :448-449
            if (stats.f) fprintf(stats.f, "%s\t%d\t%zu\t", sample.c_str(), best_set_difference, nb);   // :450
            if (nb > 1) {                                                                      // :453-462
                low_confidence_samples.emplace_back(sample);
                if (nb > max_uncertainty)
                    fprintf(stderr, "WARNING: Number of parsimony-optimal placements exceeds maximum allowed value (%u). Ignoring sample %s.\n", max_uncertainty, sample.c_str());
                else if (best_set_difference <= (int)max_parsimony)
                    fprintf(stderr, "WARNING: Multiple parsimony-optimal placements found. Placement done without high confidence.\n");
            }
            if (best_set_difference > (int)max_parsimony)                                      // :464-466
                fprintf(stderr, "WARNING: Parsimony score of the most parsimonious placement exceeds the maximum allowed value (%u). Ignoring sample %s.\n", max_parsimony, sample.c_str());
            // :580 the placement branch (where the imputed mutations are printed) is only
            // entered within the thresholds
            if (nb <= max_uncertainty && best_set_difference <= (int)max_parsimony && imp_off[q + 1] > imp_off[q]) {
                fprintf(stderr, "Imputed mutations:\t");                                      // :764-781
                for (uint32_t i = imp_off[q]; i < imp_off[q + 1]; i++) {
                    const char* sep = (i + 1 < imp_off[q + 1]) ? ";" : "";
                    fprintf(stderr, "%i:%c%s", imp_pos[i], MAT::get_nuc((int8_t)imp_nuc[i]), sep);
                    if (stats.f) fprintf(stats.f, "%i:%c%s", imp_pos[i], MAT::get_nuc((int8_t)imp_nuc[i]), sep);
                }
                fprintf(stderr, "\n");
            }
        } else {
            fprintf(stderr, "Missing sample: %s\t Best parsimony score: %d\tNumber of parsimony-optimal placements: %zu\n",
                    sample.c_str(), best_set_difference, nb);                                  // :468-469
            if (scores.f) {
                const int32_t* nsd = node_sd.data() + (size_t)q * total_nodes;
                size_t pair = exc_first[q];
                for (size_t k = 0; k < total_nodes; k++) {                                     // :555-574
                    const bool optimal = nsd[k] == best_set_difference;
                    fprintf(scores.f, "%s\t%s\t%d\t\t%c\t", sample.c_str(), bfs[k]->identifier.c_str(), nsd[k], optimal ? 'y' : 'n');
                    if (!optimal) fprintf(scores.f, "N/A");
                    else if (nsd[k] == 0) fprintf(scores.f, "*");
                    else {
                        // the reference prints the first node_set_difference[k] entries of the vector (which
                        // starts with the shared mutations of the node); it reads past the end when the
                        // vector is shorter (a node that does not compete reports score + 1,
                        // usher_mapper.cpp:500-505) -- here the loop stops at the end
                        const uint64_t end = std::min<uint64_t>(exc_off[pair + 1], exc_off[pair] + (uint64_t)nsd[k]);
                        for (uint64_t i = exc_off[pair]; i < end; i++) {
                            MAT::Mutation m;
                            m.position = exc_pos[i]; m.ref_nuc = (int8_t)exc_ref[i];
                            m.par_nuc = (int8_t)exc_par[i]; m.mut_nuc = (int8_t)exc_mut[i];
                            fprintf(scores.f, "%s%s", m.get_string().c_str(), i + 1 < end ? "," : "");
                        }
                        pair++;
                    }
                    fprintf(scores.f, "\n");
                }
            }
        }
        if (stats.f) fputc('\n', stats.f);   // :785
        if (results)
            results->push_back({best_set_difference, nb, (size_t)best_j[q], bfs[best_j[q]],
                                (flags[q] & WEPP_FLAG_HAS_UNIQUE) != 0});
    }
    return 0;
}
