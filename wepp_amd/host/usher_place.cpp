// usher_place.cpp -- see usher_place.hpp (citations: /root/reference/src/usher_common.cpp).
#include "usher_place.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>
#include <unordered_map>

#include "../../include/wepp_place.h"

namespace {
struct FileCloser {
    FILE* f = nullptr;
    ~FileCloser() { if (f) fclose(f); }
};
}  // namespace

usher_place_timing usher_last_timing;

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct FlatTree {                    // wepp_tree_desc of a MAT::Tree, node id = BFS index
    std::vector<MAT::Node*> bfs;
    std::vector<int32_t> parent, mut_pos;
    std::vector<uint32_t> mut_off;
    std::vector<uint8_t> mut_ref, mut_par, mut_mut;
    wepp_tree_desc desc{};
};

void flatten(MAT::Tree* T, FlatTree& f) {
    // BFS order once (the reference re-expands per sample, :339); node id = BFS
    // index, so ascending id under a parent is the stored child order.
    f.bfs = T->breadth_first_expansion();
    const size_t n = f.bfs.size();
    // the children of bfs[head] follow one another in the expansion, behind everything queued before them: the
    // parent's index falls out of the order itself (no Node* -> index map over 16 M nodes)
    f.parent.assign(n, -1);
    for (size_t head = 0, next = 1; head < n; head++)
        for (size_t c = 0; c < f.bfs[head]->children.size(); c++) f.parent[next++] = (int32_t)head;
    f.mut_off.assign(n + 1, 0);
    size_t total_muts = 0;
    for (size_t k = 0; k < n; k++) total_muts += f.bfs[k]->mutations.size();
    f.mut_pos.reserve(total_muts); f.mut_ref.reserve(total_muts); f.mut_par.reserve(total_muts); f.mut_mut.reserve(total_muts);
    for (size_t k = 0; k < n; k++) {
        for (auto& m : f.bfs[k]->mutations) {
            f.mut_pos.push_back(m.position);
            f.mut_ref.push_back((uint8_t)m.ref_nuc);
            f.mut_par.push_back((uint8_t)m.par_nuc);
            f.mut_mut.push_back((uint8_t)m.mut_nuc);
        }
        f.mut_off[k + 1] = (uint32_t)f.mut_pos.size();
    }
    f.desc = wepp_tree_desc{(uint32_t)n, f.parent.data(), f.mut_off.data(), f.mut_pos.data(), f.mut_ref.data(),
                            f.mut_par.data(), f.mut_mut.data()};
}

// read CSR of the samples sel[lo..hi), offsets starting at 0
void gather_reads(const std::vector<uint32_t>& read_off, const std::vector<uint32_t>& read_word,
                  const uint32_t* sel, uint32_t lo, uint32_t hi, std::vector<uint32_t>& off, std::vector<uint32_t>& words) {
    off.assign(1, 0);
    words.clear();
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t q = sel ? sel[i] : i;
        words.insert(words.end(), read_word.begin() + read_off[q], read_word.begin() + read_off[q + 1]);
        off.push_back((uint32_t)words.size());
    }
}

// the -p mode's data of a chunk of samples (rows of parsimony-scores.tsv)
struct ScoreChunk {
    uint32_t lo = 0, hi = 0;                 // rows [lo, hi) of the output order
    std::vector<int32_t> node_sd;            // (hi - lo) x total_nodes
    std::vector<size_t> exc_first;           // first (sample, optimal node) pair of each sample
    std::vector<uint64_t> exc_off;
    std::vector<int32_t> exc_pos;
    std::vector<uint8_t> exc_ref, exc_par, exc_mut;
    std::string error;
};

}  // namespace

int usher_place_samples(std::string outdir, uint32_t max_uncertainty, uint32_t max_parsimony,
                        bool print_parsimony_scores, std::vector<Missing_Sample>& missing_samples,
                        std::vector<std::string>& low_confidence_samples, MAT::Tree* T,
                        std::vector<usher_place_result>* results, int device, bool sort_before_placement_1,
                        bool sort_before_placement_2, bool sort_before_placement_3, bool reverse_sort) {
    return usher_place_samples(outdir, max_uncertainty, max_parsimony, print_parsimony_scores, missing_samples,
                               low_confidence_samples, T, std::vector<int>{device}, results, sort_before_placement_1,
                               sort_before_placement_2, sort_before_placement_3, reverse_sort);
}

int usher_place_samples(std::string outdir, uint32_t max_uncertainty, uint32_t max_parsimony,
                        bool print_parsimony_scores, std::vector<Missing_Sample>& missing_samples,
                        std::vector<std::string>& low_confidence_samples, MAT::Tree* T,
                        const std::vector<int>& devices, std::vector<usher_place_result>* results,
                        bool sort_before_placement_1, bool sort_before_placement_2, bool sort_before_placement_3,
                        bool reverse_sort) {
    if (!T || !T->root) {
        fprintf(stderr, "ERROR: empty tree!\n");
        return 1;
    }
    if (devices.empty()) {
        fprintf(stderr, "ERROR: no device given!\n");
        return 1;
    }
    if (sort_before_placement_3) {                                   // usher_common.cpp:140-155
        std::stable_sort(missing_samples.begin(), missing_samples.end());
        if (reverse_sort) std::reverse(missing_samples.begin(), missing_samples.end());
    }
    usher_last_timing = usher_place_timing{};
    const uint64_t flattens_before = wepp_debug_flatten_count();
    double t_mark = now_s();
    FlatTree flat;
    flatten(T, flat);
    usher_last_timing.describe_s = now_s() - t_mark;
    const std::vector<MAT::Node*>& bfs = flat.bfs;
    const size_t total_nodes = bfs.size();

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
    std::vector<int32_t> best_sd(R);
    std::vector<uint32_t> imp_off(R + 1, 0);      // imputed mutations of the chosen node (column 4 of
    std::vector<int32_t> imp_pos;                 // placement_stats.tsv, :764-781), CSR over the samples
    std::vector<uint8_t> imp_nuc;

    // ---- one host thread and one handle per device; device g places the contiguous range of samples
    // [R*g/G, R*(g+1)/G) into its slice of the output arrays (usher_common.cpp:386-446 per sample; no
    // exchange between the devices) ------------------------------------------------------------------
    const uint32_t G = (uint32_t)devices.size();
    std::vector<wepp_mat_t*> mats(G, nullptr);
    std::vector<std::string> errors(G);
    std::vector<std::vector<uint32_t>> shard_imp_cnt(G);
    std::vector<std::vector<int32_t>> shard_imp_pos(G);
    std::vector<std::vector<uint8_t>> shard_imp_nuc(G);
    auto shard_lo = [&](uint32_t g) { return (uint32_t)((uint64_t)R * g / G); };
    // ONE flatten for the node (the reference pays its expansion once per sample, usher_common.cpp:339); every device
    // thread uploads the same image
    wepp_flat_t* image = nullptr;
    t_mark = now_s();
    if (wepp_flat_create(&flat.desc, &image) != WEPP_OK) {
        fprintf(stderr, "ERROR: %s\n", wepp_last_error());
        return 1;
    }
    usher_last_timing.flatten_s = now_s() - t_mark;
    std::vector<double> upload_s(G, 0.0), place_s(G, 0.0);
    auto place_shard = [&](uint32_t g) {
        const double t0 = now_s();
        if (wepp_mat_upload(image, devices[g], &mats[g]) != WEPP_OK) { errors[g] = wepp_last_error(); return; }
        upload_s[g] = now_s() - t0;
        struct Stop { double t0; double& out; ~Stop() { out = now_s() - t0; } } stop{now_s(), place_s[g]};
        const uint32_t lo = shard_lo(g), hi = shard_lo(g + 1), n = hi - lo;
        if (n == 0) return;
        std::vector<uint32_t> off, words;
        gather_reads(read_off, read_word, nullptr, lo, hi, off, words);
        if (wepp_place_batch(mats[g], off.data(), words.data(), n, best_j.data() + lo, best_sd.data() + lo,
                             num_best.data() + lo, flags.data() + lo, nullptr) != WEPP_OK) {
            errors[g] = wepp_last_error();
            return;
        }
        if (print_parsimony_scores) return;
        size_t amb = 0;
        for (uint32_t w : words) amb += (!((w >> 28) & 1u) && (((w >> 24) & 15u) & (((w >> 24) & 15u) - 1))) ? 1 : 0;
        std::vector<uint32_t> ioff(n + 1, 0);
        shard_imp_pos[g].resize(amb + 1);
        shard_imp_nuc[g].resize(amb + 1);
        if (wepp_imputed_mutations(mats[g], off.data(), words.data(), n, best_j.data() + lo, ioff.data(),
                                   shard_imp_pos[g].data(), shard_imp_nuc[g].data(), amb) != WEPP_OK) {
            errors[g] = wepp_last_error();
            return;
        }
        shard_imp_cnt[g].resize(n);
        for (uint32_t i = 0; i < n; i++) shard_imp_cnt[g][i] = ioff[i + 1] - ioff[i];
        shard_imp_pos[g].resize(ioff[n]);
        shard_imp_nuc[g].resize(ioff[n]);
    };
    auto run_on_devices = [&](const std::function<void(uint32_t)>& fn) {
        if (G == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (uint32_t g = 0; g < G; g++) th.emplace_back(fn, g);
        for (auto& t : th) t.join();
    };
    auto destroy_all = [&]() { for (auto m : mats) if (m) wepp_mat_destroy(m); };
    run_on_devices(place_shard);
    wepp_flat_destroy(image);
    usher_last_timing.upload_s = *std::max_element(upload_s.begin(), upload_s.end());
    usher_last_timing.place_s = *std::max_element(place_s.begin(), place_s.end());
    usher_last_timing.flattens = wepp_debug_flatten_count() - flattens_before;
    t_mark = now_s();
    for (uint32_t g = 0; g < G; g++)
        if (!errors[g].empty()) {
            fprintf(stderr, "ERROR: %s\n", errors[g].c_str());
            destroy_all();
            return 1;
        }
    for (uint32_t g = 0; g < G; g++) {            // the shards' imputed mutations, concatenated in sample order
        const uint32_t lo = shard_lo(g);
        for (size_t i = 0; i < shard_imp_cnt[g].size(); i++) imp_off[lo + i + 1] = shard_imp_cnt[g][i];
        imp_pos.insert(imp_pos.end(), shard_imp_pos[g].begin(), shard_imp_pos[g].end());
        imp_nuc.insert(imp_nuc.end(), shard_imp_nuc[g].begin(), shard_imp_nuc[g].end());
    }
    for (uint32_t q = 0; q < R; q++) imp_off[q + 1] += imp_off[q];

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

    // ---- the -p mode: per-node scores and the excess mutations of the optimal nodes (last column of
    // parsimony-scores.tsv, :555-574), computed chunk by chunk in output order: a chunk holds as many
    // samples as keep chunk * total_nodes under 2^28 values (1 GiB), whatever the number of samples --
    // the reference scores one sample at a time (:386-411) ------------------------------------------
    uint32_t chunk_rows = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(R ? R : 1, (1ull << 28) / std::max<size_t>(total_nodes, 1)));
    if (const char* e = getenv("WEPP_USHER_CHUNK_ROWS"))          // tests: force several chunks on a small tree
        chunk_rows = (uint32_t)std::max(1, atoi(e));
    auto score_chunk = [&](uint32_t g, ScoreChunk& c) {
        const uint32_t n = c.hi - c.lo;
        std::vector<uint32_t> off, words;
        gather_reads(read_off, read_word, indexes.data(), c.lo, c.hi, off, words);
        std::vector<uint32_t> bj(n), nb(n), fl(n);
        std::vector<int32_t> sd(n);
        c.node_sd.resize((size_t)n * total_nodes);
        if (wepp_place_batch(mats[g], off.data(), words.data(), n, bj.data(), sd.data(), nb.data(), fl.data(),
                             c.node_sd.data()) != WEPP_OK) {
            c.error = wepp_last_error();
            return;
        }
        std::vector<uint32_t> pair_read, pair_node;
        c.exc_first.assign(n + 1, 0);
        for (uint32_t i = 0; i < n; i++) {
            c.exc_first[i] = pair_read.size();
            const int32_t* nsd = c.node_sd.data() + (size_t)i * total_nodes;
            for (size_t k = 0; k < total_nodes; k++)
                if (nsd[k] == sd[i] && nsd[k] != 0) { pair_read.push_back(i); pair_node.push_back((uint32_t)k); }
        }
        c.exc_first[n] = pair_read.size();
        c.exc_off.assign(pair_read.size() + 1, 0);
        uint64_t cap = 0;
        for (int attempt = 0; attempt < 2; attempt++) {
            c.exc_pos.resize(cap + 1); c.exc_ref.resize(cap + 1); c.exc_par.resize(cap + 1); c.exc_mut.resize(cap + 1);
            int rc = wepp_excess_mutations(mats[g], off.data(), words.data(), n, (uint32_t)pair_read.size(),
                                           pair_read.data(), pair_node.data(), c.exc_off.data(), c.exc_pos.data(),
                                           c.exc_ref.data(), c.exc_par.data(), c.exc_mut.data(), cap);
            if (rc == WEPP_ELIMIT && attempt == 0 && c.exc_off.back() > cap) { cap = c.exc_off.back(); continue; }
            if (rc != WEPP_OK) c.error = wepp_last_error();
            break;
        }
    };

    FileCloser stats, scores;
    if (!outdir.empty()) {
        // (usher creates a missing output directory, usher_common.cpp:78-83)
        if (mkdir(outdir.c_str(), 0755) != 0 && errno != EEXIST) {
            fprintf(stderr, "ERROR: cannot create %s\n", outdir.c_str());
            destroy_all();
            return 1;
        }
        stats.f = fopen((outdir + "/placement_stats.tsv").c_str(), "w");                      // :303-304
        if (!stats.f) {
            fprintf(stderr, "ERROR: cannot write to %s\n", outdir.c_str());
            destroy_all();
            return 1;
        }
    }
    if (results) results->clear();
    std::vector<ScoreChunk> chunks(G);            // one chunk in flight per device
    uint32_t chunks_lo = 0, chunks_hi = 0;        // rows covered by `chunks`
    for (uint32_t qi = 0; qi < R; qi++) {
        const uint32_t q = indexes[qi];
        if (print_parsimony_scores && qi >= chunks_hi) {
            chunks_lo = qi;
            uint32_t used = 0;
            for (uint32_t g = 0; g < G; g++) {
                chunks[g].lo = (uint32_t)std::min<uint64_t>(R, (uint64_t)chunks_lo + (uint64_t)g * chunk_rows);
                chunks[g].hi = (uint32_t)std::min<uint64_t>(R, (uint64_t)chunks[g].lo + chunk_rows);
                chunks[g].error.clear();
                if (chunks[g].hi > chunks[g].lo) used = g + 1;
            }
            chunks_hi = chunks[used - 1].hi;
            run_on_devices([&](uint32_t g) { if (chunks[g].hi > chunks[g].lo) score_chunk(g, chunks[g]); });
            for (uint32_t g = 0; g < used; g++)
                if (!chunks[g].error.empty()) {
                    fprintf(stderr, "ERROR: %s\n", chunks[g].error.c_str());
                    destroy_all();
                    return 1;
                }
        }
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
                const ScoreChunk& ck = chunks[(qi - chunks_lo) / chunk_rows];
                const std::vector<uint64_t>& exc_off = ck.exc_off;
                const std::vector<int32_t>& exc_pos = ck.exc_pos;
                const std::vector<uint8_t>&exc_ref = ck.exc_ref, &exc_par = ck.exc_par, &exc_mut = ck.exc_mut;
                const int32_t* nsd = ck.node_sd.data() + (size_t)(qi - ck.lo) * total_nodes;
                size_t pair = ck.exc_first[qi - ck.lo];
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
    destroy_all();
    usher_last_timing.write_s = now_s() - t_mark;
    return 0;
}
