// wepp_epp_cli.cpp -- `wepp-epp`: the data path of `wepp detectPeaks` up to and including
// wepp_filter::cartesian_map, on files: MAT .pb[.gz] + reads .pb (sam.proto, as written by
// `wepp sam2PB`) + reference FASTA [+ mask.bed] -> haplotype scores and per-read placements.
//   wepp-epp -i tree.pb -r reads.pb -f ref.fa [-m mask.bed] -d outdir [--device N] [--dump]
// --dump prints what the loaders and the condensing step produced and exits (no GPU needed).
// Output: <outdir>/haplotype_scores.tsv (arena order: id, score, dist_divergence, sources),
//         <outdir>/read_placements.tsv  (read, start, end, degree, parsimony, epps).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "wepp_filter.hpp"

int main(int argc, char** argv) {
    std::string mat_f, reads_f, ref_f, mask_f, outdir = ".";
    int device = 0;
    bool dump = false;
    for (int i = 1; i < argc; i++) {
        auto need = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "ERROR: %s needs a value\n", flag); exit(1); }
            return argv[++i];
        };
        if (!strcmp(argv[i], "-i")) mat_f = need("-i");
        else if (!strcmp(argv[i], "-r")) reads_f = need("-r");
        else if (!strcmp(argv[i], "-f")) ref_f = need("-f");
        else if (!strcmp(argv[i], "-m")) mask_f = need("-m");
        else if (!strcmp(argv[i], "-d")) outdir = need("-d");
        else if (!strcmp(argv[i], "--device")) device = atoi(need("--device"));
        else if (!strcmp(argv[i], "--dump")) dump = true;
        else { fprintf(stderr, "usage: wepp-epp -i tree.pb -r reads.pb -f ref.fa [-m mask.bed] -d outdir [--device N]\n"); return 1; }
    }
    if (mat_f.empty() || reads_f.empty() || ref_f.empty()) {
        fprintf(stderr, "usage: wepp-epp -i tree.pb -r reads.pb -f ref.fa [-m mask.bed] -d outdir [--device N]\n");
        return 1;
    }
    try {
        std::string reference = load_reference(ref_f);
        MAT::Tree T = MAT::load_mutation_annotated_tree(mat_f);
        T.uncondense_leaves();                                            // dataset.hpp:222
        std::unordered_map<std::string, std::vector<std::string>> reverse_merge;
        std::vector<raw_read> reads = load_reads_from_proto(reference, reads_f, reverse_merge);
        std::vector<int> mask = mask_f.empty() ? std::vector<int>() : load_masked_sites(mask_f);
        mask_reads(reads, mask);                                          // arena.hpp:60-72
        std::unordered_map<MAT::Node*, std::vector<MAT::Node*>> mappings;
        MAT::Tree condensed = create_condensed_tree(T.root, site_read_map(reads, mask), mappings);
        fprintf(stderr, "%zu reads, %zu nodes, %zu haplotypes after condensing\n", reads.size(), T.size(), condensed.size());
        if (dump) {
            for (auto& r : reads) {
                printf("read %s %d %d %d", r.read.c_str(), r.start, r.end, r.degree);
                for (auto& m : r.mutations) printf(" %d:%d:%d", m.position, (int)m.ref_nuc, (int)m.mut_nuc);
                printf("\n");
            }
            for (MAT::Node* n : condensed.depth_first_expansion()) {
                printf("hap %s %s %zu", n->identifier.c_str(), n->parent ? n->parent->identifier.c_str() : "-", mappings[n].size());
                for (auto& m : n->mutations) printf(" %d:%d:%d", m.position, (int)m.ref_nuc, (int)m.mut_nuc);
                printf("\n");
            }
            for (MAT::Node* n : T.depth_first_expansion())
                printf("node %s %s\n", n->identifier.c_str(), n->parent ? n->parent->identifier.c_str() : "-");
            return 0;
        }
        cartesian_map_result res;
        if (cartesian_map(condensed, reads, reference.size(), res, device) != 0) return 1;
        FILE* f = fopen((outdir + "/haplotype_scores.tsv").c_str(), "w");
        if (!f) { fprintf(stderr, "ERROR: cannot write into %s\n", outdir.c_str()); return 1; }
        fprintf(f, "haplotype\tscore\tdist_divergence\tsources\n");
        for (size_t k = 0; k < res.haplotypes.size(); k++)
            fprintf(f, "%s\t%.12g\t%.12g\t%zu\n", res.haplotypes[k]->identifier.c_str(), res.score[k],
                    res.dist_divergence[k], mappings[res.haplotypes[k]].size());
        fclose(f);
        f = fopen((outdir + "/read_placements.tsv").c_str(), "w");
        if (!f) { fprintf(stderr, "ERROR: cannot write into %s\n", outdir.c_str()); return 1; }
        fprintf(f, "read\tstart\tend\tdegree\tparsimony\tepps\n");
        for (size_t r = 0; r < reads.size(); r++)
            fprintf(f, "%s\t%d\t%d\t%d\t%d\t%d\n", reads[r].read.c_str(), reads[r].start, reads[r].end, reads[r].degree,
                    res.max_parismony[r], res.parsimony_multiplicity[r]);
        fclose(f);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
