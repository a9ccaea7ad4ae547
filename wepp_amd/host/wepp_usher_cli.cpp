// wepp-usher: `usher -i tree.pb[.gz] -v samples.vcf[.gz] -n -d outdir [-p]` on the GPU
// (the placement-only invocation of /root/reference/src/usher.cpp:141-183).
// A thin driver around usher_place_samples; also `--dump` prints what the loaders
// parsed (used by the CPU tests, needs no GPU).
#include <sys/resource.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "usher_place.hpp"

static void usage() {
    fprintf(stderr, "usage: wepp-usher -i <mat.pb[.gz]> -v <samples.vcf[.gz]> [-d <outdir>] [-p] [-e max_uncertainty] "
                    "[-E max_parsimony] [-s|-S|-A] [-r] [--device N | --devices 0,1,...] [--dump] [--report]\n"
                    "       --report prints one JSON line to stdout: seconds per phase (load, VCF, tree description, flatten, upload, "
                    "placement, writers), full flattens, peak resident memory\n"
                    "       -n/--no-add is required: samples are placed on the tree as given, never added to it\n");
}

int main(int argc, char** argv) {
    std::string pb, vcf, outdir = ".";
    bool print_scores = false, dump = false, sort1 = false, sort2 = false, sort3 = false, reverse_sort = false;
    bool no_add = false, report = false, load_only = false;
    std::vector<int> devices;
    uint32_t max_uncertainty = 1000000, max_parsimony = 1000000;   // usher.cpp:77-80 defaults
    int device = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); exit(1); } return argv[++i]; };
        if (a == "-i" || a == "--load-mutation-annotated-tree") pb = next();
        else if (a == "-v" || a == "--vcf") vcf = next();
        else if (a == "-d" || a == "--outdir") outdir = next();
        else if (a == "-p" || a == "--write-parsimony-scores-per-node") print_scores = true;
        else if (a == "-e" || a == "--max-uncertainty-per-sample") max_uncertainty = (uint32_t)atoi(next());
        else if (a == "-E" || a == "--max-parsimony-per-sample") max_parsimony = (uint32_t)atoi(next());
        else if (a == "-n" || a == "--no-add") no_add = true;
        else if (a == "-s" || a == "--sort-before-placement-1") sort1 = true;
        else if (a == "-S" || a == "--sort-before-placement-2") sort2 = true;
        else if (a == "-A" || a == "--sort-before-placement-3") sort3 = true;
        else if (a == "-r" || a == "--reverse-sort") reverse_sort = true;
        else if (a == "--device") device = atoi(next());
        else if (a == "--devices") {                              // one host thread + one handle per listed GPU
            std::vector<std::string> ids;
            MAT::string_split(next(), ',', ids);
            for (auto& d : ids) devices.push_back(atoi(d.c_str()));
        }
        else if (a == "--dump") dump = true;
        else if (a == "--report") report = true;
        else if (a == "--load-only") load_only = true;
        else { usage(); return 1; }
    }
    if (pb.empty() || vcf.empty()) { usage(); return 1; }
    try {
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        MAT::Tree T = MAT::load_mutation_annotated_tree(pb);
        const double t_load = now() - t0;
        std::vector<Missing_Sample> missing_samples;
        MAT::read_vcf(&T, vcf, missing_samples);
        const double t_vcf = now() - t0 - t_load;
        if (load_only) {          // the loaders alone (no GPU needed): what a 16 M-node .pb.gz costs before the first kernel
            struct rusage ru{};
            getrusage(RUSAGE_SELF, &ru);
            printf("{\"nodes\": %zu, \"samples\": %zu, \"load_pb_s\": %.3f, \"read_vcf_s\": %.3f, \"peak_rss_mb\": %.1f}\n", T.size(),
                   missing_samples.size(), t_load, t_vcf, ru.ru_maxrss / 1024.0);
            return 0;
        }
        if (dump) {
            auto bfs = T.breadth_first_expansion();
            printf("nodes %zu\n", bfs.size());
            for (auto n : bfs) {
                printf("node %s parent %s muts", n->identifier.c_str(), n->parent ? n->parent->identifier.c_str() : "-");
                for (auto& m : n->mutations) printf(" %d:%d:%d:%d", m.position, m.ref_nuc, m.par_nuc, m.mut_nuc);
                printf("\n");
            }
            for (auto& s : missing_samples) {
                printf("sample %s", s.name.c_str());
                for (auto& m : s.mutations) printf(" %d:%d:%d:%d", m.position, m.ref_nuc, m.mut_nuc, (int)m.is_missing);
                printf("\n");
            }
            return 0;
        }
        std::vector<std::string> low_conf;
        if (!no_add) {
            // reference usher without -n adds every placed sample to the tree (usher_common.cpp:649-762), so later
            // placements depend on earlier ones; that sequential mode is not implemented here
            fprintf(stderr, "ERROR: wepp-usher places samples on a fixed tree only; pass -n/--no-add (adding samples to the tree, usher's default, is not implemented).\n");
            return 1;
        }
        if (sort1 && sort2) {
            fprintf(stderr, "ERROR: Can't use sort-before-placement-1 and sort-before-placement-2 simultaneously. Please specify only one.\n");
            return 1;                                               // usher_common.cpp:14-71 style validation
        }
        if (devices.empty()) devices.push_back(device);
        const int rc = usher_place_samples(outdir, max_uncertainty, max_parsimony, print_scores, missing_samples, low_conf, &T,
                                           devices, nullptr, sort1, sort2, sort3, reverse_sort);
        if (report) {
            struct rusage ru{};
            getrusage(RUSAGE_SELF, &ru);
            const usher_place_timing& t = usher_last_timing;
            printf("{\"nodes\": %zu, \"samples\": %zu, \"devices\": %zu, \"load_pb_s\": %.3f, \"read_vcf_s\": %.3f, \"describe_s\": %.3f, "
                   "\"flatten_s\": %.3f, \"upload_s\": %.3f, \"place_s\": %.3f, \"write_s\": %.3f, \"total_s\": %.3f, \"flattens\": %llu, "
                   "\"peak_rss_mb\": %.1f, \"rc\": %d}\n",
                   T.size(), missing_samples.size(), devices.size(), t_load, t_vcf, t.describe_s, t.flatten_s, t.upload_s, t.place_s, t.write_s,
                   now() - t0, (unsigned long long)t.flattens, ru.ru_maxrss / 1024.0, rc);
        }
        return rc;
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
