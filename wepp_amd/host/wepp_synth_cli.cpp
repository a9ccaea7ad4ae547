// wepp-synth: writes the synthetic workload of the bench as FILES the C++ host reads back -- a MAT as parsimony.proto
// (.pb or .pb.gz, save_mutation_annotated_tree) and samples as a VCF -- so that the .pb[.gz] loader, the Newick
// parser, the VCF reader and the flattener of wepp-usher run at the size the metric names (16 M nodes), where the
// reference reads UCSC's public MATs (src/mutation_annotated_tree.cpp:522-612).  The reference bundles no data
// (SURVEY.md F6); the generator is the library's own (wepp_gen_tree_create / wepp_gen_reads_create).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/wepp_place.h"
#include "mat.hpp"

static const char NUC_CH[16] = {'?', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'};

int main(int argc, char** argv) {
    uint32_t nodes = 100000, samples = 0, read_len = 150;
    uint64_t seed = 21, sample_seed = 22;
    double p_n = 0.005, p_sub = 0.001, p_hub = 0.0, p_back = 0.02;
    uint32_t depth_choices = 0;
    std::string pb, vcf;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return argv[++i]; };
        if (a == "--nodes") nodes = (uint32_t)atoll(next());
        else if (a == "--seed") seed = (uint64_t)atoll(next());
        else if (a == "--pb") pb = next();
        else if (a == "--samples") samples = (uint32_t)atoll(next());
        else if (a == "--sample-seed") sample_seed = (uint64_t)atoll(next());
        else if (a == "--read-len") read_len = (uint32_t)atoll(next());
        else if (a == "--p-n") p_n = atof(next());
        else if (a == "--p-sub") p_sub = atof(next());
        else if (a == "--vcf") vcf = next();
        else if (a == "--depth-choices") depth_choices = (uint32_t)atoll(next());     // tree shape: longer root paths
        else if (a == "--p-hub") p_hub = atof(next());                                // tree shape: polytomies
        else if (a == "--p-back") p_back = atof(next());
        else {
            fprintf(stderr, "usage: wepp-synth --nodes N [--seed S] --pb out.pb[.gz] [--samples K --vcf out.vcf[.gz] [--sample-seed S] "
                            "[--read-len L] [--p-n x] [--p-sub x]] [--depth-choices c] [--p-hub x] [--p-back x]\n");
            return 1;
        }
    }
    // the same parameters as wepp_amd.generate_tree(seed, nodes): the tree of the bench and of the tests
    wepp_gen_tree_params tp{seed, nodes, 29903, 0.25, 0.6, p_back, 0.0, 0.0, 0, depth_choices, p_hub, 0};
    wepp_gen_tree_t* gt = nullptr;
    if (wepp_gen_tree_create(&tp, &gt) != WEPP_OK) { fprintf(stderr, "ERROR: %s\n", wepp_last_error()); return 1; }
    wepp_tree_desc d{};
    wepp_gen_tree_desc(gt, &d);
    try {
        if (!pb.empty()) {
            std::vector<bool> has_child(d.n_nodes, false);
            for (uint32_t i = 0; i < d.n_nodes; i++)
                if (d.parent[i] >= 0) has_child[(size_t)d.parent[i]] = true;
            MAT::Tree T;
            std::vector<MAT::Node*> node(d.n_nodes, nullptr);
            for (uint32_t i = 0; i < d.n_nodes; i++) {          // (a parent's id is smaller than its children's)
                const std::string name = has_child[i] ? T.new_internal_node_id() : "s" + std::to_string(i);
                node[i] = d.parent[i] < 0 ? T.create_node(name, 0.0f) : T.create_node(name, node[(size_t)d.parent[i]], 1.0f);
                for (uint32_t k = d.mut_off[i]; k < d.mut_off[i + 1]; k++) {
                    MAT::Mutation m;
                    m.position = d.mut_pos[k];
                    m.ref_nuc = (int8_t)d.mut_ref[k];
                    m.par_nuc = (int8_t)(d.mut_par ? d.mut_par[k] : d.mut_ref[k]);
                    m.mut_nuc = (int8_t)d.mut_mut[k];
                    node[i]->mutations.push_back(m);
                }
            }
            MAT::save_mutation_annotated_tree(T, pb);
            fprintf(stderr, "wrote %s: %u nodes, %u mutations\n", pb.c_str(), d.n_nodes, d.mut_off[d.n_nodes]);
        }
        if (!vcf.empty() && samples) {
            // one sample per generated read: a window of a random leaf's genotype with errors and Ns (read_len = the
            // genome's length gives whole-genome samples)
            wepp_gen_reads_params rp{sample_seed, samples, read_len, std::max(read_len, 400u), std::max(read_len * 3 / 4, 1u), p_sub, p_n, 0.0};
            wepp_gen_reads_t* gr = nullptr;
            if (wepp_gen_reads_create(gt, &rp, &gr) != WEPP_OK) { fprintf(stderr, "ERROR: %s\n", wepp_last_error()); return 1; }
            uint32_t n = 0;
            const uint32_t *off = nullptr, *word = nullptr;
            wepp_gen_reads_get(gr, &n, &off, &word);
            struct Row { uint8_t ref = 0; std::string alts; std::map<uint32_t, int> gt; };   // sample -> allele number, -1 = missing
            std::map<uint32_t, Row> rows;
            for (uint32_t s = 0; s < n; s++)
                for (uint32_t k = off[s]; k < off[s + 1]; k++) {
                    const uint32_t w = word[k], pos = w & 0xFFFFFu, ref = (w >> 20) & 15u, a = (w >> 24) & 15u, miss = (w >> 28) & 1u;
                    Row& r = rows[pos];
                    r.ref = (uint8_t)ref;
                    if (miss && a == 15) { r.gt[s] = -1; continue; }
                    size_t at = r.alts.find(NUC_CH[a]);
                    if (at == std::string::npos) { at = r.alts.size(); r.alts.push_back(NUC_CH[a]); }
                    r.gt[s] = (int)at + 1;
                }
            std::string out = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT";
            for (uint32_t s = 0; s < n; s++) out += "\tsample_" + std::to_string(s);
            out += "\n";
            for (auto& pr : rows) {
                const Row& r = pr.second;
                out += "NC_045512v2\t" + std::to_string(pr.first) + "\t.\t" + NUC_CH[r.ref] + "\t";
                if (r.alts.empty()) out += ".";
                for (size_t i = 0; i < r.alts.size(); i++) { if (i) out += ","; out += r.alts[i]; }
                out += "\t.\t.\t.\tGT";
                auto it = r.gt.begin();
                for (uint32_t s = 0; s < n; s++) {
                    if (it != r.gt.end() && it->first == s) { out += it->second < 0 ? "\t." : "\t" + std::to_string(it->second); ++it; }
                    else out += "\t0";
                }
                out += "\n";
            }
            const bool gz = vcf.size() > 3 && vcf.compare(vcf.size() - 3, 3, ".gz") == 0;
            if (gz) {
                gzFile f = gzopen(vcf.c_str(), "wb1");
                if (!f || gzwrite(f, out.data(), (unsigned)out.size()) != (int)out.size()) { fprintf(stderr, "ERROR: could not write %s\n", vcf.c_str()); return 1; }
                gzclose(f);
            } else {
                FILE* f = fopen(vcf.c_str(), "wb");
                if (!f || fwrite(out.data(), 1, out.size(), f) != out.size()) { fprintf(stderr, "ERROR: could not write %s\n", vcf.c_str()); return 1; }
                fclose(f);
            }
            fprintf(stderr, "wrote %s: %u samples, %zu rows\n", vcf.c_str(), n, rows.size());
            wepp_gen_reads_destroy(gr);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    wepp_gen_tree_destroy(gt);
    return 0;
}
