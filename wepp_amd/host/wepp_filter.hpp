// wepp_filter.hpp -- host-side mirror of the slice of WEPP's own interface that feeds and
// consumes wepp_filter::cartesian_map (/root/reference/src/WEPP/): raw_read, the reads .pb
// loader, read masking, the condensed tree, and the call itself on top of wepp_epp_map.
// Same names and argument meaning as the reference; errors throw MAT::mat_error.
#pragma once
#include <array>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "mat.hpp"

static constexpr int NUM_RANGE_BINS = 50;          // src/WEPP/config.hpp:13
static constexpr int MAX_CACHED_EPP_SIZE = 2048;   // src/WEPP/config.hpp:9

struct raw_read {                                   // src/WEPP/read.hpp:8-14
    std::string read;
    std::vector<MAT::Mutation> mutations;
    int start = 0;
    int end = 0;
    int degree = 0;
};

// FASTA -> upper-cased sequence without the header line (dataset.hpp:179-203)
std::string load_reference(std::string const& fasta_filename);
// third column of every line of mask.bed; a missing file means no mask (dataset.hpp:90-113)
std::vector<int> load_masked_sites(std::string const& bed_filename);

// sam.proto message `sam` -> raw_reads (sam2pb.cpp:489-549): start = start_idx (1-based),
// end = start + len(content) - 1, one mutation wherever content differs from the reference and
// is not '_'; 'N' is missing.  reverse_merge receives the column merge table (:539-545).
std::vector<raw_read> load_reads_from_proto(std::string const& reference, std::string const& filename,
                                            std::unordered_map<std::string, std::vector<std::string>>& reverse_merge);
// the writer side of the same message (sam::dump_proto, sam2pb.cpp:111-147), for fixtures:
// reads given as (name, 1-based start, aligned content over ACGTN_, degree)
struct sam_read_record { std::string name; int start_idx; std::string content; int degree; };
void dump_reads_proto(std::vector<sam_read_record> const& reads, std::string const& filename);

// arena::arena, arena.hpp:58-72: mutations at masked sites are removed from the reads
void mask_reads(std::vector<raw_read>& reads, std::vector<int> const& masked_sites);
// arena::site_read_map, arena.hpp:157-175: sites covered by a non-N, non-masked read base
std::unordered_set<int> site_read_map(std::vector<raw_read> const& reads, std::vector<int> const& masked_sites);
// util.cpp:79-133: keeps the mutations at covered sites; a node left without mutations is
// merged into its nearest kept ancestor (node_mappings: condensed node -> original nodes)
MAT::Tree create_condensed_tree(MAT::Node* ref_root, const std::unordered_set<int>& site_read_map,
                                std::unordered_map<MAT::Node*, std::vector<MAT::Node*>>& node_mappings);

// what wepp_filter::cartesian_map leaves behind (initial_filter.cpp:140-239, without the final
// sort): haplotypes in arena order = pre-order of the condensed tree (arena.cpp:3-55)
struct cartesian_map_result {
    std::vector<MAT::Node*> haplotypes;                          // condensed_source of haplotype k
    std::vector<double> score, dist_divergence;                  // haplotype::score (= orig_score), ::dist_divergence
    std::vector<std::array<int, NUM_RANGE_BINS>> mapped_read_counts;
    std::vector<int> max_parismony, parsimony_multiplicity;      // per read (:203-204)
    std::vector<std::vector<int>> epp_positions_cache;           // per read: arena indices, sorted (:205-210)
};
// returns 0, or 1 after printing the error (the reference's convention for this layer)
int cartesian_map(MAT::Tree& condensed, const std::vector<raw_read>& reads, size_t genome_size,
                  cartesian_map_result& out, int device = 0);
