// mat.hpp -- host-side mirror of the slice of the reference's MAT interface that
// the placement path touches, so that code written against
// /root/reference/src/mutation_annotated_tree.hpp (Mutation / Node / Tree,
// load_mutation_annotated_tree, read_vcf) can drive the GPU placer unchanged.
// Same names, same argument meaning; failures throw mat_error instead of the
// reference's exit(1) (mutation_annotated_tree.cpp:474,514,533) so that nothing
// below the C-ABI terminates the process.  Independent implementation: no TBB,
// no Boost, no libprotobuf (the .pb wire format is parsed by hand, gzip via zlib).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace Mutation_Annotated_Tree {

struct mat_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// IUPAC character <-> 4-bit mask (A=1 C=2 G=4 T=8), mutation_annotated_tree.cpp:19-139.
// Like the reference, 'V' maps to N (its switch falls through, :65-71) and only
// a,c,g,t,n are accepted in lower case.
int8_t get_nuc_id(char nuc);
int8_t get_nuc_id(const std::vector<int8_t>& nuc_vec);   // sum of 1 << idx, :77-85
char get_nuc(int8_t nuc_id);

// mutation_annotated_tree.hpp:44-78 (chrom is carried but ignored, like upstream)
struct Mutation {
    std::string chrom;
    int position = 0;
    int8_t ref_nuc = 0, par_nuc = 0, mut_nuc = 0;
    bool is_missing = false;
    bool operator<(const Mutation& m) const { return position < m.position; }
    bool is_masked() const { return position < 0; }
    std::string get_string() const;
};

class Node {   // mutation_annotated_tree.hpp:80-102
  public:
    size_t level = 0;
    float branch_length = -1.0f;
    std::string identifier;
    Node* parent = nullptr;
    std::vector<Node*> children;
    std::vector<Mutation> mutations;
    size_t dfs_idx = 0, dfs_end_idx = 0;
    bool is_leaf() const { return children.empty(); }
    bool is_root() const { return parent == nullptr; }
    void add_mutation(Mutation mut);   // :720-746: sorted insert / overwrite / reversal removes
};

class Tree {   // mutation_annotated_tree.hpp:104-152
  public:
    Tree() = default;
    Tree(const Tree&) = delete;
    Tree& operator=(const Tree&) = delete;
    Tree(Tree&& o) noexcept;
    Tree& operator=(Tree&& o) noexcept;
    ~Tree();

    Node* root = nullptr;
    size_t curr_internal_node = 0;
    std::unordered_map<std::string, std::vector<std::string>> condensed_nodes;
    std::unordered_set<std::string> condensed_leaves;

    std::string new_internal_node_id() { return "node_" + std::to_string(++curr_internal_node); }
    Node* create_node(std::string const& identifier, float branch_length = -1.0f);
    Node* create_node(std::string const& identifier, Node* par, float branch_length = -1.0f);
    Node* create_node(std::string const& identifier, std::string const& parent_id, float branch_length = -1.0f);
    Node* get_node(std::string const& identifier) const;
    size_t get_num_leaves(Node* node = nullptr) const;
    std::vector<Node*> breadth_first_expansion(std::string nid = "") const;
    std::vector<Node*> depth_first_expansion(Node* node = nullptr) const;   // also sets dfs_idx / dfs_end_idx
    size_t size() const { return all_nodes.size(); }
    void reserve(size_t n_nodes);          // room in the name table for n_nodes nodes (no rehash while they are created)
    // condensed identical-sequence leaves back into separate leaves (mutation_annotated_tree.cpp:1224-1272)
    void uncondense_leaves();

  private:
    std::unordered_map<std::string, Node*> all_nodes;
    void clear();
};

void string_split(std::string const& s, char delim, std::vector<std::string>& words);
void string_split(std::string const& s, std::vector<std::string>& words);

// Newick -> Tree: internal nodes are renamed node_<k> in order of appearance and
// their labels are discarded (mutation_annotated_tree.cpp:415-508).
Tree create_tree_from_newick_string(std::string const& newick_string);

// parsimony.proto `data` message (optionally gzipped) -> Tree, with the three
// normalisations of mutation_annotated_tree.cpp:556-596: mutations with
// mut_nuc == par_nuc are dropped, position < 0 = masked with zero nucleotides,
// unsorted lists are sorted (warning on stderr).
Tree load_mutation_annotated_tree(std::string const& filename);
// writes the same message (used by the tests to make fixtures): uncompressed
void save_mutation_annotated_tree(const Tree& tree, std::string const& filename);
std::string get_newick_string(const Tree& T);

}  // namespace Mutation_Annotated_Tree

namespace MAT = Mutation_Annotated_Tree;

// usher_graph.hpp:34-54
struct Missing_Sample {
    std::string name;
    std::vector<MAT::Mutation> mutations;
    size_t num_ambiguous = 0;
    explicit Missing_Sample(std::string sample_name) : name(std::move(sample_name)) {}
    bool operator==(const Missing_Sample& other) const { return name == other.name; }
    bool operator<(const Missing_Sample& other) const { return num_ambiguous < other.num_ambiguous; }
};

namespace Mutation_Annotated_Tree {
// VCF (optionally gzipped) -> Missing_Sample list for samples not yet in the tree
// (mutation_annotated_tree.cpp:2033-2130: numeric GT > 0 picks ALT[id-1][0],
// non-numeric GT or an N allele = missing).
void read_vcf(Tree* T, std::string const& vcf_filename, std::vector<Missing_Sample>& missing_samples);
}  // namespace Mutation_Annotated_Tree
