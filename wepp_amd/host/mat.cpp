// mat.cpp -- see mat.hpp.  Independent implementation of the reference's MAT
// interface subset (citations are to /root/reference/src/mutation_annotated_tree.cpp).
#include "mat.hpp"
#include "pbwire.hpp"

#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <queue>
#include <sstream>

namespace Mutation_Annotated_Tree {

// ---- nucleotide codec ---------------------------------------------------------
int8_t get_nuc_id(char nuc) {
    static const struct { char c; int8_t m; } table[] = {
        {'A', 1}, {'a', 1}, {'C', 2}, {'c', 2}, {'G', 4}, {'g', 4}, {'T', 8}, {'t', 8},
        {'R', 5}, {'Y', 10}, {'S', 6}, {'W', 9}, {'K', 12}, {'M', 3}, {'B', 14}, {'D', 13}, {'H', 11},
        // 'V' deliberately absent: upstream's case 'V' falls through to N (:65-71)
    };
    for (auto& e : table)
        if (e.c == nuc) return e.m;
    return 15;
}

int8_t get_nuc_id(const std::vector<int8_t>& nuc_vec) {
    int8_t ret = 0;
    for (int8_t n : nuc_vec) {
        if (n < 0 || n > 3) throw mat_error("ERROR: nucleotide index outside 0..3");
        ret = (int8_t)(ret + (int8_t)(1 << n));
    }
    return ret;
}

char get_nuc(int8_t nuc_id) {
    static const char iupac[16] = {'N', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'};
    return (nuc_id >= 1 && nuc_id <= 15) ? iupac[(int)nuc_id] : 'N';
}

std::string Mutation::get_string() const {
    if (is_masked()) return "MASKED";
    return std::string(1, get_nuc(par_nuc)) + std::to_string(position) + std::string(1, get_nuc(mut_nuc));
}

// ---- Node -----------------------------------------------------------------------
void Node::add_mutation(Mutation mut) {
    auto it = std::lower_bound(mutations.begin(), mutations.end(), mut);
    if (it != mutations.end() && it->position == mut.position) {
        if (it->par_nuc != mut.mut_nuc) it->mut_nuc = mut.mut_nuc;   // later allele replaces the earlier one
        else mutations.erase(it);                                     // reversal to the parent allele
    } else {
        mutations.insert(it, std::move(mut));
    }
}

// ---- Tree -----------------------------------------------------------------------
void Tree::clear() {
    for (auto& kv : all_nodes) delete kv.second;
    all_nodes.clear();
    root = nullptr;
}
Tree::~Tree() { clear(); }
Tree::Tree(Tree&& o) noexcept { *this = std::move(o); }
Tree& Tree::operator=(Tree&& o) noexcept {
    if (this != &o) {
        clear();
        root = o.root;
        curr_internal_node = o.curr_internal_node;
        all_nodes = std::move(o.all_nodes);
        condensed_nodes = std::move(o.condensed_nodes);
        condensed_leaves = std::move(o.condensed_leaves);
        o.root = nullptr;
        o.all_nodes.clear();
    }
    return *this;
}

Node* Tree::create_node(std::string const& identifier, float branch_length) {
    clear();                                   // creating a root starts a new tree (:854-863)
    Node* n = new Node();
    n->identifier = identifier;
    n->branch_length = branch_length;
    root = n;
    all_nodes[identifier] = n;
    return n;
}

Node* Tree::create_node(std::string const& identifier, Node* par, float branch_length) {
    if (!par) throw mat_error("create_node: null parent for " + identifier);
    // (one hash lookup: the slot is taken first, the node made only when the name is new)
    auto slot = all_nodes.try_emplace(identifier, nullptr);
    if (!slot.second) throw mat_error("Error: " + identifier + " already in the tree!");
    Node* n = new Node();
    n->identifier = identifier;
    n->branch_length = branch_length;
    n->parent = par;
    n->level = par->level + 1;
    slot.first->second = n;
    par->children.push_back(n);
    return n;
}

void Tree::reserve(size_t n_nodes) { all_nodes.reserve(n_nodes); }

Node* Tree::create_node(std::string const& identifier, std::string const& parent_id, float branch_length) {
    return create_node(identifier, get_node(parent_id), branch_length);
}

Node* Tree::get_node(std::string const& identifier) const {
    auto it = all_nodes.find(identifier);
    return it == all_nodes.end() ? nullptr : it->second;
}

size_t Tree::get_num_leaves(Node* node) const {
    if (!node) node = root;
    if (!node) return 0;
    size_t leaves = 0;
    std::vector<Node*> st{node};
    while (!st.empty()) {
        Node* c = st.back();
        st.pop_back();
        if (c->is_leaf()) leaves++;
        for (Node* k : c->children) st.push_back(k);
    }
    return leaves;
}

std::vector<Node*> Tree::breadth_first_expansion(std::string nid) const {
    std::vector<Node*> order;
    Node* start = nid.empty() ? root : get_node(nid);
    if (!start) return order;
    order.push_back(start);
    for (size_t head = 0; head < order.size(); head++)
        for (Node* c : order[head]->children) order.push_back(c);
    return order;
}

std::vector<Node*> Tree::depth_first_expansion(Node* node) const {
    std::vector<Node*> order;
    if (!node) node = root;
    if (!node) return order;
    // pre-order with an explicit stack of (node, next child) frames
    std::vector<std::pair<Node*, size_t>> st;
    node->dfs_idx = 0;
    order.push_back(node);
    st.emplace_back(node, 0);
    while (!st.empty()) {
        auto& fr = st.back();
        if (fr.second < fr.first->children.size()) {
            Node* c = fr.first->children[fr.second++];
            c->dfs_idx = order.size();
            order.push_back(c);
            st.emplace_back(c, 0);
        } else {
            fr.first->dfs_end_idx = order.size();
            st.pop_back();
        }
    }
    return order;
}

// ---- strings --------------------------------------------------------------------
void Tree::uncondense_leaves() {
    auto rename = [&](Node* n, std::string id) {
        all_nodes.erase(n->identifier);
        n->identifier = std::move(id);
        all_nodes[n->identifier] = n;
    };
    for (auto& entry : condensed_nodes) {
        const std::vector<std::string>& samples = entry.second;
        Node* n = get_node(entry.first);
        if (!n || samples.empty()) continue;
        // A condensed leaf that carries mutations of its own stays as an internal node above all of its
        // samples; one without mutations turns into its first sample and the others become its siblings
        // (children of the root itself when the condensed node is the root).
        const bool stays_internal = samples.size() > 1 && !n->mutations.empty();
        Node* attach_to = stays_internal ? n : (n->parent ? n->parent : n);
        const float branch = stays_internal ? -1.0f : n->branch_length;
        rename(n, stays_internal ? new_internal_node_id() : samples.front());
        for (size_t k = stays_internal ? 0 : 1; k < samples.size(); k++) create_node(samples[k], attach_to, branch);
    }
    condensed_nodes.clear();
    condensed_leaves.clear();
}

void string_split(std::string const& s, char delim, std::vector<std::string>& words) {
    size_t start = 0;
    for (;;) {
        size_t end = s.find(delim, start);
        if (end == std::string::npos) break;
        words.emplace_back(s, start, end - start);
        start = end + 1;
    }
    if (start < s.size()) words.emplace_back(s, start, std::string::npos);
}

void string_split(std::string const& s, std::vector<std::string>& words) {
    std::istringstream ss(s);
    std::string w;
    while (ss >> w) words.push_back(w);
}

// ---- Newick -----------------------------------------------------------------------
Tree create_tree_from_newick_string(std::string const& nw) {
    Tree T;
    std::vector<Node*> open;   // internal nodes whose child list is still being read
    size_t i = 0;
    const size_t n = nw.size();
    // (strchr finds the terminator of its pattern for a NUL byte: a damaged file must not pass for a delimiter)
    auto one_of = [](const char* set, char c) { return c != '\0' && strchr(set, c) != nullptr; };
    auto read_length = [&](float& len) {   // optional ":<number>"
        if (i < n && nw[i] == ':') {
            size_t j = ++i;
            while (i < n && (isdigit((unsigned char)nw[i]) || one_of(".eE+-", nw[i]))) i++;
            if (i > j) {
                const std::string num = nw.substr(j, i - j);
                char* endp = nullptr;
                const float v = std::strtof(num.c_str(), &endp);
                if (endp == num.c_str() || *endp != '\0') throw mat_error("ERROR: incorrect Newick format (branch length '" + num + "')!");
                len = v;
            }
        }
    };
    {
        // every node but the root follows a '(' or a ',': the name table is sized once (at 16 M nodes the rehashes of a
        // growing table were a third of the parse)
        size_t nodes = 1;
        for (char ch : nw) nodes += (ch == '(' || ch == ',') ? 1 : 0;
        T.reserve(nodes);
    }
    bool any = false;
    while (i < n) {
        char ch = nw[i];
        if (ch == '(') {
            std::string nid = T.new_internal_node_id();
            Node* nd = open.empty() ? T.create_node(nid) : T.create_node(nid, open.back());
            open.push_back(nd);
            any = true;
            i++;
        } else if (ch == ')') {
            if (open.empty()) throw mat_error("ERROR: incorrect Newick format!");
            Node* nd = open.back();
            open.pop_back();
            i++;
            // the label of an internal node is discarded (upstream keeps node_<k>)
            while (i < n && !one_of(",:();", nw[i])) i++;
            read_length(nd->branch_length);
        } else if (ch == ',' || ch == ';' || isspace((unsigned char)ch)) {
            i++;
        } else {
            size_t j = i;
            while (i < n && !one_of(",:();", nw[i])) i++;
            if (i == j) throw mat_error("ERROR: incorrect Newick format!");    // (a ':' without a label)
            std::string name = nw.substr(j, i - j);
            Node* leaf = open.empty() ? T.create_node(name) : T.create_node(name, open.back());
            any = true;
            read_length(leaf->branch_length);
        }
    }
    if (!open.empty()) throw mat_error("ERROR: incorrect Newick format!");
    if (!any) fprintf(stderr, "WARNING: Tree found empty!\n");
    return T;
}

std::string get_newick_string(const Tree& T) {
    std::string out;
    if (!T.root) return ";";
    std::vector<std::pair<Node*, size_t>> st;
    st.emplace_back(T.root, 0);
    if (!T.root->is_leaf()) out += '(';
    else out += T.root->identifier;
    while (!st.empty()) {
        auto& fr = st.back();
        Node* nd = fr.first;
        if (fr.second < nd->children.size()) {
            if (fr.second) out += ',';
            Node* c = nd->children[fr.second++];
            if (c->is_leaf()) out += c->identifier;
            else out += '(';
            st.emplace_back(c, 0);
        } else {
            if (!nd->is_leaf()) out += ")" + nd->identifier;
            st.pop_back();
        }
    }
    return out + ";";
}

// ---- protobuf wire format (parsimony.proto) ---------------------------------------------
using pbwire::slurp;
using pbwire::Wire;
using pbwire::put_varint;
using pbwire::put_len;
namespace {
struct PbMut { int32_t position = 0, ref_nuc = 0, par_nuc = 0; std::vector<int8_t> mut_nuc; std::string chrom; };

PbMut parse_mut(Wire w) {
    PbMut m;
    while (!w.eof()) {
        uint64_t key = w.varint();
        uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
        if (field == 1 && wt == 0) m.position = (int32_t)w.varint();
        else if (field == 2 && wt == 0) m.ref_nuc = (int32_t)w.varint();
        else if (field == 3 && wt == 0) m.par_nuc = (int32_t)w.varint();
        else if (field == 4 && wt == 0) m.mut_nuc.push_back((int8_t)w.varint());
        else if (field == 4 && wt == 2) { Wire s = w.sub(); while (!s.eof()) m.mut_nuc.push_back((int8_t)s.varint()); }
        else if (field == 5 && wt == 2) { Wire s = w.sub(); m.chrom.assign((const char*)s.p, (size_t)(s.end - s.p)); }
        else w.skip(wt);
    }
    return m;
}

}  // namespace

Tree load_mutation_annotated_tree(std::string const& filename) {
    // WEPP_LOAD_TIMING=1: seconds per phase to stderr (read + inflate, message scan, Newick, mutation lists)
    const bool timing = getenv("WEPP_LOAD_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const double t = now();
        fprintf(stderr, "[load] %s %.2f s\n", what, t - t_last);
        t_last = t;
    };
    std::string raw = slurp(filename, "mutation-annotated tree");
    lap("read + inflate");
    Wire top{(const uint8_t*)raw.data(), (const uint8_t*)raw.data() + raw.size()};
    std::string newick;
    std::vector<Wire> node_lists;
    std::vector<std::pair<std::string, std::vector<std::string>>> condensed;
    while (!top.eof()) {
        uint64_t key = top.varint();
        uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
        if (field == 1 && wt == 2) { Wire s = top.sub(); newick.assign((const char*)s.p, (size_t)(s.end - s.p)); }
        else if (field == 2 && wt == 2) node_lists.push_back(top.sub());
        else if (field == 3 && wt == 2) {
            Wire c = top.sub();
            std::pair<std::string, std::vector<std::string>> cn;
            while (!c.eof()) {
                uint64_t k2 = c.varint();
                if ((k2 >> 3) == 1 && (k2 & 7) == 2) { Wire s = c.sub(); cn.first.assign((const char*)s.p, (size_t)(s.end - s.p)); }
                else if ((k2 >> 3) == 2 && (k2 & 7) == 2) { Wire s = c.sub(); cn.second.emplace_back((const char*)s.p, (size_t)(s.end - s.p)); }
                else c.skip((uint32_t)(k2 & 7));
            }
            condensed.push_back(std::move(cn));
        } else top.skip(wt);
    }
    lap("message scan");
    Tree tree = create_tree_from_newick_string(newick);
    lap("Newick");
    auto dfs = tree.depth_first_expansion();
    if (node_lists.size() < dfs.size())
        throw mat_error("ERROR: .pb holds " + std::to_string(node_lists.size()) + " mutation lists for " +
                        std::to_string(dfs.size()) + " nodes");
    for (size_t idx = 0; idx < dfs.size(); idx++) {
        Node* node = dfs[idx];
        Wire lst = node_lists[idx];
        while (!lst.eof()) {
            uint64_t key = lst.varint();
            if ((key >> 3) != 1 || (key & 7) != 2) { lst.skip((uint32_t)(key & 7)); continue; }
            PbMut pm = parse_mut(lst.sub());
            Mutation m;
            m.chrom = pm.chrom;
            m.position = pm.position;
            if (!m.is_masked()) {
                if (pm.ref_nuc < 0 || pm.ref_nuc > 3 || pm.par_nuc < 0 || pm.par_nuc > 3)
                    throw mat_error("ERROR: corrupt mutation-annotated tree: nucleotide index outside 0..3 at position " +
                                    std::to_string(pm.position));
                m.ref_nuc = (int8_t)(1 << pm.ref_nuc);
                m.par_nuc = (int8_t)(1 << pm.par_nuc);
                m.mut_nuc = get_nuc_id(pm.mut_nuc);
                if (m.mut_nuc != m.par_nuc) node->add_mutation(m);   // :580-582
            } else {
                m.ref_nuc = m.par_nuc = m.mut_nuc = 0;                // :583-589
                node->add_mutation(m);
            }
        }
        if (!std::is_sorted(node->mutations.begin(), node->mutations.end())) {
            fprintf(stderr, "WARNING: Mutations not sorted!\n");
            std::sort(node->mutations.begin(), node->mutations.end());
        }
    }
    lap("mutation lists");
    for (auto& cn : condensed) {
        for (auto& l : cn.second) tree.condensed_leaves.insert(l);
        tree.condensed_nodes[cn.first] = std::move(cn.second);
    }
    return tree;
}

void save_mutation_annotated_tree(const Tree& tree, std::string const& filename) {
    std::string out;
    put_len(out, 1, get_newick_string(tree));
    for (Node* n : tree.depth_first_expansion()) {
        std::string lst;
        for (auto& m : n->mutations) {
            std::string mm;
            put_varint(mm, (1 << 3) | 0); put_varint(mm, (uint64_t)(int64_t)m.position);
            if (!m.is_masked()) {
                int ref = __builtin_ctz((unsigned)m.ref_nuc | 16), par = __builtin_ctz((unsigned)m.par_nuc | 16);
                put_varint(mm, (2 << 3) | 0); put_varint(mm, (uint64_t)ref);
                put_varint(mm, (3 << 3) | 0); put_varint(mm, (uint64_t)par);
                std::string packed;
                for (int b = 0; b < 4; b++) if (m.mut_nuc & (1 << b)) put_varint(packed, (uint64_t)b);
                put_len(mm, 4, packed);
            }
            if (!m.chrom.empty()) put_len(mm, 5, m.chrom);
            put_len(lst, 1, mm);
        }
        put_len(out, 2, lst);
    }
    for (auto& cn : tree.condensed_nodes) {
        std::string c;
        put_len(c, 1, cn.first);
        for (auto& l : cn.second) put_len(c, 2, l);
        put_len(out, 3, c);
    }
    if (filename.size() > 3 && filename.compare(filename.size() - 3, 3, ".gz") == 0) {
        // gzip like UCSC's public MATs (the loader inflates, mutation_annotated_tree.cpp:528-543); in slices: gzwrite takes 32-bit lengths
        gzFile f = gzopen(filename.c_str(), "wb1");
        if (!f) throw mat_error("ERROR: Could not write the mutation-annotated tree file: " + filename + "!");
        for (size_t at = 0; at < out.size();) {
            const unsigned n = (unsigned)std::min<size_t>(out.size() - at, 1u << 30);
            if (gzwrite(f, out.data() + at, n) != (int)n) { gzclose(f); throw mat_error("ERROR: Could not write the mutation-annotated tree file: " + filename + "!"); }
            at += n;
        }
        if (gzclose(f) != Z_OK) throw mat_error("ERROR: Could not write the mutation-annotated tree file: " + filename + "!");
        return;
    }
    std::ofstream f(filename, std::ios::binary);
    if (!f) throw mat_error("ERROR: Could not write the mutation-annotated tree file: " + filename + "!");
    f.write(out.data(), (std::streamsize)out.size());
}

// ---- VCF -----------------------------------------------------------------------------
void read_vcf(Tree* T, std::string const& vcf_filename, std::vector<Missing_Sample>& missing_samples) {
    fprintf(stderr, "Loading VCF file\n");
    std::string raw = slurp(vcf_filename, "VCF");
    // a number of a VCF field, or mat_error (std::stoi would throw std::invalid_argument / std::out_of_range)
    auto field_int = [](const std::string& w, const char* what) -> int {
        char* endp = nullptr;
        errno = 0;
        const long v = std::strtol(w.c_str(), &endp, 10);
        if (w.empty() || endp == w.c_str() || errno == ERANGE || v < 0 || v > 0x7FFFFFFFL)
            throw mat_error(std::string("ERROR! Incorrect VCF format: ") + what + " '" + w + "'.");
        return (int)v;
    };
    std::istringstream in(raw);
    std::string line;
    bool header_found = false;
    size_t n_columns = 0;
    std::vector<size_t> sample_column;   // VCF column of missing_samples[first + k]
    const size_t first = missing_samples.size();
    while (std::getline(in, line)) {
        std::vector<std::string> words;
        string_split(line, words);
        if (!header_found) {
            if (words.size() > 1 && words[1] == "POS") {
                for (size_t j = 9; j < words.size(); j++) {
                    if (T->get_node(words[j]) == nullptr && !T->condensed_leaves.count(words[j])) {
                        missing_samples.emplace_back(words[j]);
                        sample_column.push_back(j);
                    } else {
                        fprintf(stderr, "WARNING: Ignoring sample %s as it is already in the tree.\n", words[j].c_str());
                    }
                }
                if (words.size() < 9)
                    throw mat_error("ERROR! Incorrect VCF format. The header names " + std::to_string(words.size()) + " columns, at least 9 expected.");
                n_columns = words.size();
                header_found = true;
            }
            continue;
        }
        if (words.size() != n_columns)
            throw mat_error("ERROR! Incorrect VCF format. Expected " + std::to_string(n_columns) + " columns but got " +
                            std::to_string(words.size()) + ".");
        std::vector<std::string> alleles;
        string_split(words[4], ',', alleles);
        for (size_t k = 0; k < sample_column.size(); k++) {
            const std::string& gt = words[sample_column[k]];
            Mutation m;
            m.chrom = words[0];
            m.position = field_int(words[1], "position");
            m.ref_nuc = get_nuc_id(words[3][0]);
            m.par_nuc = m.ref_nuc;
            bool emit = true;
            if (!gt.empty() && isdigit((unsigned char)gt[0])) {
                // (the leading run of digits: "1", "1/1", "1|0" all name allele 1, as std::stoi read them)
                size_t nd = 0;
                while (nd < gt.size() && isdigit((unsigned char)gt[nd])) nd++;
                int allele_id = field_int(gt.substr(0, nd), "genotype");
                if (allele_id > 0) {
                    if ((size_t)allele_id > alleles.size()) throw mat_error("ERROR! VCF genotype refers to a missing ALT allele.");
                    m.mut_nuc = get_nuc_id(alleles[(size_t)allele_id - 1][0]);
                    m.is_missing = (m.mut_nuc == 15);
                } else emit = false;      // reference call: no entry
            } else {
                m.is_missing = true;      // '.' and friends
                m.mut_nuc = 15;
            }
            if (!emit) continue;
            Missing_Sample& ms = missing_samples[first + k];
            ms.mutations.push_back(m);
            if (m.mut_nuc & (m.mut_nuc - 1)) ms.num_ambiguous++;
        }
    }
}

}  // namespace Mutation_Annotated_Tree
