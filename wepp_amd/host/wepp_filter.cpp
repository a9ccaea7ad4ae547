// wepp_filter.cpp -- see wepp_filter.hpp (citations: /root/reference/src/WEPP/).
#include "wepp_filter.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <sstream>

#include "../../include/wepp_place.h"
#include "pbwire.hpp"

using MAT::mat_error;
namespace pbwire = MAT::pbwire;

std::string load_reference(std::string const& fasta_filename) {
    std::ifstream fasta_f(fasta_filename);
    if (!fasta_f.is_open()) throw mat_error("Error: Unable to open file " + fasta_filename);
    std::string header, temp, ref_seq;
    std::getline(fasta_f, header);
    while (std::getline(fasta_f, temp)) {
        if (!temp.empty() && temp.back() == '\r') temp.pop_back();
        std::transform(temp.begin(), temp.end(), temp.begin(), [](unsigned char c) { return (char)std::toupper(c); });
        ref_seq += temp;
    }
    return ref_seq;
}

std::vector<int> load_masked_sites(std::string const& bed_filename) {
    std::vector<int> mask;
    std::ifstream in(bed_filename);
    if (!in.is_open()) return mask;                       // assume no masks
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string c1, c2;
        int c3;
        if (ls >> c1 >> c2 >> c3) mask.push_back(c3);
    }
    return mask;
}

std::vector<raw_read> load_reads_from_proto(std::string const& reference, std::string const& filename,
                                            std::unordered_map<std::string, std::vector<std::string>>& reverse_merge) {
    std::string raw = pbwire::slurp(filename, "read protobuf");
    pbwire::Wire top{(const uint8_t*)raw.data(), (const uint8_t*)raw.data() + raw.size()};
    std::vector<raw_read> reads;
    while (!top.eof()) {
        uint64_t key = top.varint();
        uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
        if (field == 1 && wt == 2) {                      // read_info
            pbwire::Wire w = top.sub();
            std::string name, content;
            int start_idx = 0, degree = 0;
            while (!w.eof()) {
                uint64_t k = w.varint();
                uint32_t f = (uint32_t)(k >> 3), t = (uint32_t)(k & 7);
                if (f == 1 && t == 2) name = w.str();
                else if (f == 3 && t == 0) start_idx = (int)(int32_t)w.varint();
                else if (f == 6 && t == 2) content = w.str();
                else if (f == 5 && t == 0) degree = (int)(int32_t)w.varint();
                else w.skip(t);
            }
            raw_read out;
            out.start = start_idx;
            out.end = start_idx + (int)content.size() - 1;
            out.degree = degree;
            out.read = name;
            if (start_idx < 1 || (size_t)out.end > reference.size())
                throw mat_error("ERROR: read " + name + " does not lie inside the reference");
            for (size_t i = 0; i < content.size(); ++i) {
                const char ref_c = reference[(size_t)start_idx + i - 1];
                if (content[i] != ref_c && content[i] != '_') {
                    MAT::Mutation m;
                    m.is_missing = content[i] == 'N';
                    m.ref_nuc = m.par_nuc = MAT::get_nuc_id(ref_c);
                    m.mut_nuc = MAT::get_nuc_id(content[i]);
                    m.position = out.start + (int)i;
                    out.mutations.push_back(std::move(m));
                }
            }
            reads.push_back(std::move(out));
        } else if (field == 2 && wt == 2) {               // column_info
            pbwire::Wire w = top.sub();
            std::string col;
            std::vector<std::string> inputs;
            while (!w.eof()) {
                uint64_t k = w.varint();
                uint32_t f = (uint32_t)(k >> 3), t = (uint32_t)(k & 7);
                if (f == 1 && t == 2) col = w.str();
                else if (f == 2 && t == 2) inputs.push_back(w.str());
                else w.skip(t);
            }
            for (auto& s : inputs) reverse_merge[col].push_back(s);
        } else {
            top.skip(wt);
        }
    }
    return reads;
}

void dump_reads_proto(std::vector<sam_read_record> const& reads, std::string const& filename) {
    std::string out;
    for (auto const& r : reads) {
        std::string m;
        pbwire::put_len(m, 1, r.name);
        pbwire::put_varint(m, (3u << 3) | 0);
        pbwire::put_varint(m, (uint64_t)(int64_t)r.start_idx);
        pbwire::put_varint(m, (5u << 3) | 0);
        pbwire::put_varint(m, (uint64_t)(int64_t)r.degree);
        pbwire::put_len(m, 6, r.content);
        pbwire::put_len(out, 1, m);
    }
    std::ofstream f(filename, std::ios::out | std::ios::binary);
    if (!f) throw mat_error("ERROR: Could not write the read protobuf: " + filename);
    f.write(out.data(), (std::streamsize)out.size());
}

void mask_reads(std::vector<raw_read>& reads, std::vector<int> const& masked_sites) {
    if (masked_sites.empty()) return;
    std::unordered_set<int> mask(masked_sites.begin(), masked_sites.end());
    for (auto& rd : reads)
        rd.mutations.erase(std::remove_if(rd.mutations.begin(), rd.mutations.end(),
                                          [&](const MAT::Mutation& m) { return mask.count(m.position) != 0; }),
                           rd.mutations.end());
}

std::unordered_set<int> site_read_map(std::vector<raw_read> const& reads, std::vector<int> const& masked_sites) {
    // A site is covered when some read spans it with a base other than N, and it is not masked:
    // (reads spanning j) - (reads with an N at j) > 0.  Both counts come from one pass over the reads
    // (a difference array over the genome for the spans), not from a walk over every base of every read.
    int last = 0;
    for (auto const& rd : reads) last = std::max(last, rd.end);
    std::vector<int32_t> span((size_t)last + 2, 0), n_at((size_t)last + 2, 0);
    for (auto const& rd : reads) {
        if (rd.end < rd.start) continue;
        const int lo = std::max(rd.start, 0);
        span[(size_t)lo] += 1;
        span[(size_t)rd.end + 1] -= 1;
        for (auto const& mut : rd.mutations)
            if (mut.mut_nuc == 0b1111 && mut.position >= lo && mut.position <= rd.end) n_at[(size_t)mut.position] += 1;
    }
    std::vector<char> masked((size_t)last + 2, 0);
    for (int site : masked_sites)
        if (site >= 0 && site <= last) masked[(size_t)site] = 1;
    std::unordered_set<int> covered;
    int32_t open_reads = 0;
    for (int j = 0; j <= last; j++) {
        open_reads += span[(size_t)j];
        if (open_reads - n_at[(size_t)j] > 0 && !masked[(size_t)j]) covered.insert(j);
    }
    return covered;
}

MAT::Tree create_condensed_tree(MAT::Node* ref_root, const std::unordered_set<int>& site_read_map,
                                std::unordered_map<MAT::Node*, std::vector<MAT::Node*>>& node_mappings) {
    // Breadth-first order of the original tree; rep[k] = the condensed node that stands for original
    // node k: a new node when k keeps a mutation at a covered site (the root always), else the
    // representative of its parent.  Parents precede children in this order, and the children of a
    // condensed node are created in the order their originals appear in it -- which is what fixes the
    // pre-order (arena) index of every haplotype.
    MAT::Tree T;
    std::vector<MAT::Node*> order{ref_root};
    std::vector<size_t> parent_slot{0};
    for (size_t head = 0; head < order.size(); head++)
        for (MAT::Node* c : order[head]->children) { order.push_back(c); parent_slot.push_back(head); }
    std::vector<MAT::Node*> rep(order.size(), nullptr);
    for (size_t k = 0; k < order.size(); k++) {
        MAT::Node* orig = order[k];
        std::vector<MAT::Mutation> kept;
        std::copy_if(orig->mutations.begin(), orig->mutations.end(), std::back_inserter(kept),
                     [&](const MAT::Mutation& m) { return site_read_map.count(m.position) != 0; });
        if (k == 0 || !kept.empty()) {
            rep[k] = k == 0 ? T.create_node(orig->identifier, -1.0f)
                            : T.create_node(orig->identifier, rep[parent_slot[k]], -1.0f);
            rep[k]->mutations = std::move(kept);
            node_mappings[rep[k]] = {orig};
        } else {
            rep[k] = rep[parent_slot[k]];
            node_mappings[rep[k]].push_back(orig);
        }
    }
    return T;
}

int cartesian_map(MAT::Tree& condensed, const std::vector<raw_read>& reads, size_t genome_size,
                  cartesian_map_result& out, int device) {
    if (!condensed.root) {
        fprintf(stderr, "ERROR: empty tree!\n");
        return 1;
    }
    // node ids = BFS order (children of a node ascending = stored order), as in usher_place.cpp
    std::vector<MAT::Node*> bfs = condensed.breadth_first_expansion();
    const size_t N = bfs.size();
    std::unordered_map<const MAT::Node*, int32_t> id;
    id.reserve(N * 2);
    for (size_t k = 0; k < N; k++) id[bfs[k]] = (int32_t)k;
    std::vector<int32_t> parent(N), mut_pos;
    std::vector<uint32_t> mut_off(N + 1, 0);
    std::vector<uint8_t> mut_ref, mut_par, mut_mut;
    for (size_t k = 0; k < N; k++) {
        parent[k] = bfs[k]->parent ? id[bfs[k]->parent] : -1;
        std::vector<MAT::Mutation> muts = bfs[k]->mutations;
        std::sort(muts.begin(), muts.end());               // arena.cpp:48
        for (auto& m : muts) {
            mut_pos.push_back(m.position);
            mut_ref.push_back((uint8_t)m.ref_nuc);
            mut_par.push_back((uint8_t)m.par_nuc);
            mut_mut.push_back((uint8_t)m.mut_nuc);
        }
        mut_off[k + 1] = (uint32_t)mut_pos.size();
    }
    wepp_tree_desc desc{(uint32_t)N, parent.data(), mut_off.data(), mut_pos.data(), mut_ref.data(), mut_par.data(),
                        mut_mut.data()};
    wepp_mat_t* mat = nullptr;
    if (wepp_mat_create(&desc, device, &mat) != WEPP_OK) {
        fprintf(stderr, "ERROR: %s\n", wepp_last_error());
        return 1;
    }
    const size_t R = reads.size();
    std::vector<uint32_t> off(1, 0), words;
    std::vector<int32_t> start(R), end(R), degree(R);
    for (size_t r = 0; r < R; r++) {
        for (const MAT::Mutation& m : reads[r].mutations)
            words.push_back(wepp_pack_read_word((uint32_t)m.position, (uint32_t)m.ref_nuc, (uint32_t)m.mut_nuc,
                                                m.mut_nuc == 0b1111 ? 1u : 0u));
        off.push_back((uint32_t)words.size());
        start[r] = reads[r].start; end[r] = reads[r].end; degree[r] = reads[r].degree;
    }
    wepp_epp_reads in{(uint32_t)R, off.data(), words.data(), start.data(), end.data(), degree.data()};
    std::vector<int32_t> pars(R), counts(N * NUM_RANGE_BINS);
    std::vector<uint32_t> mult(R), epp, order(N);
    std::vector<uint64_t> epp_off(R + 1);
    out.score.assign(N, 0.0);
    out.dist_divergence.assign(N, 0.0);
    // The EPP lists hold what the reads' multiplicities add up to, at most MAX_CACHED_EPP_SIZE each: known only once
    // the map has run (the worst case, R * 2048 entries, is 8 GiB per million reads).  A typical guess first; when it is
    // short the map still delivers everything else and keeps the lists (WEPP_ELIMIT): they are fetched into a buffer
    // of the reported size -- the map itself runs ONCE.
    epp.resize(std::max<size_t>(R, 1) * 16);
    wepp_epp_out o{pars.data(), mult.data(), epp_off.data(), epp.data(), epp.size(), out.score.data(), counts.data(),
                   out.dist_divergence.data()};
    int rc = wepp_epp_map(mat, &in, (uint32_t)genome_size, MAX_CACHED_EPP_SIZE, &o);
    if (rc == WEPP_ELIMIT && epp_off[R] > epp.size()) {
        epp.assign((size_t)epp_off[R], 0);
        rc = wepp_epp_fetch_lists(mat, epp.data(), epp.size());
    }
    if (rc == WEPP_OK) rc = wepp_mat_dfs_order(mat, order.data());
    if (rc != WEPP_OK) {
        fprintf(stderr, "ERROR: %s\n", wepp_last_error());
        wepp_mat_destroy(mat);
        return 1;
    }
    wepp_mat_destroy(mat);
    out.haplotypes.resize(N);
    out.mapped_read_counts.resize(N);
    for (size_t k = 0; k < N; k++) {
        out.haplotypes[k] = bfs[order[k]];
        std::copy_n(&counts[k * NUM_RANGE_BINS], NUM_RANGE_BINS, out.mapped_read_counts[k].begin());
    }
    out.max_parismony.assign(pars.begin(), pars.end());
    out.parsimony_multiplicity.assign(mult.begin(), mult.end());
    out.epp_positions_cache.assign(R, {});
    for (size_t r = 0; r < R; r++)
        out.epp_positions_cache[r].assign(epp.begin() + (long)epp_off[r], epp.begin() + (long)epp_off[r + 1]);
    return 0;
}
