// Mutation fuzz of two more loaders of the C++ host mirror, under AddressSanitizer + UBSan (tests/test_pb_fuzz.py):
//   vcf   <tree.pb> <samples.vcf> <tmp> <rounds>   read_vcf(T, vcf, missing_samples)          (host/mat.cpp)
//   reads <ref.fa>  <reads.pb>    <tmp> <rounds>   load_reads_from_proto(reference, file, ..)  (host/wepp_filter.cpp)
// Damaged copies of a valid file must load or be rejected with MAT::mat_error; nothing else.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "mat.hpp"
#include "wepp_filter.hpp"

static std::vector<char> slurp_file(const std::string& p) {
    std::ifstream in(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    const std::string mode = argv[1], aux = argv[2], src = argv[3], tmp = argv[4];
    const int rounds = std::atoi(argv[5]);
    const bool text = mode == "vcf";
    std::vector<char> good = slurp_file(src);
    if (good.empty()) return 2;
    MAT::Tree T;
    std::string reference;
    if (mode == "vcf") T = MAT::load_mutation_annotated_tree(aux);
    else reference = load_reference(aux);
    auto load = [&](const std::string& file) {
        if (mode == "vcf") {
            std::vector<Missing_Sample> ms;
            MAT::read_vcf(&T, file, ms);
        } else {
            std::unordered_map<std::string, std::vector<std::string>> rev;
            (void)load_reads_from_proto(reference, file, rev);
        }
    };
    load(src);                                      // the undamaged file loads
    std::mt19937 rng(4321);
    const char alphabet[] = "ACGTN.,\t\n0123456789|/:#-";
    int loaded = 0, rejected = 0;
    for (int r = 0; r < rounds; r++) {
        std::vector<char> b = good;
        const int kind = (int)(rng() % 4), hits = 1 + (int)(rng() % 4);
        for (int h = 0; h < hits; h++) {
            const size_t at = rng() % b.size();
            const char c = text ? alphabet[rng() % (sizeof(alphabet) - 1)] : (char)(rng() & 0xFF);
            if (kind == 0) b[at] = c;
            else if (kind == 1) b[at] ^= (char)(1u << (rng() % 8));
            else if (kind == 2) { b.resize(at + 1); break; }
            else b.insert(b.begin() + (std::ptrdiff_t)at, c);
        }
        { std::ofstream out(tmp, std::ios::binary | std::ios::trunc); out.write(b.data(), (std::streamsize)b.size()); }
        try { load(tmp); loaded++; } catch (const MAT::mat_error&) { rejected++; }
    }
    std::printf("ok %d loaded %d rejected\n", loaded, rejected);
    return 0;
}
