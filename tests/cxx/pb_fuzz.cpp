// Mutation fuzz of the hand-written protobuf wire parser behind load_mutation_annotated_tree
// (wepp_amd/host/mat.cpp, pbwire.hpp): a valid .pb is read, then damaged copies of it (byte flips, truncations,
// inserted bytes) are loaded -- every copy must either load or be rejected with MAT::mat_error; nothing else
// (no crash, no sanitizer report, no other exception, no runaway allocation).  Built with AddressSanitizer + UBSan by
// tests/test_pb_fuzz.py.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "mat.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string src = argv[1], tmp = argv[2];
    const int rounds = std::atoi(argv[3]);
    std::ifstream in(src, std::ios::binary);
    std::vector<char> good((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (good.empty()) return 2;
    { MAT::Tree t = MAT::load_mutation_annotated_tree(src); if (t.size() == 0) return 3; }
    std::mt19937 rng(12345);
    int loaded = 0, rejected = 0;
    for (int r = 0; r < rounds; r++) {
        std::vector<char> b = good;
        const int kind = (int)(rng() % 4);
        const int hits = 1 + (int)(rng() % 4);
        for (int h = 0; h < hits; h++) {
            const size_t at = rng() % b.size();
            if (kind == 0) b[at] = (char)(rng() & 0xFF);
            else if (kind == 1) b[at] ^= (char)(1u << (rng() % 8));
            else if (kind == 2) { b.resize(at + 1); break; }
            else b.insert(b.begin() + (std::ptrdiff_t)at, (char)(rng() & 0xFF));
        }
        { std::ofstream out(tmp, std::ios::binary | std::ios::trunc); out.write(b.data(), (std::streamsize)b.size()); }
        try {
            MAT::Tree t = MAT::load_mutation_annotated_tree(tmp);
            loaded++;
        } catch (const MAT::mat_error&) {
            rejected++;
        }
    }
    std::printf("ok %d loaded %d rejected\n", loaded, rejected);
    return 0;
}
