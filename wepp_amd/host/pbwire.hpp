// pbwire.hpp -- the few lines of protobuf wire format the host layer needs (parsimony.proto,
// sam.proto): varints, length-delimited fields, and a gz-transparent file reader.  No libprotobuf.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <string>

#include "mat.hpp"

namespace Mutation_Annotated_Tree {
namespace pbwire {

inline std::string slurp(std::string const& filename, const char* what) {
    // gzopen reads plain files transparently, so one path serves .pb and .pb.gz / .vcf and .vcf.gz
    gzFile f = gzopen(filename.c_str(), "rb");
    if (!f) throw mat_error(std::string("ERROR: Could not open the ") + what + " file: " + filename + "!");
    std::string data;
    char buf[1 << 16];
    int got;
    while ((got = gzread(f, buf, sizeof buf)) > 0) data.append(buf, (size_t)got);
    gzclose(f);
    if (got < 0) throw mat_error(std::string("ERROR: Could not read the ") + what + " file: " + filename + "!");
    return data;
}

struct Wire {
    const uint8_t* p;
    const uint8_t* end;
    bool eof() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) throw mat_error("truncated varint in .pb");
            uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
        }
        throw mat_error("malformed varint in .pb");
    }
    Wire sub() {
        uint64_t len = varint();
        if (len > (uint64_t)(end - p)) throw mat_error("truncated field in .pb");
        Wire w{p, p + len};
        p += len;
        return w;
    }
    std::string str() {
        Wire s = sub();
        return std::string((const char*)s.p, (size_t)(s.end - s.p));
    }
    void skip(uint32_t wt) {
        switch (wt) {
        case 0: varint(); break;
        case 1: if (end - p < 8) throw mat_error("truncated .pb"); p += 8; break;
        case 2: sub(); break;
        case 5: if (end - p < 4) throw mat_error("truncated .pb"); p += 4; break;
        default: throw mat_error("unsupported wire type in .pb");
        }
    }
};

inline void put_varint(std::string& o, uint64_t v) {
    while (v >= 0x80) { o.push_back((char)(v | 0x80)); v >>= 7; }
    o.push_back((char)v);
}
inline void put_len(std::string& o, uint32_t field, const std::string& payload) {
    put_varint(o, (field << 3) | 2);
    put_varint(o, payload.size());
    o += payload;
}

}  // namespace pbwire
}  // namespace Mutation_Annotated_Tree
