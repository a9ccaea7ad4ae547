// flat_io.cpp -- the flat image (wepp_flat_t) as a file: ONE flatten per node serves every rank of it.
//
// The C++ host flattens once per process and uploads from every device thread (wepp_flat_create + wepp_mat_upload);
// ranks that are processes of their own (bench.py --gpus N under torch.distributed.run, one rank per GPU) cannot share
// a pointer: rank 0 flattens and writes the image to /dev/shm, the other ranks read it back -- seconds instead of
// another 16 s x 16 threads and 15 GB of resident memory per rank for a 16 M-node tree.  The reference re-expands the
// tree per SAMPLE (src/usher_common.cpp:339).  The file is a plain dump of the image's arrays behind a header that
// pins the build's layout constants; it is a cache, not an exchange format.
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/wepp_place.h"
#include "errors.hpp"
#include "flatmat.hpp"

namespace {

using namespace wepp;

constexpr char MAGIC[8] = {'W', 'E', 'P', 'P', 'F', 'L', 'T', '2'};
struct Header {
    char magic[8];
    uint32_t sizes[8];      // sizeof of the record types + the layout constants the image depends on
};
Header this_build() {
    Header h{};
    std::memcpy(h.magic, MAGIC, 8);
    const uint32_t s[8] = {(uint32_t)sizeof(IxEnt), (uint32_t)sizeof(BlkSum), (uint32_t)sizeof(SegNode), (uint32_t)sizeof(NodeRec),
                           WIN_SIZE, WIN_STRIDE, MAX_STREAMS * 65536u + WC_MAX, RQ_BLK * 65536u + BLK_MAX_NODES};
    std::memcpy(h.sizes, s, sizeof(s));
    return h;
}

struct Writer {
    FILE* f;
    bool ok = true;
    template <typename T> void pod(T& v) { ok = ok && std::fwrite(&v, sizeof(T), 1, f) == 1; }
    template <typename T> void vec(std::vector<T>& v) {
        uint64_t n = v.size();
        pod(n);
        if (n) ok = ok && std::fwrite(v.data(), sizeof(T), n, f) == n;
    }
    template <typename T, typename F> void list(std::vector<T>& v, F&& each) {
        uint64_t n = v.size();
        pod(n);
        for (T& x : v) each(*this, x);
    }
};
struct Reader {
    FILE* f;
    bool ok = true;
    template <typename T> void pod(T& v) { ok = ok && std::fread(&v, sizeof(T), 1, f) == 1; }
    template <typename T> void vec(std::vector<T>& v) {
        uint64_t n = 0;
        pod(n);
        if (!ok || n > (1ull << 40) / sizeof(T)) { ok = false; return; }
        v.resize(n);
        if (n) ok = ok && std::fread(v.data(), sizeof(T), n, f) == n;
    }
    template <typename T, typename F> void list(std::vector<T>& v, F&& each) {
        uint64_t n = 0;
        pod(n);
        if (!ok || n > (1u << 20)) { ok = false; return; }
        v.resize(n);
        for (T& x : v) each(*this, x);
    }
};

template <typename Ar> void io_stream(Ar& a, Stream& s) {
    a.pod(s.tau); a.pod(s.n); a.pod(s.NB); a.pod(s.cp_stride); a.pod(s.E);
    a.vec(s.nkey); a.vec(s.nstat); a.vec(s.ncnt); a.vec(s.blk_node0); a.vec(s.blk_eoff); a.vec(s.blk_sum);
    a.vec(s.ev_word); a.vec(s.ev_meta); a.vec(s.ev_lb); a.vec(s.cp_off); a.vec(s.cp_word);
    a.vec(s.ix_head); a.vec(s.ix_ent); a.vec(s.ix_nest); a.vec(s.ix_pre); a.vec(s.nrec);
    a.pod(s.sp_levels); a.pod(s.rq_blocks); a.pod(s.rq_levels);
    a.vec(s.sp); a.vec(s.rq_pre); a.vec(s.rq_suf); a.vec(s.rq_dst); a.pod(s.whole);
}
template <typename Ar> void io_flat(Ar& a, FlatMAT& f) {
    a.pod(f.N); a.pod(f.n_leaves); a.pod(f.max_depth); a.pod(f.max_pos); a.pod(f.M); a.pod(f.n_masked); a.pod(f.root_base);
    a.vec(f.node_woff); a.vec(f.words); a.vec(f.nkey); a.vec(f.nstat); a.vec(f.rank2dfs); a.vec(f.dfs2bfs); a.vec(f.rank2bfs);
    a.vec(f.bfs2id); a.vec(f.dfs2id); a.vec(f.parent_dfs); a.vec(f.dfs_end); a.vec(f.num_leaves); a.vec(f.maxnest);
    a.list(f.streams, [](Ar& b, Stream& s) { io_stream(b, s); });
    a.list(f.wstreams, [](Ar& b, Stream& s) { io_stream(b, s); });
    a.list(f.wcrowns, [](Ar& b, std::vector<Stream>& v) { b.list(v, [](Ar& c, Stream& s) { io_stream(c, s); }); });
    a.vec(f.epp_word); a.vec(f.epp_node);
    a.pod(f.seed_stride); a.pod(f.seed_chunks); a.pod(f.seed_row_words); a.vec(f.seed_sig);
}

}  // namespace

extern "C" int wepp_flat_save(const wepp_flat_t* flat, const char* path) {
    if (!flat || !path) return wepp::set_error(WEPP_EINVAL, "null argument");
    const std::string tmp = std::string(path) + ".part";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return wepp::set_error(WEPP_EINVAL, std::string("cannot create ") + tmp);
    std::setvbuf(f, nullptr, _IOFBF, 8u << 20);
    Header h = this_build();
    Writer w{f};
    w.pod(h);
    io_flat(w, const_cast<wepp::FlatMAT&>(flat->f));     // (the writer only reads)
    const bool ok = w.ok && std::fclose(f) == 0;
    if (!ok || std::rename(tmp.c_str(), path) != 0) {     // (readers never see a partial file)
        std::remove(tmp.c_str());
        return wepp::set_error(WEPP_ENOMEM, std::string("writing the flat image to ") + path + " failed (disk or /dev/shm full?)");
    }
    return WEPP_OK;
}

extern "C" int wepp_flat_load(const char* path, wepp_flat_t** out) {
    if (!path || !out) return wepp::set_error(WEPP_EINVAL, "null argument");
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) return wepp::set_error(WEPP_EINVAL, std::string("cannot open ") + path);
    std::setvbuf(f, nullptr, _IOFBF, 8u << 20);
    wepp_flat* h = new (std::nothrow) wepp_flat();
    if (!h) { std::fclose(f); return wepp::set_error(WEPP_ENOMEM, "out of host memory"); }
    int rc = WEPP_OK;
    try {
        Header got{}, want = this_build();
        Reader r{f};
        r.pod(got);
        if (!r.ok || std::memcmp(&got, &want, sizeof(Header)) != 0)
            rc = wepp::set_error(WEPP_EINVAL, std::string(path) + " is not a flat image of this build of the library");
        else {
            io_flat(r, h->f);
            if (!r.ok || h->f.streams.empty() || h->f.node_woff.size() != (size_t)h->f.N + 1)
                rc = wepp::set_error(WEPP_EINVAL, std::string(path) + ": truncated or damaged flat image");
        }
    } catch (const std::bad_alloc&) {
        rc = wepp::set_error(WEPP_ENOMEM, "out of host memory while reading the flat image");
    }
    std::fclose(f);
    if (rc != WEPP_OK) { delete h; return rc; }
    *out = h;
    return WEPP_OK;
}
