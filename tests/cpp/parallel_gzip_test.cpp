// One gzip stream on several threads (deacon-server_amd/cli/parallel_gzip.hpp) against the bytes that went in: built with
// -fsanitize=address,undefined and with -fsanitize=thread by tests/test_fast_inflate.py.  Streams are made here with zlib
// (levels 1..9, one or several members, four kinds of content), read back with chunks so small that a stream is dozens of
// them -- every stretch a worker decodes from a block it found, every stretch the driver decodes itself, blocks longer than a
// chunk's overlap -- whole, cut short, and with bits flipped.  Exit code 0 = every intact stream came back byte for byte, every
// cut one was refused as truncated, and no damaged one came back as other bytes without an error.
#include "parallel_gzip.hpp"

#include <cstdio>
#include <random>

namespace {
struct Mem {
    const std::vector<unsigned char> *v;
    size_t at = 0;
    size_t piece;  // bytes per call: a pipe delivers less than asked for
};
size_t mem_source(void *ctx, unsigned char *dst, size_t cap) {
    Mem *m = (Mem *)ctx;
    const size_t n = std::min(std::min(cap, m->piece), m->v->size() - m->at);
    std::memcpy(dst, m->v->data() + m->at, n);
    m->at += n;
    return n;
}
std::vector<unsigned char> gz_member(const unsigned char *p, size_t n, int level) {
    z_stream z{};
    deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY);
    std::vector<unsigned char> out(deflateBound(&z, n) + 64);
    z.next_in = (Bytef *)p;
    z.avail_in = (uInt)n;
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    deflate(&z, Z_FINISH);
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}
// 0 = read back equal, 1 = error reported, 2 = other bytes without an error
int read_back(const std::vector<unsigned char> &gz, const std::vector<unsigned char> &want, unsigned workers, size_t chunk, size_t piece, std::string *err) {
    Mem m{&gz, 0, piece};
    fastgz::ParallelGzReader r(mem_source, &m, workers, chunk);
    std::vector<unsigned char> got;
    std::vector<char> buf(1 << 20);
    for (size_t k; (k = r.read(buf.data(), 1 + (got.size() * 7919) % buf.size())) > 0;) got.insert(got.end(), buf.begin(), buf.begin() + (long)k);
    if (!r.error().empty()) {
        if (err) *err = r.error();
        return 1;
    }
    return got == want ? 0 : 2;
}
}  // namespace

int main(int argc, char **argv) {
    std::mt19937_64 rng(11);
    int bad = 0;
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 36;
    for (int t = 0; t < rounds; ++t) {
        const size_t n = 200000 + rng() % 3000000;
        std::vector<unsigned char> data(n);
        const int mode = t % 4;
        for (size_t i = 0; i < n; ++i)
            data[i] = mode == 0 ? "ACGT"[rng() & 3] : mode == 1 ? (unsigned char)(33 + rng() % 41) : mode == 2 ? (unsigned char)("ACGT\nI@+"[rng() % 8]) : (unsigned char)(i % 251);
        // one member, or three of different levels (the middle one may be empty)
        std::vector<unsigned char> gz;
        if (t % 3 == 0) gz = gz_member(data.data(), n, 1 + t % 9);
        else {
            const size_t a = rng() % n, b = t % 2 ? a : a + rng() % (n - a);
            for (auto part : {std::make_pair((size_t)0, a), std::make_pair(a, b), std::make_pair(b, n)}) {
                auto m = gz_member(data.data() + part.first, part.second - part.first, 1 + (int)(rng() % 9));
                gz.insert(gz.end(), m.begin(), m.end());
            }
        }
        const size_t chunk = t % 2 ? 20000 : 150000;
        const unsigned workers = 2 + t % 3;
        const size_t piece = t % 5 == 0 ? 4093 : (size_t)1 << 30;
        std::string err;
        int rc = read_back(gz, data, workers, chunk, piece, &err);
        if (rc != 0) {
            ++bad;
            std::printf("FAIL intact %d: rc %d %s\n", t, rc, err.c_str());
        }
        // cut short: refused as truncated
        std::vector<unsigned char> cut(gz.begin(), gz.begin() + (long)(gz.size() - 1 - rng() % (gz.size() / 2)));
        err.clear();
        rc = read_back(cut, data, workers, chunk, piece, &err);
        if (rc != 1 || err != "truncated gzip stream") {
            ++bad;
            std::printf("FAIL cut %d: rc %d '%s'\n", t, rc, err.c_str());
        }
        // bits flipped: an error, or (a flip the format does not notice) the same bytes -- never other bytes in silence
        std::vector<unsigned char> hurt = gz;
        for (int q = 0; q < 3; ++q) hurt[10 + rng() % (hurt.size() - 10)] ^= (unsigned char)(1u << (rng() & 7));
        rc = read_back(hurt, data, workers, chunk, piece, &err);
        if (rc == 2) {
            ++bad;
            std::printf("FAIL hurt %d: other bytes accepted\n", t);
        }
    }
    std::printf("bad %d\n", bad);
    return bad != 0;
}
