// The tool's gzip decoder on memory it does not own: built with -fsanitize=address,undefined by
// tests/test_fast_inflate.py::test_decoder_stays_inside_its_buffers_under_the_sanitizers.  Raw deflate payloads of BGZF-member
// size (every level, fixed and dynamic codes, four kinds of content), whole and with bits flipped, into buffers of exactly
// the size the interface promises (fastgz::PAD bytes behind the input, 258 + 16 behind the output): any access outside is
// the sanitizer's to report.  Exit code 0 = every intact payload came back byte for byte.
#include "fast_inflate.hpp"
#include <cstdio>
#include <memory>
#include <random>
int main() {
    std::mt19937_64 rng(7);
    std::unique_ptr<fastgz::BlockDecoder> dec(new fastgz::BlockDecoder());
    int bad = 0;
    for (int t = 0; t < 3000; ++t) {
        size_t n = rng() % 65536;
        std::vector<unsigned char> data(n);
        int mode = t % 4;
        for (size_t i = 0; i < n; ++i) data[i] = mode == 0 ? "ACGT"[rng() & 3] : mode == 1 ? (unsigned char)rng() : mode == 2 ? (unsigned char)(33 + rng() % 41) : (unsigned char)(i % 7);
        z_stream z{};
        deflateInit2(&z, t % 10, Z_DEFLATED, -15, 8, (t / 10) % 5 == 4 ? Z_FIXED : Z_DEFAULT_STRATEGY);
        std::vector<unsigned char> comp(deflateBound(&z, n) + 16);
        z.next_in = data.data(); z.avail_in = n; z.next_out = comp.data(); z.avail_out = comp.size();
        deflate(&z, Z_FINISH);
        size_t cn = z.total_out;
        deflateEnd(&z);
        std::vector<unsigned char> in(comp.begin(), comp.begin() + cn);
        bool corrupt = t % 3 == 0 && cn > 4;
        if (corrupt) for (int q = 0; q < 3; ++q) in[rng() % cn] ^= 1u << (rng() & 7);
        in.resize(cn + fastgz::PAD, 0);
        std::vector<unsigned char> out(n + 258 + 16);
        bool ok = fastgz::inflate_whole(*dec, in.data(), cn, out.data(), n);
        if (!corrupt && (!ok || memcmp(out.data(), data.data(), n))) { ++bad; printf("FAIL %d\n", t); }
        if (corrupt && ok && memcmp(out.data(), data.data(), n)) { /* accepted other bytes: the caller's CRC check catches it */ }
    }
    printf("bad %d\n", bad);
    return bad != 0;
}
