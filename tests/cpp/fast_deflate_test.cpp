// The tool's deflate compressor (deacon-server_amd/cli/fast_deflate.hpp) against two decoders -- zlib's inflate and the tool's own
// (fast_inflate.hpp) -- on 4,000 inputs of 0 ... 65,535 bytes of six kinds (bases, random bytes, quality-like, a short period, one
// byte repeated, FASTQ's alphabet), each in a buffer of exactly bound(n) bytes.  Built with -fsanitize=address,undefined by
// tests/test_fast_inflate.py; exit code 0 = every input came back byte for byte from both.
#include "fast_deflate.hpp"
#include "fast_inflate.hpp"
#include <chrono>
#include <cstdio>
#include <random>
#include <memory>
#include <vector>
#include <zlib.h>
static bool inflate_check(const unsigned char *c, size_t cn, const unsigned char *want, size_t n) {
    std::vector<unsigned char> out(n + 16);
    z_stream z{};
    inflateInit2(&z, -15);
    z.next_in = (Bytef *)c; z.avail_in = (uInt)cn; z.next_out = out.data(); z.avail_out = (uInt)out.size();
    int r = inflate(&z, Z_FINISH);
    bool ok = r == Z_STREAM_END && z.total_out == n && z.total_in == cn && (n == 0 || memcmp(out.data(), want, n) == 0);
    inflateEnd(&z);
    return ok;
}
int main() {
    std::unique_ptr<fastgz::FastDeflate> fd(new fastgz::FastDeflate());
    std::unique_ptr<fastgz::BlockDecoder> dec(new fastgz::BlockDecoder());
    int bad = 0;
    // correctness: sizes 0..., contents of several kinds
    std::mt19937_64 rng(3);
    for (int t = 0; t < 4000; ++t) {
        size_t n = t < 300 ? (size_t)t : rng() % 65536;
        std::vector<unsigned char> d(n);
        int mode = t % 6;
        for (size_t i = 0; i < n; ++i)
            d[i] = mode == 0 ? "ACGT"[rng() & 3] : mode == 1 ? (unsigned char)rng() : mode == 2 ? (unsigned char)(33 + rng() % 41) : mode == 3 ? (unsigned char)(i % 7) : mode == 4 ? 'I' : (unsigned char)("ACGT\nI@+"[rng() % 8]);
        std::vector<unsigned char> cbuf(fastgz::FastDeflate::bound(n));
        size_t cn = fd->compress(d.data(), n, cbuf.data());
        bool ok = cn <= fastgz::FastDeflate::bound(n) && inflate_check(cbuf.data(), cn, d.data(), n);
        if (ok) {  // ... and the tool's own decoder
            std::vector<unsigned char> padded(cbuf.begin(), cbuf.begin() + (long)cn), back(n + fastgz::BlockDecoder::SLACK);
            padded.resize(cn + fastgz::PAD, 0);
            ok = fastgz::inflate_whole(*dec, padded.data(), cn, back.data(), n) && (n == 0 || memcmp(back.data(), d.data(), n) == 0);
        }
        if (!ok) { ++bad; printf("FAIL t=%d n=%zu mode=%d cn=%zu\n", t, n, mode, cn); if (bad > 5) return 1; }
    }
    printf("random cases bad %d\n", bad);
    return bad != 0;
}
