// Host-only check of the index file codec (csrc/index_file.cpp): threaded writer vs the byte-by-byte definition of the
// bincode-2 varint format (src/index.rs:130-164), and the reader on what it wrote.  Built by tests/test_index_file.py.
//   index_file_test <scratch.idx>                      the self-check below
//   index_file_test --roundtrip <given.idx> <out.idx>  read a file somebody else wrote (the real bincode 2.0.1, from
//                                                      tests/golden/crate_vectors.json), write its keys back in the same
//                                                      order and require the same bytes; prints k, w and the keys
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

int dcn_fail(int code, const std::string &msg) {
    std::fprintf(stderr, "dcn_fail(%d): %s\n", code, msg.c_str());
    return code;
}
int dcn_read_index_file(const char *path, uint8_t *k, uint8_t *w, std::vector<uint64_t> *keys);
int dcn_write_index_file(const char *path, uint8_t k, uint8_t w, const uint64_t *keys, uint64_t n);

static void ref_varint(std::vector<uint8_t> &o, uint64_t v) {
    int nb = 0;
    if (v < 251) { o.push_back((uint8_t)v); return; }
    if (v <= 0xFFFF) { o.push_back(0xFB); nb = 2; }
    else if (v <= 0xFFFFFFFFull) { o.push_back(0xFC); nb = 4; }
    else { o.push_back(0xFD); nb = 8; }
    for (int i = 0; i < nb; ++i) o.push_back((uint8_t)(v >> (8 * i)));
}

static std::vector<uint8_t> slurp(const char *path) {
    std::vector<uint8_t> b;
    FILE *f = std::fopen(path, "rb");
    if (!f) return b;
    uint8_t buf[4096];
    size_t m;
    while ((m = std::fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + m);
    std::fclose(f);
    return b;
}

static int roundtrip(const char *given, const char *out) {
    uint8_t k = 0, w = 0;
    std::vector<uint64_t> keys;
    if (dcn_read_index_file(given, &k, &w, &keys) != 0) return 4;
    if (dcn_write_index_file(out, k, w, keys.data(), keys.size()) != 0) return 5;
    if (slurp(given) != slurp(out)) {
        std::fprintf(stderr, "the codec does not write back the bytes it was given\n");
        return 6;
    }
    std::printf("roundtrip ok k=%u w=%u n=%zu\n", (unsigned)k, (unsigned)w, keys.size());
    for (uint64_t v : keys) std::printf("0x%llx\n", (unsigned long long)v);
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 4 && std::string(argv[1]) == "--roundtrip") return roundtrip(argv[2], argv[3]);
    const char *path = argc > 1 ? argv[1] : "/tmp/index_file_test.idx";
    uint64_t x = 88172645463325252ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (uint64_t n : {0ull, 1ull, 250ull, 251ull, 70000ull, 9000000ull}) { // the last one spans two encoder blocks
        std::vector<uint64_t> keys(n);
        for (uint64_t i = 0; i < n; ++i) {
            uint64_t r = rnd();
            switch (r % 11) {  // mostly full-width hashes, some of every shorter encoding and the boundary values
            case 0: keys[i] = r % 251; break;
            case 1: keys[i] = 251 + r % 65285; break;
            case 2: keys[i] = 65536 + (r >> 40); break;
            case 3: keys[i] = (uint64_t[]){0, 250, 251, 65535, 65536, 0xFFFFFFFFull, 0x100000000ull, ~0ull}[(r >> 8) % 8]; break;
            default: keys[i] = r;
            }
        }
        if (dcn_write_index_file(path, 31, 15, keys.data(), n) != 0) return 1;
        std::vector<uint8_t> want = {2, 31, 15};
        ref_varint(want, n);
        for (uint64_t v : keys) ref_varint(want, v);
        std::vector<uint8_t> got(want.size() + 16);
        FILE *f = std::fopen(path, "rb");
        size_t m = std::fread(got.data(), 1, got.size(), f);
        std::fclose(f);
        got.resize(m);
        if (got != want) {
            std::fprintf(stderr, "n=%llu: file differs from the reference encoding (%zu vs %zu bytes)\n", (unsigned long long)n, m, want.size());
            return 2;
        }
        uint8_t k = 0, w = 0;
        std::vector<uint64_t> back;
        if (dcn_read_index_file(path, &k, &w, &back) != 0 || k != 31 || w != 15 || back != keys) {
            std::fprintf(stderr, "n=%llu: read-back differs\n", (unsigned long long)n);
            return 3;
        }
    }
    std::remove(path);
    std::puts("index file codec ok");
    return 0;
}
