// host_pack.cpp -- host half of the staging path: ASCII -> 2-bit stream + invalid-base bitmask on the CPU, so that
// 0.375 instead of 1 byte per base crosses PCIe (packed_seq::PackedSeqVec::from_ascii + the mask loop of
// get_minimizer_hashes_and_positions, src/filter_common.rs:238-258, for a whole batch buffer at once).
//
// Same layout as pack.hip produces on the device: base i of the batch = bits [2(i%16), +2) of u32 packed[i/16]
// (= bits 2(i%4) of byte i/4, packed-seq's own byte order) and bit i%32 of u32 invmask[i/32].  This is input
// formatting for the GPU pipeline, not a CPU path of the filter: nothing here hashes, probes or decides.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <immintrin.h>

namespace {

inline uint32_t pack4_scalar(uint32_t x) {
    uint32_t y = (x >> 1) & 0x03030303u;
    return (y * ((1u << 24) | (1u << 18) | (1u << 12) | (1u << 6))) >> 24;
}

inline uint32_t invalid_scalar(uint8_t c) {
    c |= 0x20;
    return (c == 'a' || c == 'c' || c == 'g' || c == 't') ? 0u : 1u;
}

bool pack_group_scalar(const uint8_t *src, uint32_t n, uint32_t *packed2, uint32_t *mask1) {
    uint8_t buf[32];
    const bool nl = n > 0 && memchr(src, '\n', n) != nullptr;
    if (n < 32) {
        memset(buf, 'A', sizeof buf);
        memcpy(buf, src, n);
        src = buf;
    }
    uint32_t p0 = 0, p1 = 0, m = 0;
    for (int q = 0; q < 4; ++q) {
        uint32_t a, b;
        memcpy(&a, src + 4 * q, 4);
        memcpy(&b, src + 16 + 4 * q, 4);
        p0 |= pack4_scalar(a) << (8 * q);
        p1 |= pack4_scalar(b) << (8 * q);
    }
    for (int i = 0; i < 32; ++i) m |= invalid_scalar(src[i]) << i;
    packed2[0] = p0;
    packed2[1] = p1;
    *mask1 = m;
    return nl;
}

__attribute__((target("avx2,bmi2"))) bool pack_groups_avx2(const uint8_t *src, uint64_t n_groups, uint32_t *packed,
                                                            uint32_t *mask) {
    const __m256i lower = _mm256_set1_epi8(0x20), newline = _mm256_set1_epi8('\n');
    __m256i any_nl = _mm256_setzero_si256();
    const __m256i ca = _mm256_set1_epi8('a'), cc = _mm256_set1_epi8('c'), cg = _mm256_set1_epi8('g'),
                  ct = _mm256_set1_epi8('t');
    const uint64_t sel = 0x0606060606060606ull; // bits 1..2 of every byte = (c >> 1) & 3
    for (uint64_t g = 0; g < n_groups; ++g) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(src + 32 * g));
        const uint64_t q0 = _pext_u64((uint64_t)_mm256_extract_epi64(v, 0), sel);
        const uint64_t q1 = _pext_u64((uint64_t)_mm256_extract_epi64(v, 1), sel);
        const uint64_t q2 = _pext_u64((uint64_t)_mm256_extract_epi64(v, 2), sel);
        const uint64_t q3 = _pext_u64((uint64_t)_mm256_extract_epi64(v, 3), sel);
        const uint64_t both = q0 | (q1 << 16) | (q2 << 32) | (q3 << 48);
        memcpy(packed + 2 * g, &both, 8);
        const __m256i l = _mm256_or_si256(v, lower);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(l, ca), _mm256_cmpeq_epi8(l, cc)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(l, cg), _mm256_cmpeq_epi8(l, ct)));
        mask[g] = ~(uint32_t)_mm256_movemask_epi8(ok);
        any_nl = _mm256_or_si256(any_nl, _mm256_cmpeq_epi8(v, newline));
    }
    return _mm256_movemask_epi8(any_nl) != 0;
}

bool have_avx2() {
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    return ok;
}

} // namespace

// Packs the 32-base groups [g0, g1) of the stream `ascii` (n_bases bytes; bytes at or past n_bases count as 'A',
// valid) into packed[2 * (g - g0) ..] and mask[g - g0].  Returns whether any of those bytes is '\n'.
bool dcn_host_pack_groups(const uint8_t *ascii, uint64_t n_bases, uint64_t g0, uint64_t g1, uint32_t *packed,
                          uint32_t *mask) {
    if (g1 <= g0) return false;
    bool nl = false;
    const uint64_t full_end = n_bases / 32; // groups below this one are complete
    uint64_t g = g0;
    if (have_avx2() && full_end > g0) {
        const uint64_t n = (full_end < g1 ? full_end : g1) - g0;
        nl = pack_groups_avx2(ascii + 32 * g0, n, packed, mask);
        g += n;
    }
    for (; g < g1; ++g) {
        const uint64_t first = 32 * g;
        const uint32_t n = first >= n_bases ? 0u : (uint32_t)(n_bases - first < 32 ? n_bases - first : 32);
        nl |= pack_group_scalar(ascii + first, n, packed + 2 * (g - g0), mask + (g - g0));
    }
    return nl;
}
