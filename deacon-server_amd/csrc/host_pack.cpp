// host_pack.cpp -- host half of the staging path: ASCII -> 2-bit stream + invalid-base bitmask on the CPU, so that
// 0.375 instead of 1 byte per base crosses PCIe (packed_seq::PackedSeqVec::from_ascii + the mask loop of
// get_minimizer_hashes_and_positions, src/filter_common.rs:238-258, for a whole batch buffer at once).
//
// Same layout as pack.hip produces on the device: base i of the batch = bits [2(i%16), +2) of u32 packed[i/16]
// (= bits 2(i%4) of byte i/4, packed-seq's own byte order) and bit i%32 of u32 invmask[i/32].  This is input
// formatting for the GPU pipeline, not a CPU path of the filter: nothing here hashes, probes or decides.
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
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

// 64 bases per step where the host has AVX-512BW: codes (c >> 1) & 3 in every byte, pairs folded by a multiply-add of
// bytes (1, 4), quads by a multiply-add of words (1, 16), the 16 result bytes narrowed out of their dwords; the invalid
// bits come straight out of four byte compares as a 64-bit mask.  About half the instructions per base of the pext form
// (which moves every 8 bytes through a general register).
__attribute__((target("avx512f,avx512bw"))) bool pack_groups_avx512(const uint8_t *src, uint64_t n_groups, uint32_t *packed,
                                                                    uint32_t *mask) {
    const __m512i three = _mm512_set1_epi8(3), lower = _mm512_set1_epi8(0x20), newline = _mm512_set1_epi8('\n');
    const __m512i m14 = _mm512_set1_epi16(0x0401), m116 = _mm512_set1_epi32(0x00100001);
    const __m512i ca = _mm512_set1_epi8('a'), cc = _mm512_set1_epi8('c'), cg = _mm512_set1_epi8('g'), ct = _mm512_set1_epi8('t');
    __mmask64 any_nl = 0;
    uint64_t g = 0;
    for (; g + 2 <= n_groups; g += 2) {
        const __m512i v = _mm512_loadu_si512((const void *)(src + 32 * g));
        const __m512i code = _mm512_and_si512(_mm512_srli_epi16(v, 1), three);
        const __m512i quads = _mm512_madd_epi16(_mm512_maddubs_epi16(code, m14), m116);
        _mm_storeu_si128((__m128i *)(packed + 2 * g), _mm512_cvtepi32_epi8(quads));
        const __m512i l = _mm512_or_si512(v, lower);
        const __mmask64 ok = _mm512_cmpeq_epi8_mask(l, ca) | _mm512_cmpeq_epi8_mask(l, cc) | _mm512_cmpeq_epi8_mask(l, cg) |
                             _mm512_cmpeq_epi8_mask(l, ct);
        const uint64_t inv = ~(uint64_t)ok;
        memcpy(mask + g, &inv, 8);
        any_nl |= _mm512_cmpeq_epi8_mask(v, newline);
    }
    return any_nl != 0;
}

bool have_avx2() {
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    return ok;
}

bool have_avx512() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && !getenv("DCN_NO_AVX512");
    return ok;
}

} // namespace

// whether this host packs with the AVX-512 form (about 1.7 x the AVX2 + pext form's rate on a whole socket share): the host
// entry points then pack page-locked ASCII on the host too instead of sending it as it is (api.hip)
bool dcn_host_pack_is_wide() { return have_avx2() && have_avx512(); }

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
        uint64_t done = 0;
        if (have_avx512()) {
            done = n & ~1ull; // whole pairs of groups; an odd last one goes through the 32-base form
            nl = pack_groups_avx512(ascii + 32 * g0, done, packed, mask);
        }
        if (done < n) nl |= pack_groups_avx2(ascii + 32 * (g0 + done), n - done, packed + 2 * done, mask + done);
        g += n;
    }
    for (; g < g1; ++g) {
        const uint64_t first = 32 * g;
        const uint32_t n = first >= n_bases ? 0u : (uint32_t)(n_bases - first < 32 ? n_bases - first : 32);
        nl |= pack_group_scalar(ascii + first, n, packed + 2 * (g - g0), mask + (g - g0));
    }
    return nl;
}
