// dcn_probe.h -- device helpers shared by the kernels: set lookup, XXH3-64 of an 8/16-byte k-mer,
// canonical k-mer extraction from the 2-bit stream.
#pragma once

#include "dcn_internal.h"

// ---- set lookup (FxHashSet::contains, src/filter_common.rs:144,185) ---------------------------------
struct dcn_group {
    ulonglong2 a;
#if DCN_GROUP_SLOTS == 4
    ulonglong2 b;
#endif
};

__device__ inline dcn_group dcn_load_group(const dcn_table_view &t, uint32_t g) {
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(t.slots + (uint64_t)g * DCN_GROUP_SLOTS);
    dcn_group r;
#ifdef DCN_NT_PROBE
    typedef unsigned long long dcn_u64x2 __attribute__((ext_vector_type(2)));
    dcn_u64x2 q = __builtin_nontemporal_load(reinterpret_cast<const dcn_u64x2 *>(p));
    r.a.x = q.x;
    r.a.y = q.y;
#else
    r.a = p[0];
#endif
#if DCN_GROUP_SLOTS == 4
    r.b = p[1];
#endif
    return r;
}

// 1 = key present in this group, 0 = group has an empty slot (key absent), -1 = walk on
__device__ inline int dcn_group_resolve(const dcn_group &g, uint64_t key) {
#if DCN_GROUP_SLOTS == 4
    if (g.a.x == key || g.a.y == key || g.b.x == key || g.b.y == key) return 1;
    if (g.a.x == 0 || g.a.y == 0 || g.b.x == 0 || g.b.y == 0) return 0;
#else
    if (g.a.x == key || g.a.y == key) return 1;
    if (g.a.x == 0 || g.a.y == 0) return 0;
#endif
    return -1;
}

__device__ inline bool dcn_table_contains_dev(const dcn_table_view &t, uint64_t key) {
    if (key == 0) return t.has_zero != 0;
    uint32_t g = dcn_group_of(key, t.group_shift, t.group_mask);
    for (;;) {
        int r = dcn_group_resolve(dcn_load_group(t, g), key);
        if (r >= 0) return r == 1;
        g = (g + 1) & t.group_mask;
    }
}

// ---- XXH3-64, seed 0, 8-byte and 16-byte inputs (xxh3_64(&kmer.to_le_bytes()),
//      src/filter_common.rs:296,305) -------------------------------------------------------------------
__device__ inline uint64_t dcn_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

__device__ inline uint64_t dcn_xxh3_u64(uint64_t v) {
    uint64_t x = dcn_rotl64(v, 32) ^ 0xC73AB174C5ECD5A2ull; // secret[8..16) ^ secret[16..24)
    x ^= dcn_rotl64(x, 49) ^ dcn_rotl64(x, 24);
    x *= 0x9FB21C651E98DF25ull;
    x ^= (x >> 35) + 8;
    x *= 0x9FB21C651E98DF25ull;
    return x ^ (x >> 28);
}

__device__ inline uint64_t dcn_xxh3_u128(uint64_t v_lo, uint64_t v_hi) {
    uint64_t lo = v_lo ^ 0x6782737BEA4239B9ull; // secret[24..32) ^ secret[32..40)
    uint64_t hi = v_hi ^ 0xAF56BC3B0996523Aull; // secret[40..48) ^ secret[48..56)
    uint64_t fold = (lo * hi) ^ __umul64hi(lo, hi);
    uint64_t acc = 16 + __builtin_bswap64(lo) + hi + fold;
    acc ^= acc >> 37;
    acc *= 0x165667919E3779F9ull;
    return acc ^ (acc >> 32);
}

// ---- canonical k-mer value from the packed stream (read_kmer / read_revcomp_kmer + min,
//      src/filter_common.rs:289-307) -----------------------------------------------------------------------

// reverse the order of the 32 two-bit groups of x and complement every base (code ^ 2)
__device__ inline uint64_t dcn_revcomp64(uint64_t x) {
    uint64_t r = __brevll(x);
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    return r ^ 0xAAAAAAAAAAAAAAAAull;
}

// bits [2p, 2p+64) of the packed stream (p = absolute base index)
__device__ inline uint64_t dcn_packed_u64(const uint32_t *packed, uint64_t p) {
    uint64_t wi = p >> 4;
    uint32_t sh = (uint32_t)(p & 15) * 2;
    uint32_t w0 = packed[wi], w1 = packed[wi + 1], w2 = packed[wi + 2];
    uint32_t lo = __funnelshift_r(w0, w1, sh);
    uint32_t hi = __funnelshift_r(w1, w2, sh);
    return ((uint64_t)hi << 32) | lo;
}

// hash of the canonical k-mer whose 2k bits start at bit 0 of `bits` (upper bits arbitrary), k <= 32
__device__ inline uint64_t dcn_kmer_hash64_bits(uint64_t bits, uint32_t k) {
    uint32_t sh = 64 - 2 * k;
    uint64_t a = (bits << sh) >> sh;
    uint64_t b = dcn_revcomp64(a) >> sh;
    return dcn_xxh3_u64(a < b ? a : b);
}

// hash of the canonical k-mer starting at absolute base p, k <= 32
__device__ inline uint64_t dcn_kmer_hash64(const uint32_t *packed, uint64_t p, uint32_t k) {
    return dcn_kmer_hash64_bits(dcn_packed_u64(packed, p), k);
}

__device__ inline uint64_t dcn_kmer_hash128_bits(uint64_t lo, uint64_t hi, uint32_t k);

// hash of the canonical k-mer starting at absolute base p, 32 < k <= 56 (u128 value, 16-byte hash)
__device__ inline uint64_t dcn_kmer_hash128(const uint32_t *packed, uint64_t p, uint32_t k) {
    return dcn_kmer_hash128_bits(dcn_packed_u64(packed, p), dcn_packed_u64(packed, p + 32), k);
}

// same from the two 64-bit words holding the k-mer's 2k bits (bits above 2k arbitrary)
__device__ inline uint64_t dcn_kmer_hash128_bits(uint64_t lo, uint64_t hi, uint32_t k) {
    uint32_t hb = 2 * k - 64; // valid bits in the high word, 2..48
    hi &= (~0ull) >> (64 - hb);
    // reverse complement of the 2k-bit value: reverse both words, swap them, shift right by 128-2k
    uint64_t rl = dcn_revcomp64(hi), rh = dcn_revcomp64(lo); // (rh:rl) = revcomp of 128-bit (hi:lo)
    uint32_t sh = 128 - 2 * k;                                // 16..62
    uint64_t blo = (rl >> sh) | (rh << (64 - sh));
    uint64_t bhi = rh >> sh;
    bool a_less = (hi < bhi) || (hi == bhi && lo < blo);
    return a_less ? dcn_xxh3_u128(lo, hi) : dcn_xxh3_u128(blo, bhi);
}

// all k bases starting at absolute base p are ACGT: mask bits [p, p+k) zero (src/filter_common.rs:275-286)
__device__ inline bool dcn_kmer_valid(const uint32_t *invmask, uint64_t p, uint32_t k) {
    uint64_t wi = p >> 5;
    uint32_t sh = (uint32_t)(p & 31);
    uint32_t w0 = invmask[wi], w1 = invmask[wi + 1], w2 = invmask[wi + 2];
    uint64_t bits = ((uint64_t)__funnelshift_r(w1, w2, sh) << 32) | __funnelshift_r(w0, w1, sh);
    return (bits & ((~0ull) >> (64 - k))) == 0;
}
