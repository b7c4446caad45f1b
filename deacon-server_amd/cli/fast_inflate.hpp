// fast_inflate.hpp -- gzip members decoded faster than zlib's inflate() does it, for the tool's input side.
//
// Why: a gzip input is ONE deflate stream, so it is decoded by one thread, and everything behind it -- parser pool, GPU,
// formatters -- waits for that thread (zlib 1.2.11: 0.66 GB/s of FASTQ on the box; the GPU stage takes 150 GB/s).  The
// reference reads through niffler / flate2, which is bound the same way (src/local_filter.rs:41-55).  Most reads in the
// world sit in .fastq.gz files.
//
// What: a raw-deflate decoder (RFC 1951) in the manner of the fast decoders that exist for it (a 64-bit bit buffer refilled
// with one unaligned load, an 11-bit table for literal / length codes and a 9-bit one for distance codes that yield symbol,
// extra-bit count and code length in one lookup, longer codes finished canonically bit by bit, matches copied eight bytes at
// a time), a gzip member layer (RFC 1952: header with its optional fields, CRC-32 and length of every member checked) and a
// CRC-32 on carry-less multiplication where the CPU has it.  Whole deflate BLOCKS are decoded at a time into a buffer that
// keeps the last 32 KB in front of the write position; a block that runs out of input or of room is simply decoded again
// from its first bit once there is more of either, so the inner loops carry no resumable state.
//
// Every member's CRC-32 and length are checked, as zlib does: a decoding mistake cannot pass silently.  DCN_CLI_ZLIB_INFLATE=1
// makes the tool use zlib's decoder instead (tests compare the two byte for byte).
#ifndef DCN_FAST_INFLATE_HPP
#define DCN_FAST_INFLATE_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace fastgz {

// ---- CRC-32 (the gzip polynomial, reflected) -----------------------------------------------------------------------------
// Folding with PCLMULQDQ: four 128-bit lanes folded over 64 bytes per step, then down to one lane, then Barrett reduction.
// The constants are x^(n) mod P for the distances folded over, in the reflected domain.  Buffers shorter than 64 bytes, the
// unaligned head and the tail go through zlib's crc32().
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) inline __m128i crc_fold1(__m128i a, __m128i b, __m128i k3k4) {
    const __m128i t = _mm_clmulepi64_si128(a, k3k4, 0x00);
    a = _mm_clmulepi64_si128(a, k3k4, 0x11);
    return _mm_xor_si128(_mm_xor_si128(a, t), b);
}
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul(uint32_t crc, const unsigned char *p, size_t n) {
    const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596ll, 0x0000000154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009ell, 0x00000001751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0000000163cd6124ll);
    const __m128i poly_mu = _mm_set_epi64x(0x00000001f7011641ll, 0x00000001db710641ll);
    __m128i x0 = _mm_loadu_si128((const __m128i *)(p + 0)), x1 = _mm_loadu_si128((const __m128i *)(p + 16)),
            x2 = _mm_loadu_si128((const __m128i *)(p + 32)), x3 = _mm_loadu_si128((const __m128i *)(p + 48));
    x0 = _mm_xor_si128(x0, _mm_cvtsi32_si128((int)~crc));
    p += 64;
    n -= 64;
    while (n >= 64) {
        __m128i t0 = _mm_clmulepi64_si128(x0, k1k2, 0x00), t1 = _mm_clmulepi64_si128(x1, k1k2, 0x00),
                t2 = _mm_clmulepi64_si128(x2, k1k2, 0x00), t3 = _mm_clmulepi64_si128(x3, k1k2, 0x00);
        x0 = _mm_clmulepi64_si128(x0, k1k2, 0x11);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x0 = _mm_xor_si128(_mm_xor_si128(x0, t0), _mm_loadu_si128((const __m128i *)(p + 0)));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, t1), _mm_loadu_si128((const __m128i *)(p + 16)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, t2), _mm_loadu_si128((const __m128i *)(p + 32)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, t3), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64;
        n -= 64;
    }
    x0 = crc_fold1(x0, x1, k3k4);
    x0 = crc_fold1(x0, x2, k3k4);
    x0 = crc_fold1(x0, x3, k3k4);
    while (n >= 16) {
        x0 = crc_fold1(x0, _mm_loadu_si128((const __m128i *)p), k3k4);
        p += 16;
        n -= 16;
    }
    // 128 -> 64 bits
    const __m128i mask32 = _mm_set_epi32(0, 0, 0, ~0);
    __m128i t = _mm_clmulepi64_si128(x0, k3k4, 0x10);
    x0 = _mm_xor_si128(_mm_srli_si128(x0, 8), t);
    t = _mm_and_si128(x0, mask32);
    x0 = _mm_srli_si128(x0, 4);
    t = _mm_clmulepi64_si128(t, k5, 0x00);
    x0 = _mm_xor_si128(x0, t);
    // Barrett reduction 64 -> 32 bits
    t = _mm_and_si128(x0, mask32);
    t = _mm_clmulepi64_si128(t, poly_mu, 0x10);
    t = _mm_and_si128(t, mask32);
    t = _mm_clmulepi64_si128(t, poly_mu, 0x00);
    x0 = _mm_xor_si128(x0, t);
    uint32_t c = ~(uint32_t)_mm_extract_epi32(x0, 1);
    return n ? (uint32_t)::crc32(c, p, (uInt)n) : c;
}
#endif

inline uint32_t crc32_fast(uint32_t crc, const unsigned char *p, size_t n) {
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have && n >= 64) {
        while (n > (1u << 30)) {
            crc = crc32_clmul(crc, p, 1u << 30);
            p += 1u << 30;
            n -= 1u << 30;
        }
        return n >= 64 ? crc32_clmul(crc, p, n) : (uint32_t)::crc32(crc, p, (uInt)n);
    }
#endif
    while (n > (1u << 30)) {
        crc = (uint32_t)::crc32(crc, p, 1u << 30);
        p += 1u << 30;
        n -= 1u << 30;
    }
    return (uint32_t)::crc32(crc, p, (uInt)n);
}

// ---- raw deflate, one block at a time ------------------------------------------------------------------------------------
enum Status { OK = 0, NEED_INPUT = 1, NEED_ROOM = 2, BAD = 3 };
// Input buffers are readable this many bytes past their end: the decoder loads eight bytes at a time and tests whether the
// bits it consumed were there once per symbol group, by when the read position is at most 7 (bits loaded, not yet known to
// be missing) + 6 (one length/distance pair) + 7 (the loads in between) + 8 bytes past the end.
constexpr size_t PAD = 64;

// T = the type of an output element: unsigned char for bytes; uint16_t where a decoder starts in the middle of a stream and
// what lies before its start is not known yet (parallel_gzip.hpp: the 32 K elements in front of the output are then numbered
// markers, and a match that reaches back to them copies markers instead of bytes).
template <typename T>
class BlockDecoderT {
  public:
    static constexpr int LIT_BITS = 11, DIST_BITS = 9;
    static constexpr size_t SLACK = 258 + 16;  // elements behind out_end that may be written
    struct Bits {
        const unsigned char *in;
        uint64_t buf;
        unsigned cnt;
    };

    // One deflate block starting at bit state `b` (updated on OK), written at `out` (advanced on OK).  `out_begin` is the oldest
    // byte a match may reach (>= 32 KB before `out` once that much has been produced), `out_end` the end of the buffer, `in_end`
    // the end of the input available; the input buffer must be readable for PAD bytes past in_end.  `final` returns BFINAL.
    Status block(Bits &b, const unsigned char *in_end, T *out_begin, T *&out, T *out_end, bool &final) {
        Bits s = b;
        T *o = out;
        fill(s, 3);
        final = (s.buf & 1) != 0;
        const unsigned type = (unsigned)(s.buf >> 1) & 3;
        drop(s, 3);
        if (overran(s, in_end)) return NEED_INPUT;
        Status st;
        if (type == 0) st = stored(s, in_end, o, out_end);
        else if (type == 1) {
            fixed_tables();
            st = huffman(s, in_end, out_begin, o, out_end, fixed_lit_, fixed_dist_, fixed_lit_long_, fixed_dist_long_);
        } else if (type == 2) {
            st = dynamic_tables(s, in_end);
            if (st == OK) st = huffman(s, in_end, out_begin, o, out_end, lit_, dist_, lit_long_, dist_long_);
        } else st = BAD;
        if (st == BAD && overran(s, in_end)) st = NEED_INPUT;  // (what was decoded came out of the padding)
        if (st != OK) return st;
        b = s;
        out = o;
        return OK;
    }
    static void start(Bits &b, const unsigned char *in) {
        b.in = in;
        b.buf = 0;
        b.cnt = 0;
    }
    // the same at a bit position of the buffer `base` (a deflate block starts at any bit)
    static void start_at_bit(Bits &b, const unsigned char *base, uint64_t bit) {
        start(b, base + (bit >> 3));
        refill(b);
        drop(b, (unsigned)(bit & 7));
    }
    static uint64_t bit_position(const Bits &b, const unsigned char *base) { return (uint64_t)(b.in - base) * 8 - b.cnt; }
    // the byte position behind the bits consumed so far, rounded up to a whole byte (end of a deflate stream)
    static const unsigned char *byte_position(const Bits &b) { return b.in - (b.cnt >> 3); }

  private:
    struct Long {  // canonical decoding of the codes that do not fit the table
        uint16_t count[16], first[16], offset[16];
        uint16_t sorted[288];
    };
    static inline void refill(Bits &s) {
        uint64_t w;
        std::memcpy(&w, s.in, 8);
        s.buf |= w << s.cnt;
        s.in += (63 - s.cnt) >> 3;
        s.cnt |= 56;
    }
    static inline void drop(Bits &s, unsigned n) {
        s.buf >>= n;
        s.cnt -= n;
    }
    // at least n (<= 56) bits in the buffer.  Whether they were bits of the input (and not of the padding behind it) is asked
    // afterwards, of what was CONSUMED (overran): asking for n bits ahead of need would refuse a stream whose last symbols are
    // shorter than the longest a lookup may see -- the last code lengths of a 15-byte member, say
    static inline void fill(Bits &s, unsigned n) {
        if (s.cnt < n) refill(s);
    }
    static inline bool overran(const Bits &s, const unsigned char *in_end) {  // consumed bits that the input does not hold
        return (int64_t)(s.in - in_end) * 8 > (int64_t)s.cnt;
    }

    Status stored(Bits &s, const unsigned char *in_end, T *&o, T *out_end) {
        drop(s, s.cnt & 7);
        const unsigned char *p = s.in - (s.cnt >> 3);
        if (in_end - p < 4) return NEED_INPUT;
        const unsigned len = p[0] | (unsigned)p[1] << 8, nlen = p[2] | (unsigned)p[3] << 8;
        if ((len ^ nlen) != 0xFFFF) return BAD;
        p += 4;
        if ((size_t)(in_end - p) < len) return NEED_INPUT;
        if ((size_t)(out_end - o) < len) return NEED_ROOM;
        if (sizeof(T) == 1) std::memcpy(o, p, len);
        else
            for (unsigned i = 0; i < len; ++i) o[i] = (T)p[i];
        o += len;
        start(s, p + len);
        return OK;
    }

    // code lengths -> table of 2^bits entries + the canonical arrays for longer codes.  kind: 0 = literal / length alphabet,
    // 1 = distance alphabet.  false: over-subscribed, or incomplete with a code longer than one bit -- zlib's rule
    // (inftrees.c: "incomplete set" unless max == 1); the unused pattern of a one-code set decodes as BAD
    static bool build(const uint8_t *lens, int n, int bits, int kind, uint32_t *table, Long &lg, bool strict = true) {
        static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        std::memset(lg.count, 0, sizeof lg.count);
        for (int i = 0; i < n; ++i) lg.count[lens[i]]++;
        lg.count[0] = 0;
        int left = 1, max_len = 0;
        for (int l = 1; l <= 15; ++l) {
            left = (left << 1) - lg.count[l];
            if (left < 0) return false;
            if (lg.count[l]) max_len = l;
        }
        if (strict && left > 0 && max_len > 1) return false;
        uint16_t off = 0, code = 0;
        for (int l = 1; l <= 15; ++l) {
            code = (uint16_t)((code + lg.count[l - 1]) << 1);
            lg.first[l] = code;
            lg.offset[l] = off;
            off = (uint16_t)(off + lg.count[l]);
        }
        uint16_t next[16];
        for (int l = 1; l <= 15; ++l) next[l] = lg.offset[l];
        for (int i = 0; i < n; ++i)
            if (lens[i]) lg.sorted[next[lens[i]]++] = (uint16_t)i;
        const int size = 1 << bits;
        std::memset(table, 0, sizeof(uint32_t) * (size_t)size);
        // entry: bits 0..7 code length (0 = no code of <= table bits ends here), 8..15 extra bits, 16..23 literal byte / length base - 3 /
        // distance symbol, 24..25 kind: 0 literal, 1 length or distance, 2 end of block, 3 longer than the table (finished bit by bit)
        auto entry_of = [&](int sym, int l) -> uint32_t {
            if (kind == 0) {
                if (sym < 256) return (uint32_t)sym << 16 | (uint32_t)l;
                if (sym == 256) return 2u << 24 | (uint32_t)l;
                if (sym > 285) return 0;  // 286, 287: never valid in a stream
                return 1u << 24 | (uint32_t)(len_base[sym - 257] - 3) << 16 | (uint32_t)len_extra[sym - 257] << 8 | (uint32_t)l;
            }
            if (sym > 29) return 0;
            return 1u << 24 | (uint32_t)sym << 16 | (uint32_t)dist_extra[sym] << 8 | (uint32_t)l;
        };
        (void)dist_base;
        // codes of <= bits bits: every table index whose low l bits are the reversed code
        for (int l = 1; l <= 15; ++l) {
            for (int j = 0; j < lg.count[l]; ++j) {
                const int sym = lg.sorted[lg.offset[l] + j];
                const unsigned c = (unsigned)lg.first[l] + (unsigned)j;
                if (l <= bits) {
                    unsigned rev = 0;
                    for (int q = 0; q < l; ++q) rev |= ((c >> q) & 1u) << (l - 1 - q);
                    const uint32_t e = entry_of(sym, l);
                    for (int i = (int)rev; i < size; i += 1 << l) table[i] = e;
                } else {
                    // mark the prefix: the first `bits` bits of the code (most significant first) reversed into an index
                    const unsigned prefix = c >> (l - bits);
                    unsigned rev = 0;
                    for (int q = 0; q < bits; ++q) rev |= ((prefix >> q) & 1u) << (bits - 1 - q);
                    table[rev] = 3u << 24;
                }
            }
        }
        return true;
    }
    // Two literals per lookup where both codes fit the table's bits: an index whose low l1 bits are a literal's code and whose
    // next l2 <= bits - l1 bits are another literal's becomes kind 4 (first byte in bits 16..23, second in 8..15, length l1 + l2).
    // Streams of literals -- quality strings -- are a chain of dependent lookups; this halves the chain.
    static void pair_literals(uint32_t *table, int bits, uint32_t *scratch) {
        const int size = 1 << bits;
        std::memcpy(scratch, table, sizeof(uint32_t) * (size_t)size);
        for (int i = 0; i < size; ++i) {
            const uint32_t e1 = scratch[i];
            if ((e1 >> 24) != 0 || (e1 & 0xFF) == 0) continue;
            const int l1 = (int)(e1 & 0xFF);
            if (l1 >= bits) continue;
            const uint32_t e2 = scratch[(unsigned)i >> l1];  // (the bits above the known ones are zero: only a code of <= bits - l1 bits is decided by them)
            const int l2 = (int)(e2 & 0xFF);
            if ((e2 >> 24) != 0 || l2 == 0 || l1 + l2 > bits) continue;
            table[i] = 4u << 24 | (e1 & 0x00FF0000u) | ((e2 >> 16) & 0xFF) << 8 | (uint32_t)(l1 + l2);
        }
    }

    // a code longer than the table: the canonical way, one bit at a time (rare: < 0.1 % of the symbols of real streams)
    static inline int long_symbol(Bits &s, const Long &lg, int bits) {
        unsigned code = 0;
        // the first `bits` bits are in the buffer, least significant first
        for (int q = 0; q < bits; ++q) code = code << 1 | (unsigned)((s.buf >> q) & 1u);
        for (int l = bits + 1; l <= 15; ++l) {
            code = code << 1 | (unsigned)((s.buf >> (l - 1)) & 1u);
            const unsigned rel = code - lg.first[l];
            if (code >= lg.first[l] && rel < lg.count[l]) {
                drop(s, (unsigned)l);
                return lg.sorted[lg.offset[l] + rel];
            }
        }
        return -1;
    }

    void fixed_tables() {
        if (fixed_ready_) return;
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        build(l, 288, LIT_BITS, 0, fixed_lit_, fixed_lit_long_);
        pair_literals(fixed_lit_, LIT_BITS, scratch_);
        uint8_t d[30];
        for (int i = 0; i < 30; ++i) d[i] = 5;
        build(d, 30, DIST_BITS, 1, fixed_dist_, fixed_dist_long_, false);  // (30 of the 32 five-bit codes: incomplete by definition)
        fixed_ready_ = true;
    }

    Status dynamic_tables(Bits &s, const unsigned char *in_end) {
        fill(s, 14);
        const int hlit = (int)(s.buf & 31) + 257, hdist = (int)((s.buf >> 5) & 31) + 1, hclen = (int)((s.buf >> 10) & 15) + 4;
        drop(s, 14);
        if (hlit > 286 || hdist > 30) return BAD;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; ++i) {
            fill(s, 3);
            cl[order[i]] = (uint8_t)(s.buf & 7);
            drop(s, 3);
        }
        if (overran(s, in_end)) return NEED_INPUT;
        uint32_t ct[128];
        Long clong;
        if (!build_plain(cl, 19, 7, ct, clong)) return BAD;
        uint8_t lens[286 + 30 + 138];
        int i = 0;
        while (i < hlit + hdist) {
            if (s.in > in_end && overran(s, in_end)) return NEED_INPUT;  // (never more than a symbol into the padding)
            fill(s, 7 + 7);
            const uint32_t e = ct[s.buf & 127];
            const int l = (int)(e & 0xFF);
            if (l == 0) return BAD;
            const int sym = (int)(e >> 16);
            drop(s, (unsigned)l);
            if (sym < 16) {
                lens[i++] = (uint8_t)sym;
            } else {
                int rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return BAD;
                    val = lens[i - 1];
                    rep = 3 + (int)(s.buf & 3);
                    drop(s, 2);
                } else if (sym == 17) {
                    rep = 3 + (int)(s.buf & 7);
                    drop(s, 3);
                } else {
                    rep = 11 + (int)(s.buf & 127);
                    drop(s, 7);
                }
                if (i + rep > hlit + hdist) return BAD;
                while (rep--) lens[i++] = (uint8_t)val;
            }
        }
        if (overran(s, in_end)) return NEED_INPUT;
        if (lens[256] == 0) return BAD;  // no end-of-block code
        if (!build(lens, hlit, LIT_BITS, 0, lit_, lit_long_)) return BAD;
        pair_literals(lit_, LIT_BITS, scratch_);
        if (!build(lens + hlit, hdist, DIST_BITS, 1, dist_, dist_long_)) return BAD;
        return OK;
    }
    // plain symbol table (code-length alphabet): entry = symbol << 16 | length; every code fits (<= 7 bits)
    static bool build_plain(const uint8_t *lens, int n, int bits, uint32_t *table, Long &lg) {
        std::memset(lg.count, 0, sizeof lg.count);
        for (int i = 0; i < n; ++i) lg.count[lens[i]]++;
        lg.count[0] = 0;
        int left = 1;
        for (int l = 1; l <= bits; ++l) {
            left = (left << 1) - lg.count[l];
            if (left < 0) return false;
        }
        for (int l = bits + 1; l <= 15; ++l)
            if (lg.count[l]) return false;
        if (left != 0) return false;  // (zlib: an incomplete code-length code is always an error)
        uint16_t code = 0, nextc[16];
        for (int l = 1; l <= bits; ++l) {
            code = (uint16_t)((code + lg.count[l - 1]) << 1);
            nextc[l] = code;
        }
        std::memset(table, 0, sizeof(uint32_t) << bits);
        for (int i = 0; i < n; ++i) {
            const int l = lens[i];
            if (!l) continue;
            const unsigned c = nextc[l]++;
            unsigned rev = 0;
            for (int q = 0; q < l; ++q) rev |= ((c >> q) & 1u) << (l - 1 - q);
            for (int j = (int)rev; j < (1 << bits); j += 1 << l) table[j] = (uint32_t)i << 16 | (uint32_t)l;
        }
        return true;
    }

    Status huffman(Bits &s, const unsigned char *in_end, T *out_begin, T *&o, T *out_end, const uint32_t *lit, const uint32_t *dist,
                   const Long &lit_long, const Long &dist_long) {
        static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        const uint32_t lit_mask = (1u << LIT_BITS) - 1, dist_mask = (1u << DIST_BITS) - 1;
        // the input is padded, so loads never fault; whether the bits consumed really existed is checked once per symbol group
        // against in_end (a block cut short decodes garbage from the padding, is refused, and is decoded again with more input)
        T *const o_safe = out_end - SLACK;
        for (;;) {
            if (o > o_safe) return NEED_ROOM;
            refill(s);
            uint32_t e = lit[s.buf & lit_mask];
            unsigned kind = e >> 24;
            if (kind == 0 || kind == 4) {
                if ((e & 0xFF) == 0) return overran(s, in_end) ? NEED_INPUT : BAD;
                // literals: up to four lookups per refill (4 x 11 bits < 56; a single literal of a lookup is <= 11 bits as well),
                // each one or two bytes; the second byte of a single is written and overwritten (o moves by the count)
                for (int q = 0;;) {
                    o[0] = (T)(unsigned char)(e >> 16);
                    o[1] = (T)(unsigned char)(e >> 8);
                    o += 1 + (e >> 26);  // kind 4 -> 2 bytes, kind 0 -> 1
                    drop(s, e & 0xFF);
                    if (++q == 4) break;
                    e = lit[s.buf & lit_mask];
                    const unsigned k2 = e >> 24;
                    if ((k2 != 0 && k2 != 4) || (e & 0xFF) == 0) break;
                }
                if (s.in > in_end && overran(s, in_end)) return NEED_INPUT;
                continue;
            }
            unsigned length;
            if (kind == 3) {
                const int sym = long_symbol(s, lit_long, LIT_BITS);
                if (sym < 0) return overran(s, in_end) ? NEED_INPUT : BAD;
                if (sym < 256) {
                    *o++ = (T)sym;
                    if (s.in > in_end && overran(s, in_end)) return NEED_INPUT;
                    continue;
                }
                if (sym == 256) break;
                if (sym > 285) return BAD;
                refill(s);
                const unsigned eb = len_extra[sym - 257];
                length = len_base[sym - 257] + (unsigned)(s.buf & ((1u << eb) - 1));
                drop(s, eb);
            } else if (kind == 2) {
                drop(s, e & 0xFF);
                break;
            } else {
                drop(s, e & 0xFF);
                const unsigned eb = (e >> 8) & 0xFF;
                length = ((e >> 16) & 0xFF) + 3 + (unsigned)(s.buf & ((1u << eb) - 1));
                drop(s, eb);
            }
            if (s.cnt < 32) refill(s);
            uint32_t d = dist[s.buf & dist_mask];
            int dsym;
            if ((d >> 24) == 3) {
                dsym = long_symbol(s, dist_long, DIST_BITS);
                if (dsym < 0 || dsym > 29) return overran(s, in_end) ? NEED_INPUT : BAD;
                if (s.cnt < 16) refill(s);
            } else {
                if ((d & 0xFF) == 0) return overran(s, in_end) ? NEED_INPUT : BAD;
                dsym = (int)((d >> 16) & 0xFF);
                drop(s, d & 0xFF);
            }
            const unsigned deb = dist_extra[dsym];
            const unsigned distance = dist_base[dsym] + (unsigned)(s.buf & ((1u << deb) - 1));
            drop(s, deb);
            if (s.in > in_end && overran(s, in_end)) return NEED_INPUT;
            if ((size_t)(o - out_begin) < distance) return BAD;
            const T *src = o - distance;
            T *const end = o + length;
            if (distance >= 8) {
                do {
                    std::memcpy(o, src, 8 * sizeof(T));
                    o += 8;
                    src += 8;
                } while (o < end);
            } else if (distance == 1) {
                if (sizeof(T) == 1) std::memset(o, (int)*src, length);
                else std::fill_n(o, length, *src);
            } else {
                do *o++ = *src++;
                while (o < end);
            }
            o = end;
        }
        if (overran(s, in_end)) return NEED_INPUT;
        return OK;
    }
    uint32_t lit_[1 << LIT_BITS], dist_[1 << DIST_BITS], scratch_[1 << LIT_BITS];
    uint32_t fixed_lit_[1 << LIT_BITS], fixed_dist_[1 << DIST_BITS];
    Long lit_long_, dist_long_, fixed_lit_long_, fixed_dist_long_;
    bool fixed_ready_ = false;
};
using BlockDecoder = BlockDecoderT<unsigned char>;

// ---- one whole raw deflate stream in memory (a BGZF member's payload) -----------------------------------------------------
// `in` must be readable for PAD bytes past in + n.  false: not a valid stream of exactly out_len bytes.
inline bool inflate_whole(BlockDecoder &dec, const unsigned char *in, size_t n, unsigned char *out, size_t out_len) {
    BlockDecoder::Bits b;
    BlockDecoder::start(b, in);
    unsigned char *o = out, *const end = out + out_len;
    // (the decoder wants SLACK bytes behind the write position: the caller's buffer has them, see Input::fill_bgzf)
    for (bool final = false; !final;) {
        const Status st = dec.block(b, in + n, out, o, end + BlockDecoder::SLACK, final);
        if (st != OK) return false;
        if (o > end) return false;
    }
    return o == end;
}

// ---- a gzip member's header (RFC 1952 2.3) at p: OK and its length in `at`, NEED_INPUT when n bytes do not hold all of it, BAD
// when it is none (bytes behind the last member that are no member -- zero padding, garbage -- are an error, as they are to the
// reference's reader, flate2's MultiGzDecoder, and to the zlib path of this tool)
inline Status gzip_header(const unsigned char *p, size_t n, size_t &at) {
    if ((n > 0 && p[0] != 0x1F) || (n > 1 && p[1] != 0x8B)) return BAD;
    if (n < 10) return NEED_INPUT;
    if (p[2] != 8 || (p[3] & 0xE0)) return BAD;
    const unsigned flg = p[3];
    at = 10;
    if (flg & 4) {  // FEXTRA
        if (n < at + 2) return NEED_INPUT;
        at += 2 + (p[at] | (size_t)p[at + 1] << 8);
        if (n < at) return NEED_INPUT;
    }
    for (unsigned bit = 8; bit <= 16; bit <<= 1)  // FNAME, FCOMMENT: zero-terminated
        if (flg & bit) {
            const void *z = std::memchr(p + at, 0, n - at);
            if (!z) return NEED_INPUT;
            at = (size_t)((const unsigned char *)z - p) + 1;
        }
    if (flg & 2) {  // FHCRC: the low half of the CRC-32 of the header before it (checked, as zlib's inflate checks it)
        if (n < at + 2) return NEED_INPUT;
        const uint32_t want = p[at] | (uint32_t)p[at + 1] << 8;
        if ((crc32_fast(0, p, at) & 0xFFFF) != want) return BAD;
        at += 2;
    }
    return OK;
}

// ---- gzip members from a stream of bytes ------------------------------------------------------------------------------------
class GzReader {
  public:
    using Source = size_t (*)(void *ctx, unsigned char *dst, size_t cap);  // 0 = end of input
    GzReader(Source src, void *ctx) : src_(src), ctx_(ctx), ibuf_(IN_CAP + PAD), obuf_(WINDOW + OUT_CAP + SLACK) {
        o_begin_ = o_read_ = o_write_ = obuf_.data() + WINDOW;
    }
    // decompressed bytes; 0 = end of input.  error() is set on a malformed or truncated stream (and 0 is returned).
    size_t read(char *dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (o_read_ == o_write_ && !produce()) break;
            const size_t take = std::min<size_t>(n - got, (size_t)(o_write_ - o_read_));
            std::memcpy(dst + got, o_read_, take);
            o_read_ += take;
            got += take;
        }
        return got;
    }
    const std::string &error() const { return err_; }

  private:
    static constexpr size_t IN_CAP = 4u << 20, OUT_CAP = 8u << 20, WINDOW = 32768, SLACK = 258 + 16 + 64;
    enum State { HEADER, BLOCKS, TRAILER, DONE };

    bool fail(const char *m) {
        err_ = m;
        state_ = DONE;
        return false;
    }
    // more input behind in_end_; what lies before `keep` is dropped.  Returns the distance everything moved by.
    size_t more_input(const unsigned char *keep) {
        unsigned char *base = ibuf_.data();
        const size_t k = (size_t)(keep - base), have = in_end_ - k;
        if (k) std::memmove(base, keep, have);
        in_end_ = have;
        if (in_end_ == ibuf_.size() - PAD) ibuf_.resize(ibuf_.size() * 2);  // (a block or header larger than the buffer)
        base = ibuf_.data();
        while (!eof_ && in_end_ < ibuf_.size() - PAD) {
            const size_t r = src_(ctx_, base + in_end_, ibuf_.size() - PAD - in_end_);
            if (r == 0) eof_ = true;
            else in_end_ += r;
            // (filled to the brim: a block that meets the end of the buffer is decoded again from its first bit, so the fewer
            // ends the better; a source that delivers less than a megabyte at a time -- a pipe -- is not waited for beyond that)
            if (in_end_ >= (1u << 20) && r < (1u << 16)) break;
        }
        std::memset(base + in_end_, 0, PAD);
        return k;
    }
    // one step: a header, as many blocks as fit, or a trailer.  false: nothing more will come.
    bool produce() {
        for (;;) {
            unsigned char *base = ibuf_.data();
            if (state_ == DONE) return false;
            if (state_ == HEADER) {
                const unsigned char *p = base + in_pos_, *e = base + in_end_;
                if (p == e && eof_) return state_ = DONE, false;
                size_t at = 0;
                const Status hs = gzip_header(p, (size_t)(e - p), at);
                if (hs == BAD) return fail("invalid gzip stream");
                const bool ok = hs == OK;
                if (!ok) {
                    if (eof_) return fail("truncated gzip stream");
                    in_pos_ -= more_input(base + in_pos_);
                    continue;
                }
                in_pos_ += at;
                BlockDecoder::start(bits_, ibuf_.data() + in_pos_);
                crc_ = 0;
                isize_ = 0;
                // a member's matches reach back into its own output only
                if (o_read_ == o_write_) o_begin_ = o_read_ = o_write_ = obuf_.data() + WINDOW;
                else o_begin_ = o_write_;
                state_ = BLOCKS;
                continue;
            }
            if (state_ == BLOCKS) {
                bool produced = false;
                for (;;) {
                    bool final = false;
                    unsigned char *const before = o_write_;
                    const Status st = dec_.block(bits_, ibuf_.data() + in_end_, o_begin_, o_write_, obuf_.data() + obuf_.size() - 64, final);
                    if (st == OK) {
                        crc_ = crc32_fast(crc_, before, (size_t)(o_write_ - before));
                        isize_ += (uint64_t)(o_write_ - before);
                        produced = produced || o_write_ != before;
                        if (final) {
                            state_ = TRAILER;
                            break;
                        }
                        if ((size_t)(o_write_ - o_read_) >= (4u << 20)) break;  // enough to hand out
                        continue;
                    }
                    if (st == BAD) return fail("invalid gzip stream");
                    if (st == NEED_INPUT) {
                        if (eof_) return fail("truncated gzip stream");
                        // keep eight bytes in front of the reader's position: the bits it holds came from them
                        const unsigned char *keep = bits_.in - 8 >= ibuf_.data() ? bits_.in - 8 : ibuf_.data();
                        const unsigned char *old_base = ibuf_.data();
                        const size_t moved = more_input(keep);
                        bits_.in = ibuf_.data() + ((bits_.in - old_base) - moved);
                        continue;
                    }
                    // NEED_ROOM: hand out what there is; with nothing pending, slide the window to the front (or grow)
                    if (o_read_ != o_write_) break;
                    const size_t hist = std::min<size_t>((size_t)(o_write_ - o_begin_), WINDOW);
                    if (o_write_ - hist == obuf_.data() + (WINDOW - hist)) {  // already at the front: one block larger than the buffer
                        const size_t wofs = (size_t)(o_write_ - obuf_.data()), bofs = (size_t)(o_begin_ - obuf_.data());
                        obuf_.resize(obuf_.size() * 2);
                        o_write_ = o_read_ = obuf_.data() + wofs;
                        o_begin_ = obuf_.data() + bofs;
                    } else {
                        std::memmove(obuf_.data() + (WINDOW - hist), o_write_ - hist, hist);
                        o_write_ = o_read_ = obuf_.data() + WINDOW;
                        o_begin_ = o_write_ - hist;
                    }
                }
                if (state_ == TRAILER) in_pos_ = (size_t)(BlockDecoder::byte_position(bits_) - ibuf_.data());
                if (o_read_ != o_write_) return true;
                if (produced) return true;
                continue;
            }
            // TRAILER
            {
                const unsigned char *p = base + in_pos_, *e = base + in_end_;
                if ((size_t)(e - p) < 8) {
                    if (eof_) return fail("truncated gzip stream");
                    in_pos_ -= more_input(base + in_pos_);
                    continue;
                }
                const uint32_t crc = p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
                const uint32_t len = p[4] | (uint32_t)p[5] << 8 | (uint32_t)p[6] << 16 | (uint32_t)p[7] << 24;
                if (crc != crc_ || len != (uint32_t)isize_) return fail("invalid gzip stream");
                in_pos_ += 8;
                state_ = HEADER;
                if (o_read_ != o_write_) return true;
            }
        }
    }

    Source src_;
    void *ctx_;
    std::vector<unsigned char> ibuf_, obuf_;
    size_t in_pos_ = 0, in_end_ = 0;
    bool eof_ = false;
    unsigned char *o_begin_, *o_read_, *o_write_;
    State state_ = HEADER;
    BlockDecoder dec_;
    BlockDecoder::Bits bits_{};
    uint32_t crc_ = 0;
    uint64_t isize_ = 0;
    std::string err_;
};

}  // namespace fastgz
#endif  // DCN_FAST_INFLATE_HPP
