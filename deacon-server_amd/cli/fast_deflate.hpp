// fast_deflate.hpp -- raw deflate blocks written faster than zlib's deflate() writes them at its low levels, for the members of
// the tool's .gz outputs.
//
// Why: a .gz output is written as BGZF members of <= 65,280 input bytes, each compressed on its own by one of the formatter
// threads (Output::compress_member).  zlib at the reference's default level (--compression-level 2, src/main.rs:72) packs FASTQ
// at 85-170 MB/s per thread on the box: a run that keeps half of a large input then waits for its compressor, not for the GPU
// (630 MB of kept records: 0.95 s against 0.41 s uncompressed).  Levels 1-3 go through this file; 4 and above stay on zlib, whose
// longer searches are what those levels ask for.
//
// What: one input of <= 65,535 bytes -> one final deflate block (RFC 1951).  A greedy parse with one hash probe per position (4-byte
// hashes, the whole input is inside the 32 KB window's reach or nearly), a match being taken only where it is cheaper than the
// literals it replaces are likely to be (short matches at long distances cost more bits than FASTQ's 2-bit bases); symbol counts;
// length-limited Huffman codes built the way zlib builds them (heap order, overflow moved down the tree); the block written with
// a 64-bit bit buffer.  A block that would be larger than its input is stored.  The decoder's checks are the tests: every
// member carries the CRC-32 of its INPUT (computed apart from this code), so a wrong bit here cannot pass a reader silently.
#ifndef DCN_FAST_DEFLATE_HPP
#define DCN_FAST_DEFLATE_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>

namespace fastgz {

class FastDeflate {
  public:
    static constexpr size_t MAX_IN = 65535;
    static size_t bound(size_t n) { return n + 5 + 16; }  // (a stored block, and the bit writer's slack)

    // in[0, n), n <= MAX_IN -> one final raw-deflate block at out (bound(n) bytes of room); returns its length
    size_t compress(const unsigned char *in, size_t n, unsigned char *out) {
        if (n == 0) {  // the empty fixed-Huffman block zlib writes: BGZF's end-of-file member is recognised by its exact 28 bytes
            out[0] = 3, out[1] = 0;
            return 2;
        }
        n_tok_ = 0;
        std::memset(lfreq_, 0, sizeof lfreq_);
        std::memset(dfreq_, 0, sizeof dfreq_);
        uint64_t extra_bits = 0;
        parse(in, n, extra_bits);
        lfreq_[256] = 1;
        // at least two distance codes, as zlib has it: one bit is sent even where one code would do, and no decoder minds
        if (dfreq_[0] == 0) dfreq_[0] = 1;
        int used = 0;
        for (int i = 0; i < 30; ++i) used += dfreq_[i] != 0;
        if (used < 2) dfreq_[dfreq_[1] ? 2 : 1] = 1;
        uint8_t llen[288] = {0}, dlen[32] = {0};
        build_lengths(lfreq_, 286, 15, llen);
        build_lengths(dfreq_, 30, 15, dlen);
        int hlit = 286, hdist = 30;
        while (hlit > 257 && llen[hlit - 1] == 0) --hlit;
        while (hdist > 1 && dlen[hdist - 1] == 0) --hdist;
        // the code lengths, run-length coded with the code-length alphabet (16: repeat previous 3-6, 17: zeros 3-10, 18: zeros 11-138)
        uint8_t all[286 + 30];
        std::memcpy(all, llen, (size_t)hlit);
        std::memcpy(all + hlit, dlen, (size_t)hdist);
        uint16_t rle[286 + 30];  // symbol | extra << 8
        int n_rle = 0;
        uint32_t cfreq[19] = {0};
        for (int i = 0; i < hlit + hdist;) {
            const int v = all[i];
            int run = 1;
            while (i + run < hlit + hdist && all[i + run] == v) ++run;
            i += run;
            if (v == 0) {
                while (run >= 11) {
                    const int r = std::min(run, 138);
                    rle[n_rle++] = (uint16_t)(18 | (r - 11) << 8), ++cfreq[18], run -= r;
                }
                if (run >= 3) rle[n_rle++] = (uint16_t)(17 | (run - 3) << 8), ++cfreq[17], run = 0;
            } else {
                rle[n_rle++] = (uint16_t)v, ++cfreq[v], --run;
                while (run >= 3) {
                    const int r = std::min(run, 6);
                    rle[n_rle++] = (uint16_t)(16 | (r - 3) << 8), ++cfreq[16], run -= r;
                }
            }
            while (run-- > 0) rle[n_rle++] = (uint16_t)v, ++cfreq[v];
        }
        {  // (a code-length code of one symbol would be incomplete, which no decoder accepts for this alphabet)
            int used_c = 0;
            for (int i = 0; i < 19; ++i) used_c += cfreq[i] != 0;
            if (used_c < 2) cfreq[cfreq[0] ? 1 : 0] = 1;
        }
        uint8_t clen[19] = {0};
        build_lengths(cfreq, 19, 7, clen);
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = 19;
        while (hclen > 4 && clen[order[hclen - 1]] == 0) --hclen;
        // what the block costs, against storing it
        uint64_t bits = 3 + 14 + 3 * (uint64_t)hclen + extra_bits;
        for (int i = 0; i < 19; ++i) bits += (uint64_t)cfreq[i] * clen[i];
        bits += 2 * (uint64_t)cfreq[16] + 3 * (uint64_t)cfreq[17] + 7 * (uint64_t)cfreq[18];
        for (int i = 0; i < hlit; ++i) bits += (uint64_t)lfreq_[i] * llen[i];
        for (int i = 0; i < hdist; ++i) bits += (uint64_t)dfreq_[i] * dlen[i];
        if ((bits + 7) / 8 >= n + 5) return stored(in, n, out);
        uint16_t lcode[288], dcode[32], ccode[19];
        make_codes(llen, 286, lcode);
        make_codes(dlen, 30, dcode);
        make_codes(clen, 19, ccode);
        Writer w{out, 0, 0};
        w.put(1 | 2 << 1, 3);  // BFINAL, dynamic
        w.put((uint64_t)(hlit - 257) | (uint64_t)(hdist - 1) << 5 | (uint64_t)(hclen - 4) << 10, 14);
        for (int i = 0; i < hclen; ++i) {
            w.put(clen[order[i]], 3);
            if (w.cnt > 48) w.flush();
        }
        for (int i = 0; i < n_rle; ++i) {
            const int s = rle[i] & 0xFF, e = rle[i] >> 8;
            w.flush();
            w.put(ccode[s], clen[s]);
            if (s == 16) w.put((uint64_t)e, 2);
            else if (s == 17) w.put((uint64_t)e, 3);
            else if (s == 18) w.put((uint64_t)e, 7);
        }
        // literal: code | length << 16 in one word
        uint32_t lit[256];
        for (int i = 0; i < 256; ++i) lit[i] = (uint32_t)lcode[i] | (uint32_t)llen[i] << 16;
        for (size_t t = 0; t < n_tok_; ++t) {
            const uint32_t k = tok_[t];
            if (!(k & 0x80000000u)) {
                if (w.cnt > 48) w.flush();  // (48 + 15 <= 64)
                const uint32_t e = lit[k];
                w.put(e & 0xFFFF, e >> 16);
                continue;
            }
            const unsigned len3 = (k >> 16) & 0xFF, dist1 = k & 0xFFFF;
            const unsigned ls = len_sym_[len3], ds = dist_sym(dist1);
            // code + extra bits of the length (<= 20), then of the distance (<= 28), behind at most 7 pending bits
            w.flush();
            w.put((uint64_t)lcode[257 + ls] | (uint64_t)(len3 - len_base_[ls]) << llen[257 + ls], llen[257 + ls] + len_extra_[ls]);
            w.put((uint64_t)dcode[ds] | (uint64_t)(dist1 - dist_base_[ds]) << dlen[ds], dlen[ds] + dist_extra_[ds]);
        }
        w.flush();
        w.put(lcode[256], llen[256]);
        return w.finish(out);
    }

    FastDeflate() {
        // length - 3 -> length symbol - 257
        for (unsigned s = 0; s < 29; ++s)
            for (unsigned l = len_base_[s]; l < (s == 28 ? 256u : (unsigned)len_base_[s + 1]); ++l) len_sym_[l] = (uint8_t)s;
        len_sym_[255] = 28;  // length 258 has a code of its own
        for (unsigned s = 0; s < 30; ++s) {
            const unsigned hi = s == 29 ? 32768u : dist_base_[s + 1];
            for (unsigned d = dist_base_[s]; d < hi && d < 512; ++d) dist_sym_lo_[d] = (uint8_t)s;
            if (hi > 512)
                for (unsigned d = std::max<unsigned>(dist_base_[s], 512) >> 7; d <= (hi - 1) >> 7; ++d) dist_sym_hi_[d] = (uint8_t)s;
        }
    }

  private:
    static constexpr int HASH_BITS = 15;
    struct Writer {
        unsigned char *p;
        uint64_t buf;
        unsigned cnt;
        inline void put(uint64_t bits, unsigned n) {  // cnt + n <= 64 is the caller's business: see flush()
            buf |= bits << cnt;
            cnt += n;
        }
        inline void flush() {  // whole bytes out; at most 7 bits stay
            std::memcpy(p, &buf, 8);
            const unsigned bytes = cnt >> 3;
            p += bytes;
            buf = bytes == 8 ? 0 : buf >> (8 * bytes);
            cnt &= 7;
        }
        size_t finish(unsigned char *start) {
            flush();
            if (cnt) *p++ = (unsigned char)buf;
            return (size_t)(p - start);
        }
    };
    size_t stored(const unsigned char *in, size_t n, unsigned char *out) {
        out[0] = 1;  // BFINAL, stored; the rest of the byte is padding
        out[1] = (unsigned char)(n & 0xFF), out[2] = (unsigned char)(n >> 8);
        out[3] = (unsigned char)(~n & 0xFF), out[4] = (unsigned char)((~n >> 8) & 0xFF);
        if (n) std::memcpy(out + 5, in, n);
        return n + 5;
    }
    static inline uint32_t load32(const unsigned char *p) {
        uint32_t v;
        std::memcpy(&v, p, 4);
        return v;
    }
    static inline uint64_t load64(const unsigned char *p) {
        uint64_t v;
        std::memcpy(&v, p, 8);
        return v;
    }
    inline unsigned dist_sym(unsigned dist1) const { return dist1 < 512 ? dist_sym_lo_[dist1] : dist_sym_hi_[dist1 >> 7]; }

    // greedy parse, one probe per position
    void parse(const unsigned char *in, size_t n, uint64_t &extra_bits) {
        std::memset(head_, 0, sizeof head_);
        size_t i = 0, lit_run = 0;
        const size_t last_hash = n >= 4 ? n - 4 : 0;
        while (n >= 4 && i <= last_hash) {
            const uint32_t v = load32(in + i);
            const uint32_t h = (v * 0x9E3779B1u) >> (32 - HASH_BITS);
            const unsigned cand1 = head_[h];
            head_[h] = (uint16_t)(i + 1);
            if (cand1 && load32(in + cand1 - 1) == v) {
                const size_t c = cand1 - 1, dist = i - c;
                const size_t max_len = std::min<size_t>(258, n - i);
                size_t len = 4;
                while (len + 8 <= max_len) {
                    const uint64_t x = load64(in + i + len) ^ load64(in + c + len);
                    if (x) {
                        len += (size_t)(__builtin_ctzll(x) >> 3);
                        goto done;
                    }
                    len += 8;
                }
                while (len < max_len && in[i + len] == in[c + len]) ++len;
            done:
                // a short match far away costs more bits than the literals it stands for where literals are cheap (bases: ~2 bits
                // each; a length / distance pair: 20-30 bits)
                if (dist <= 32768 && (len >= MIN_FAR || (len >= 5 && dist <= 2048) || dist <= 256)) {
                    const unsigned len3 = (unsigned)len - 3, dist1 = (unsigned)dist - 1;
                    tok_[n_tok_++] = 0x80000000u | len3 << 16 | dist1;
                    const unsigned ls = len_sym_[len3], ds = dist_sym(dist1);
                    ++lfreq_[257 + ls];
                    ++dfreq_[ds];
                    extra_bits += len_extra_[ls] + dist_extra_[ds];
                    // positions inside the match: the last few are worth finding again (the next record's id, the next line)
                    const size_t end = i + len;
                    for (size_t j = end > 3 ? std::max(i + 1, end - 3) : i + 1; j < end && j <= last_hash; ++j)
                        head_[(load32(in + j) * 0x9E3779B1u) >> (32 - HASH_BITS)] = (uint16_t)(j + 1);
                    i = end;
                    lit_run = 0;
                    continue;
                }
            }
            // no match: this byte is a literal -- and so are the next few where there has been no match for a while (random
            // quality strings, data that is compressed already: one probe per 2, 3, ... positions)
            const size_t step = 1 + (lit_run >> ACCEL_SHIFT);
            lit_run += step;
            for (size_t e = std::min(i + step, n); i < e; ++i) tok_[n_tok_++] = in[i], ++lfreq_[in[i]];
        }
        for (; i < n; ++i) tok_[n_tok_++] = in[i], ++lfreq_[in[i]];
    }

    // Huffman code lengths for freq[0, n), none longer than max_bits: the tree by repeated merging of the two rarest (an
    // array-based heap would do; n <= 286, so two sorted queues), then zlib's repair where the tree is deeper than allowed
    // (trees.c gen_bitlen: move overflowing leaves up, pay with a leaf from the deepest level that has one to give)
    static void build_lengths(const uint32_t *freq, int n, int max_bits, uint8_t *lens) {
        struct Node {
            uint32_t f;
            int16_t sym, left, right;
        };
        Node nodes[2 * 288];
        int16_t leaves[288];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            lens[i] = 0;
            if (freq[i]) leaves[m++] = (int16_t)i;
        }
        if (m == 0) return;
        if (m == 1) {
            lens[leaves[0]] = 1;
            return;
        }
        std::sort(leaves, leaves + m, [&](int16_t a, int16_t b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
        for (int i = 0; i < m; ++i) nodes[i] = {freq[leaves[i]], leaves[i], -1, -1};
        int q1 = 0, q2 = m, end = m;  // leaves [q1, m), internal nodes [q2, end) -- both in rising order of f
        auto take = [&]() {
            if (q1 < m && (q2 >= end || nodes[q1].f <= nodes[q2].f)) return q1++;
            return q2++;
        };
        while ((m - q1) + (end - q2) > 1) {
            const int a = take(), b = take();
            nodes[end] = {nodes[a].f + nodes[b].f, -1, (int16_t)a, (int16_t)b};
            ++end;
        }
        // depths, root = end - 1
        uint8_t depth[2 * 288];
        depth[end - 1] = 0;
        int bl_count[64] = {0};
        for (int i = end - 1; i >= m; --i) {
            depth[nodes[i].left] = depth[nodes[i].right] = (uint8_t)(depth[i] + 1);
        }
        int overflow = 0;
        for (int i = 0; i < m; ++i) {
            int d = depth[i];
            if (d > max_bits) d = max_bits, ++overflow;
            ++bl_count[d];
        }
        if (overflow) {
            do {
                int bits = max_bits - 1;
                while (bl_count[bits] == 0) --bits;
                --bl_count[bits];
                bl_count[bits + 1] += 2;
                --bl_count[max_bits];
                overflow -= 2;
            } while (overflow > 0);
        }
        // hand the lengths out again: the longest to the rarest (leaves are in rising order of frequency)
        int at = 0;
        for (int bits = max_bits; bits >= 1; --bits)
            for (int c = bl_count[bits]; c > 0; --c) lens[nodes[at++].sym] = (uint8_t)bits;
    }
    // canonical codes, bit-reversed (deflate sends codes most significant bit first, everything else least significant first)
    static void make_codes(const uint8_t *lens, int n, uint16_t *codes) {
        int bl_count[16] = {0};
        for (int i = 0; i < n; ++i) ++bl_count[lens[i]];
        bl_count[0] = 0;
        unsigned next[16], code = 0;
        for (int b = 1; b <= 15; ++b) {
            code = (code + (unsigned)bl_count[b - 1]) << 1;
            next[b] = code;
        }
        for (int i = 0; i < n; ++i) {
            const int l = lens[i];
            if (!l) {
                codes[i] = 0;
                continue;
            }
            unsigned c = next[l]++, r = 0;
            for (int q = 0; q < l; ++q) r |= ((c >> q) & 1u) << (l - 1 - q);
            codes[i] = (uint16_t)r;
        }
    }

    static constexpr size_t MIN_FAR = 8;
    static constexpr unsigned ACCEL_SHIFT = 6;
    static constexpr uint16_t len_base_[29] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 255};  // length - 3
    static constexpr uint8_t len_extra_[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static constexpr uint16_t dist_base_[30] = {0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576};  // distance - 1
    static constexpr uint8_t dist_extra_[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    uint8_t len_sym_[256], dist_sym_lo_[512], dist_sym_hi_[256];
    uint16_t head_[1 << HASH_BITS];
    uint32_t tok_[MAX_IN + 8];
    size_t n_tok_ = 0;
    uint32_t lfreq_[288], dfreq_[32];
};

}  // namespace fastgz
#endif  // DCN_FAST_DEFLATE_HPP
