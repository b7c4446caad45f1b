// index_file.cpp -- reader for the reference's index file (host side of A9).
//
// Layout written by write_minimizers (src/index.rs:130-164) with bincode 2 `config::standard()`:
//   3 raw bytes  IndexHeader { format_version = 2, kmer_length, window_size }   (src/index.rs:17-31)
//   varint       count
//   count x      varint u64 minimizer hash
// varint: b < 251 -> the value; 0xFB + u16 LE; 0xFC + u32 LE; 0xFD + u64 LE.
// Loading mirrors load_minimizer_hashes (src/index.rs:80-107) minus the host hash set: the keys go straight
// to the device table builder, which merges duplicates.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dcn_internal.h"

namespace {

class ByteReader {
  public:
    explicit ByteReader(FILE *f) : f_(f), buf_(1 << 22) {}
    // makes at least n bytes (n <= 16) available at cur(); false on EOF
    bool ensure(size_t n) {
        if (end_ - pos_ >= n) return true;
        size_t rem = end_ - pos_;
        memmove(buf_.data(), buf_.data() + pos_, rem);
        pos_ = 0;
        end_ = rem;
        while (end_ < n) {
            size_t got = fread(buf_.data() + end_, 1, buf_.size() - end_, f_);
            if (got == 0) return false;
            end_ += got;
        }
        return true;
    }
    const uint8_t *cur() const { return buf_.data() + pos_; }
    void advance(size_t n) { pos_ += n; }

  private:
    FILE *f_;
    std::vector<uint8_t> buf_;
    size_t pos_ = 0, end_ = 0;
};

bool read_varint(ByteReader &r, uint64_t *out) {
    if (!r.ensure(1)) return false;
    uint8_t b = *r.cur();
    r.advance(1);
    if (b < 251) {
        *out = b;
        return true;
    }
    size_t n = b == 0xFB ? 2 : b == 0xFC ? 4 : b == 0xFD ? 8 : 0;
    if (n == 0) return false; // 0xFE (u128) / 0xFF are not valid for a u64
    if (!r.ensure(n)) return false;
    uint64_t v = 0;
    memcpy(&v, r.cur(), n); // little-endian host
    r.advance(n);
    *out = v;
    return true;
}

} // namespace

int dcn_read_index_file(const char *path, uint8_t *k, uint8_t *w, std::vector<uint64_t> *keys) {
    FILE *f = fopen(path, "rb");
    if (!f) return dcn_fail(DCN_ERR_IO, std::string("Failed to open index file ") + path);
    ByteReader r(f);
    int rc = DCN_OK;
    uint64_t count = 0;
    if (!r.ensure(3)) {
        rc = dcn_fail(DCN_ERR_FORMAT, "Failed to deserialise index header");
    } else {
        const uint8_t *h = r.cur();
        if (h[0] != 2) { // IndexHeader::validate, src/index.rs:34-43
            rc = dcn_fail(DCN_ERR_FORMAT, "Unsupported index format version: " + std::to_string((int)h[0]));
        } else {
            *k = h[1];
            *w = h[2];
            r.advance(3);
            if (!read_varint(r, &count)) rc = dcn_fail(DCN_ERR_FORMAT, "Failed to deserialise minimizer count");
        }
    }
    if (rc == DCN_OK) {
        try {
            keys->resize(count);
        } catch (...) {
            rc = dcn_fail(DCN_ERR_NOMEM, "index too large for host memory");
        }
    }
    if (rc == DCN_OK) {
        uint64_t *dst = keys->data();
        for (uint64_t i = 0; i < count; ++i)
            if (!read_varint(r, &dst[i])) {
                rc = dcn_fail(DCN_ERR_FORMAT, "Failed to deserialise minimizer hash");
                break;
            }
    }
    fclose(f);
    return rc;
}

// write_minimizers (src/index.rs:130-164): header, count, then every hash as a bincode varint
int dcn_write_index_file(const char *path, uint8_t k, uint8_t w, const uint64_t *keys, uint64_t n) {
    FILE *f = fopen(path, "wb");
    if (!f) return dcn_fail(DCN_ERR_IO, std::string("Failed to create output file ") + path);
    std::vector<uint8_t> buf;
    buf.reserve(1 << 22);
    auto put_varint = [&](uint64_t v) {
        if (v < 251) {
            buf.push_back((uint8_t)v);
            return;
        }
        size_t nb;
        if (v <= 0xFFFFull) {
            buf.push_back(0xFB);
            nb = 2;
        } else if (v <= 0xFFFFFFFFull) {
            buf.push_back(0xFC);
            nb = 4;
        } else {
            buf.push_back(0xFD);
            nb = 8;
        }
        for (size_t i = 0; i < nb; ++i) buf.push_back((uint8_t)(v >> (8 * i)));
    };
    buf.push_back(2);
    buf.push_back(k);
    buf.push_back(w);
    put_varint(n);
    int rc = DCN_OK;
    for (uint64_t i = 0; i < n && rc == DCN_OK; ++i) {
        put_varint(keys[i]);
        if (buf.size() >= (1u << 22) - 16) {
            if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) rc = dcn_fail(DCN_ERR_IO, "short write");
            buf.clear();
        }
    }
    if (rc == DCN_OK && !buf.empty() && fwrite(buf.data(), 1, buf.size(), f) != buf.size())
        rc = dcn_fail(DCN_ERR_IO, "short write");
    if (fclose(f) != 0 && rc == DCN_OK) rc = dcn_fail(DCN_ERR_IO, "close failed");
    return rc;
}
