// index_file.cpp -- reader for the reference's index file (host side of A9).
//
// Layout written by write_minimizers (src/index.rs:130-164) with bincode 2 `config::standard()`:
//   3 raw bytes  IndexHeader { format_version = 2, kmer_length, window_size }   (src/index.rs:17-31)
//   varint       count
//   count x      varint u64 minimizer hash
// varint: b < 251 -> the value; 0xFB + u16 LE; 0xFC + u32 LE; 0xFD + u64 LE.
// Loading mirrors load_minimizer_hashes (src/index.rs:80-107) minus the host hash set: the keys go straight
// to the device table builder, which merges duplicates.
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

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

// write_minimizers (src/index.rs:130-164): header, count, then every hash as a bincode varint.
// Blocks of 8 Mi hashes are encoded by a few threads (sizes first, then each thread writes its part of the block at
// its offset) while the previous block is being written out.
namespace {

inline size_t varint_size(uint64_t v) { return v < 251 ? 1 : v <= 0xFFFFull ? 3 : v <= 0xFFFFFFFFull ? 5 : 9; }

inline uint8_t *put_varint(uint8_t *p, uint64_t v) {
    if (v < 251) {
        *p++ = (uint8_t)v;
        return p;
    }
    size_t nb;
    if (v <= 0xFFFFull) {
        *p++ = 0xFB;
        nb = 2;
    } else if (v <= 0xFFFFFFFFull) {
        *p++ = 0xFC;
        nb = 4;
    } else {
        *p++ = 0xFD;
        nb = 8;
    }
    memcpy(p, &v, nb); // little-endian host
    return p + nb;
}

// encodes keys[0..n) into out (resized to the exact byte count) with `threads` workers
void encode_block(const uint64_t *keys, size_t n, unsigned threads, std::vector<uint8_t> &out) {
    threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / 65536 + 1));
    std::vector<size_t> part(threads + 1, 0);
    auto range = [&](unsigned t, size_t &lo, size_t &hi) {
        lo = n * t / threads;
        hi = n * (t + 1) / threads;
    };
    auto run = [&](auto &&fn) {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; ++t) pool.emplace_back(fn, t);
        fn(0u);
        for (auto &th : pool) th.join();
    };
    run([&](unsigned t) {
        size_t lo, hi, bytes = 0;
        range(t, lo, hi);
        for (size_t i = lo; i < hi; ++i) bytes += varint_size(keys[i]);
        part[t + 1] = bytes;
    });
    for (unsigned t = 0; t < threads; ++t) part[t + 1] += part[t];
    out.resize(part[threads]);
    run([&](unsigned t) {
        size_t lo, hi;
        range(t, lo, hi);
        uint8_t *p = out.data() + part[t];
        for (size_t i = lo; i < hi; ++i) p = put_varint(p, keys[i]);
    });
}

} // namespace

int dcn_write_index_file(const char *path, uint8_t k, uint8_t w, const uint64_t *keys, uint64_t n) {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned threads = std::min(8u, hw ? hw : 1u);
    if (const char *e = getenv("DCN_HOST_THREADS")) threads = (unsigned)std::max(1, atoi(e));
    uint8_t head[3 + 9] = {2, k, w};
    const size_t head_len = (size_t)(put_varint(head + 3, n) - head);
    // A regular file is written by all threads at once: the size of every slice's encoding is known after one pass over the
    // keys, so each thread encodes its slice block by block and writes it at its own offset (one writer behind one encoder
    // moved 1 GB/s into tmpfs -- 0.45 s of a 1.05 s index build of a 400 Mbp genome; the page cache takes several writers).
    // Anything else (a pipe, /dev/stdout) takes the blocks in order from one writer, as before.
    int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return dcn_fail(DCN_ERR_IO, std::string("Failed to create output file ") + path);
    struct stat st;
    const bool regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && !getenv("DCN_INDEX_WRITE_SERIAL");
    if (regular && n > 0) {
        threads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads, n / 65536 + 1));
        std::vector<uint64_t> part(threads + 1, 0);
        auto slice = [&](unsigned t, uint64_t &lo, uint64_t &hi) {
            lo = n * t / threads;
            hi = n * (t + 1) / threads;
        };
        auto run = [&](auto &&fn) {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < threads; ++t) pool.emplace_back(fn, t);
            fn(0u);
            for (auto &th : pool) th.join();
        };
        run([&](unsigned t) {
            uint64_t lo, hi, bytes = 0;
            slice(t, lo, hi);
            for (uint64_t i = lo; i < hi; ++i) bytes += varint_size(keys[i]);
            part[t + 1] = bytes;
        });
        part[0] = head_len;
        for (unsigned t = 0; t < threads; ++t) part[t + 1] += part[t];
        std::atomic<bool> failed{false};
        auto write_all = [&](const uint8_t *p, size_t len, uint64_t off) {
            while (len) {
                const ssize_t w_ = ::pwrite(fd, p, len, (off_t)off);
                if (w_ < 0) {
                    if (errno == EINTR) continue;
                    failed = true;
                    return;
                }
                p += w_;
                len -= (size_t)w_;
                off += (uint64_t)w_;
            }
        };
        write_all(head, head_len, 0);
        run([&](unsigned t) {
            uint64_t lo, hi;
            slice(t, lo, hi);
            std::vector<uint8_t> buf;
            uint64_t off = part[t];
            const uint64_t STEP = 1u << 20;  // keys per block
            for (uint64_t i = lo; i < hi && !failed; i += STEP) {
                const uint64_t m = std::min<uint64_t>(STEP, hi - i);
                buf.resize(m * 9);
                uint8_t *q = buf.data();
                for (uint64_t j = 0; j < m; ++j) q = put_varint(q, keys[i + j]);
                write_all(buf.data(), (size_t)(q - buf.data()), off);
                off += (uint64_t)(q - buf.data());
            }
        });
        int rc = failed ? dcn_fail(DCN_ERR_IO, "short write") : DCN_OK;
        if (::close(fd) != 0 && rc == DCN_OK) rc = dcn_fail(DCN_ERR_IO, "close failed");
        return rc;
    }
    FILE *f = fdopen(fd, "wb");
    if (!f) {
        ::close(fd);
        return dcn_fail(DCN_ERR_IO, std::string("Failed to create output file ") + path);
    }
    setvbuf(f, nullptr, _IONBF, 0); // blocks are tens of megabytes: no second copy through stdio
    int rc = fwrite(head, 1, head_len, f) == head_len ? DCN_OK : dcn_fail(DCN_ERR_IO, "short write");
    const uint64_t BLOCK = 8ull << 20;
    std::vector<uint8_t> buf[2];
    std::thread writer;
    bool write_failed = false;
    int which = 0;
    for (uint64_t off = 0; off < n && rc == DCN_OK; off += BLOCK, which ^= 1) {
        encode_block(keys + off, (size_t)std::min<uint64_t>(BLOCK, n - off), threads, buf[which]);
        if (writer.joinable()) writer.join(); // the other buffer has been written: it is free for the next block
        if (write_failed) break;
        std::vector<uint8_t> *b = &buf[which];
        writer = std::thread([f, b, &write_failed] {
            if (fwrite(b->data(), 1, b->size(), f) != b->size()) write_failed = true;
        });
    }
    if (writer.joinable()) writer.join();
    if (write_failed && rc == DCN_OK) rc = dcn_fail(DCN_ERR_IO, "short write");
    if (fclose(f) != 0 && rc == DCN_OK) rc = dcn_fail(DCN_ERR_IO, "close failed");
    return rc;
}
