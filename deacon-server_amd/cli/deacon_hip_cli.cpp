// deacon-hip -- command-line driver around libdeacon_hip.so with the reference's `deacon` surface for the path this
// repository accelerates:
//
//   deacon-hip index build <fastx> [-k 31] [-w 15] [-o out.idx] [-c capacity_millions] [-e entropy] [-q]
//   deacon-hip index info  <index>
//   deacon-hip index union <index>... [-o out.idx]
//   deacon-hip index diff  <first.idx> <second.idx | fastx> [-k K -w W] [-o out.idx]
//   deacon-hip filter <index> [input|-] [input2|-] [-o out] [-O out2] [-a 2] [-r 0.01] [-p 0] [-d] [-R]
//                     [-s summary.json] [-t threads] [--compression-level 2] [--debug] [-q]
//
// Flags, defaults, stderr messages and the JSON summary follow src/main.rs:24-234, src/local_filter.rs:575-824 and
// src/filter_common.rs:11-38 of the reference.  The per-record loop of local_filter.rs (paraseq workers calling
// should_keep_sequence / should_keep_pair) is replaced by batches through dcn_filter_batch.  Pipeline, every stage
// in input order: parse (mmap + worker pool for a plain file, one streaming reader for stdin / gzip / pairs; it starts
// before the GPU is initialised) -> GPU stage (its own thread and context) -> format kept records on a pool
// (format_record_to_buffer, src/local_filter.rs:60-92) -> one writer thread.  Output keeps the input order (the reference's depends on worker scheduling).
// The server/client commands live in deacon-server_amd/server.py / client.py.
// Input/output compression: gzip via zlib; zstd and xz through the installed runtime libraries (codecs.hpp).
#include <fcntl.h>
#include <spawn.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <sched.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <sys/vfs.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "codecs.hpp"
#include "fast_deflate.hpp"
#include "fast_inflate.hpp"
#include "parallel_gzip.hpp"
#include "deacon_hip.hpp"

namespace {

const char *VERSION = "0.4.0";

std::atomic<int> g_sparse_out_fds[2] = {{-1}, {-1}};  // output files still sized to their reservation (MappedOutput): cut on failure

void cut_sparse_outputs() {  // async-signal-safe
    for (auto &g : g_sparse_out_fds) {
        int fd = g.exchange(-1);
        if (fd >= 0 && ftruncate(fd, 0) != 0) {
        }
    }
}

[[noreturn]] void die(const std::string &msg) {
    std::fprintf(stderr, "Error: %s\n", msg.c_str());
    std::fflush(nullptr);
    cut_sparse_outputs();
    _exit(1);  // callable from any pipeline thread while the others are still running
}

bool ends_with(const std::string &s, const char *suf) {
    size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

size_t usable_cpus();  // (below) the CPUs this process may really use

// ---- input: plain, gzip, zstd or xz by content, file or stdin (niffler's role in src/local_filter.rs:41-55) ---------
// A gzip input made of BGZF members (bgzip / htslib, and what several sequencer pipelines write: every member <= 64 KB and
// carrying its own length) is inflated on several threads, 16 MB of members at a time -- 1.0 GB/s on 4 threads, 1.5 on 8,
// against 0.35 GB/s for zlib on one stream, which is what bounds every other gzip input (and the reference's reader).
class Input {
  public:
    explicit Input(const std::string &path) : raw_(1 << 20) {
        fd_ = path == "-" ? 0 : ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) die("Failed to open file " + path);
        refill();
        const unsigned char *m = (const unsigned char *)raw_.data();
        const size_t n = end_;
        std::string why;
        if (codecs::is_gzip_magic(m, n)) {
            kind_ = GZIP;
            std::memset(&zs_, 0, sizeof zs_);
            if (inflateInit2(&zs_, 15 + 32) != Z_OK) die("zlib initialisation failed");
            // blocked gzip (BGZF: what bgzip, htslib and several sequencer pipelines write): every member says how long it
            // is, so members are found without inflating them and inflated side by side (see fill_bgzf)
            bgzf_ = bgzf_block_size(m, n) != 0 && !std::getenv("DCN_CLI_NO_BGZF");
            // our own inflate (fast_inflate.hpp) unless zlib's is asked for: the same bytes out, the same errors
            fast_ = !std::getenv("DCN_CLI_ZLIB_INFLATE");
            if (fast_) {
                // one stream on several threads (parallel_gzip.hpp) where there are threads to be had
                size_t workers = std::min<size_t>(std::max<size_t>(1, usable_cpus() / 2), 16);  // (8 on a 16-CPU share; quality strings with many values want the 16)
                if (const char *e = std::getenv("DCN_CLI_GZ_THREADS")) workers = (size_t)std::max(1, std::atoi(e));
                size_t chunk = 2u << 20;  // (test hook: small chunks put every path of the reader to work on small files)
                if (const char *e = std::getenv("DCN_CLI_GZ_CHUNK")) chunk = (size_t)std::max(4096, std::atoi(e));
                if (workers > 1 && !std::getenv("DCN_CLI_NO_PARALLEL_GZ")) pgz_.reset(new fastgz::ParallelGzReader(&Input::gz_source, this, (unsigned)workers, chunk));
                else gz_.reset(new fastgz::GzReader(&Input::gz_source, this));
            }
        } else if (codecs::is_zstd_magic(m, n)) {
            kind_ = ZSTD;
            zstd_ = codecs::Zstd::get(&why);
            if (!zstd_) die("zstd input " + path + ": " + why);
            zds_ = zstd_->createDStream();
            if (!zds_ || zstd_->isError(zstd_->initDStream(zds_))) die("zstd initialisation failed");
        } else if (codecs::is_bz2_magic(m, n)) {
            kind_ = BZ2;
            bz2_ = codecs::Bz2::get(&why);
            if (!bz2_) die("bzip2 input " + path + ": " + why);
            std::memset(&bs_, 0, sizeof bs_);
            if (bz2_->decompress_init(&bs_, 0, 0) != codecs::BZ_OK) die("bzip2 initialisation failed");
        } else if (codecs::is_xz_magic(m, n)) {
            kind_ = XZ;
            lzma_ = codecs::Lzma::get(&why);
            if (!lzma_) die("xz input " + path + ": " + why);
            std::memset(&ls_, 0, sizeof ls_);
            if (lzma_->stream_decoder(&ls_, UINT64_MAX, codecs::LZMA_CONCATENATED) != codecs::LZMA_OK) die("xz initialisation failed");
        }
    }
    ~Input() {
        pgz_.reset();  // (its driver thread reads fd_: gone before the descriptor is)
        if (kind_ == GZIP) inflateEnd(&zs_);
        if (kind_ == ZSTD && zds_) zstd_->freeDStream(zds_);
        if (kind_ == XZ) lzma_->end(&ls_);
        if (kind_ == BZ2) bz2_->decompress_end(&bs_);
        if (fd_ > 0) ::close(fd_);
    }
    size_t read(char *dst, size_t n) {
        size_t got = 0;
        while (got < n && !done_) {
            // (a gzip stream's reader fetches its own input through gz_source, the parallel one on a thread of its own: from
            // its first call on, raw_ and the file belong to it)
            const bool own_input = kind_ == GZIP && fast_ && !bgzf_;
            if (!own_input && pos_ == end_ && !raw_eof_) refill();
            const size_t avail = own_input ? 0 : end_ - pos_;
            if (kind_ == PLAIN) {
                if (!avail) break;
                const size_t m = std::min(avail, n - got);
                std::memcpy(dst + got, raw_.data() + pos_, m);
                pos_ += m;
                got += m;
            } else if (kind_ == GZIP && bgzf_) {
                if (bz_pos_ == bz_out_.size() && !fill_bgzf()) continue;  // (nothing more in blocked form: the stream path takes over, or the input has ended)
                const size_t take = std::min(n - got, bz_out_.size() - bz_pos_);
                std::memcpy(dst + got, bz_out_.data() + bz_pos_, take);
                bz_pos_ += take;
                got += take;
            } else if (kind_ == GZIP && fast_) {
                // (either reader takes its input through gz_source: raw_'s rest first, then the file)
                const size_t r = pgz_ ? pgz_->read(dst + got, n - got) : gz_->read(dst + got, n - got);
                if (r == 0) {
                    const std::string &err = pgz_ ? pgz_->error() : gz_->error();
                    if (!err.empty()) die("read error: " + err);
                    done_ = true;
                }
                got += r;
            } else if (kind_ == GZIP) {
                if (!avail && raw_eof_) {
                    if (mid_stream_) die("read error: truncated gzip stream");
                    break;
                }
                zs_.next_in = (Bytef *)(raw_.data() + pos_);
                zs_.avail_in = (uInt)avail;
                zs_.next_out = (Bytef *)(dst + got);
                zs_.avail_out = (uInt)std::min<size_t>(n - got, 1u << 30);
                const uInt out0 = zs_.avail_out;
                int r = inflate(&zs_, Z_NO_FLUSH);
                if (r != Z_OK && r != Z_STREAM_END && r != Z_BUF_ERROR) die("read error: invalid gzip stream");
                pos_ += avail - zs_.avail_in;
                got += out0 - zs_.avail_out;
                mid_stream_ = r != Z_STREAM_END;
                if (r == Z_STREAM_END && inflateReset(&zs_) != Z_OK) die("zlib reset failed");  // next member (bgzip, cat a.gz b.gz)
            } else if (kind_ == ZSTD) {
                if (!avail && raw_eof_) {
                    if (mid_stream_) die("read error: truncated zstd stream");
                    break;
                }
                codecs::ZSTD_inBuffer in = {raw_.data() + pos_, avail, 0};
                codecs::ZSTD_outBuffer out = {dst + got, n - got, 0};
                const size_t r = zstd_->decompressStream(zds_, &out, &in);
                if (zstd_->isError(r)) die(std::string("read error: zstd: ") + zstd_->getErrorName(r));
                pos_ += in.pos;
                got += out.pos;
                mid_stream_ = r != 0;  // 0: a frame just ended
            } else if (kind_ == BZ2) {
                if (!avail && raw_eof_) {
                    if (mid_stream_) die("read error: truncated bzip2 stream");
                    break;
                }
                bs_.next_in = raw_.data() + pos_;
                bs_.avail_in = (unsigned)avail;
                bs_.next_out = dst + got;
                bs_.avail_out = (unsigned)std::min<size_t>(n - got, 1u << 30);
                const unsigned out0 = bs_.avail_out;
                const int r = bz2_->decompress(&bs_);
                if (r != codecs::BZ_OK && r != codecs::BZ_STREAM_END) die("read error: invalid bzip2 stream");
                pos_ += avail - bs_.avail_in;
                got += out0 - bs_.avail_out;
                mid_stream_ = r != codecs::BZ_STREAM_END;
                if (r == codecs::BZ_STREAM_END) {  // the next stream of a concatenation (pbzip2, cat a.bz2 b.bz2)
                    bz2_->decompress_end(&bs_);
                    std::memset(&bs_, 0, sizeof bs_);
                    if (bz2_->decompress_init(&bs_, 0, 0) != codecs::BZ_OK) die("bzip2 initialisation failed");
                }
            } else {
                ls_.next_in = (const uint8_t *)(raw_.data() + pos_);
                ls_.avail_in = avail;
                ls_.next_out = (uint8_t *)(dst + got);
                ls_.avail_out = n - got;
                const int r = lzma_->code(&ls_, raw_eof_ && !avail ? codecs::LZMA_FINISH : codecs::LZMA_RUN);
                pos_ += avail - ls_.avail_in;
                got += (n - got) - ls_.avail_out;
                if (r == codecs::LZMA_STREAM_END) done_ = true;
                else if (r != codecs::LZMA_OK) die("read error: invalid xz stream");
            }
        }
        return got;
    }

  private:
    // where fastgz::GzReader gets its bytes: what raw_ still holds, then the file itself (no second copy)
    static size_t gz_source(void *ctx, unsigned char *dst, size_t cap) {
        Input *self = (Input *)ctx;
        if (self->pos_ < self->end_) {
            const size_t m = std::min(cap, self->end_ - self->pos_);
            std::memcpy(dst, self->raw_.data() + self->pos_, m);
            self->pos_ += m;
            return m;
        }
        while (!self->raw_eof_) {
            const ssize_t r = ::read(self->fd_, dst, cap);
            if (r < 0) {
                if (errno == EINTR) continue;
                die("read error");
            }
            if (r == 0) self->raw_eof_ = true;
            return (size_t)r;
        }
        return 0;
    }
    // BGZF: a gzip member whose extra field holds the subfield 'B','C' with the member's total size - 1 (SAM spec 4.1).
    // Returns that total size, or 0 when the bytes at p are not the start of such a member (or too few to tell).
    static size_t bgzf_block_size(const unsigned char *p, size_t n) {
        if (n < 18 || p[0] != 0x1F || p[1] != 0x8B || p[2] != 8 || !(p[3] & 4)) return 0;
        const size_t xlen = p[10] | (size_t)p[11] << 8;
        if (n < 12 + xlen) return 0;
        for (size_t o = 12; o + 4 <= 12 + xlen;) {
            const size_t slen = p[o + 2] | (size_t)p[o + 3] << 8;
            if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) return (size_t)(p[o + 4] | (size_t)p[o + 5] << 8) + 1;
            o += 4 + slen;
        }
        return 0;
    }
    // Reads the next run of BGZF members (up to ~16 MB of them) and inflates them on a few threads, each member into its
    // own place of bz_out_ (its uncompressed size is its last four bytes).  false: no blocked member at the read position --
    // the input has ended, or an ordinary member follows (`cat a.bgz b.gz`): from there on the stream decoder reads.
    bool fill_bgzf() {
        bz_out_.clear();
        bz_pos_ = 0;
        struct Blk {
            size_t in_off, in_len, out_off, out_len;
        };
        std::vector<Blk> blks;
        size_t out_total = 0;
        bz_in_.clear();
        for (;;) {
            // at least a header's worth of bytes in raw_ (a member is at most 64 KB, raw_ is 1 MB)
            if (end_ - pos_ < 18 && !raw_eof_) {
                std::memmove(raw_.data(), raw_.data() + pos_, end_ - pos_);
                end_ -= pos_;
                pos_ = 0;
                top_up();
            }
            const size_t bs = bgzf_block_size((const unsigned char *)raw_.data() + pos_, end_ - pos_);
            if (bs == 0) break;
            if (end_ - pos_ < bs) {
                if (raw_eof_) die("read error: truncated gzip stream");
                std::memmove(raw_.data(), raw_.data() + pos_, end_ - pos_);
                end_ -= pos_;
                pos_ = 0;
                top_up();
                if (end_ - pos_ < bs) {
                    if (raw_eof_) die("read error: truncated gzip stream");
                    continue;
                }
            }
            const unsigned char *b = (const unsigned char *)raw_.data() + pos_;
            const size_t xlen = b[10] | (size_t)b[11] << 8;
            if (bs < 12 + xlen + 8) die("read error: invalid gzip stream");
            const size_t isize = b[bs - 4] | (size_t)b[bs - 3] << 8 | (size_t)b[bs - 2] << 16 | (size_t)b[bs - 1] << 24;
            if (isize > 65536) die("read error: invalid gzip stream");
            blks.push_back({bz_in_.size(), bs, out_total, isize});
            bz_in_.insert(bz_in_.end(), b, b + bs);
            out_total += isize;
            pos_ += bs;
            if (out_total >= (16u << 20)) break;
        }
        if (blks.empty()) {
            if (end_ == pos_ && raw_eof_) done_ = true;  // end of input (the empty end-of-file member has been passed)
            else bgzf_ = false;                           // an ordinary member: the stream decoder goes on from pos_
            return false;
        }
        bz_out_.resize(out_total);
        size_t nthreads = std::min<size_t>(std::max<size_t>(1, usable_cpus() / 2), 16);
        if (const char *e = std::getenv("DCN_CLI_BGZF_THREADS")) nthreads = (size_t)std::max(1, std::atoi(e));
        nthreads = std::min(nthreads, blks.size());
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        bz_in_.insert(bz_in_.end(), fastgz::PAD, 0);  // (the fast decoder loads eight bytes at a time, a little past the end)
        auto work_fast = [&] {
            // a member is decoded into a place with the slack the decoder writes into, checked, and copied to its own
            std::unique_ptr<fastgz::BlockDecoder> dec(new fastgz::BlockDecoder());
            std::vector<unsigned char> tmp(65536 + 258 + 16 + 64);
            for (size_t i; (i = next.fetch_add(1)) < blks.size();) {
                const Blk &k = blks[i];
                const unsigned char *b = bz_in_.data() + k.in_off;
                const size_t xlen = b[10] | (size_t)b[11] << 8;
                const uint32_t want = b[k.in_len - 8] | (uint32_t)b[k.in_len - 7] << 8 | (uint32_t)b[k.in_len - 6] << 16 | (uint32_t)b[k.in_len - 5] << 24;
                if (!fastgz::inflate_whole(*dec, b + 12 + xlen, k.in_len - 12 - xlen - 8, tmp.data(), k.out_len) ||
                    fastgz::crc32_fast(0, tmp.data(), k.out_len) != want) {
                    bad = true;
                    continue;
                }
                if (k.out_len) std::memcpy(bz_out_.data() + k.out_off, tmp.data(), k.out_len);
            }
        };
        auto work_zlib = [&] {
            z_stream z;
            std::memset(&z, 0, sizeof z);
            if (inflateInit2(&z, -15) != Z_OK) {
                bad = true;
                return;
            }
            for (size_t i; (i = next.fetch_add(1)) < blks.size();) {
                const Blk &k = blks[i];
                const unsigned char *b = bz_in_.data() + k.in_off;
                const size_t xlen = b[10] | (size_t)b[11] << 8;
                z.next_in = (Bytef *)(b + 12 + xlen);
                z.avail_in = (uInt)(k.in_len - 12 - xlen - 8);
                unsigned char none = 0;  // (an empty member -- the end-of-file marker -- still wants a place to write to)
                z.next_out = k.out_len ? (Bytef *)(bz_out_.data() + k.out_off) : &none;
                z.avail_out = (uInt)k.out_len;
                const int r = k.out_len ? inflate(&z, Z_FINISH) : inflate(&z, Z_SYNC_FLUSH);
                const uLong crc = crc32(crc32(0L, Z_NULL, 0), k.out_len ? (const Bytef *)(bz_out_.data() + k.out_off) : &none, (uInt)k.out_len);
                const uLong want = b[k.in_len - 8] | (uLong)b[k.in_len - 7] << 8 | (uLong)b[k.in_len - 6] << 16 | (uLong)b[k.in_len - 5] << 24;
                if (r != Z_STREAM_END || z.avail_out != 0 || crc != want) bad = true;
                if (inflateReset(&z) != Z_OK) bad = true;
            }
            inflateEnd(&z);
        };
        std::vector<std::thread> ts;
        auto work = [&] { fast_ ? work_fast() : work_zlib(); };
        for (size_t t = 1; t < nthreads; ++t) ts.emplace_back(work);
        work();
        for (auto &t : ts) t.join();
        if (bad) die("read error: invalid gzip stream");
        return true;
    }
    // more bytes behind end_ (refill() starts over at 0; here what is left stays in front)
    void top_up() {
        while (end_ < raw_.size() && !raw_eof_) {
            ssize_t r = ::read(fd_, raw_.data() + end_, raw_.size() - end_);
            if (r < 0) {
                if (errno == EINTR) continue;
                die("read error");
            }
            if (r == 0) raw_eof_ = true;
            else end_ += (size_t)r;
            if (end_ >= 65536 + 4096) break;
        }
    }
    void refill() {
        pos_ = end_ = 0;
        while (end_ < raw_.size() && !raw_eof_) {
            ssize_t r = ::read(fd_, raw_.data() + end_, raw_.size() - end_);
            if (r < 0) {
                if (errno == EINTR) continue;
                die("read error");
            }
            if (r == 0) raw_eof_ = true;
            else end_ += (size_t)r;
            if (end_ >= 4096) break;  // enough to go on with; pipes deliver what they have
        }
    }
    enum Kind { PLAIN, GZIP, ZSTD, XZ, BZ2 } kind_ = PLAIN;
    int fd_ = -1;
    std::vector<char> raw_;
    size_t pos_ = 0, end_ = 0;
    bool raw_eof_ = false, done_ = false, mid_stream_ = false;
    bool bgzf_ = false;                    // the members at the read position are BGZF blocks
    std::vector<unsigned char> bz_in_;     // a run of whole members ...
    std::vector<char> bz_out_;             // ... and what they inflate to
    size_t bz_pos_ = 0;
    z_stream zs_;
    bool fast_ = false;                    // gzip through fast_inflate.hpp (default) rather than zlib (DCN_CLI_ZLIB_INFLATE=1)
    std::unique_ptr<fastgz::GzReader> gz_;
    std::unique_ptr<fastgz::ParallelGzReader> pgz_;
    const codecs::Zstd *zstd_ = nullptr;
    void *zds_ = nullptr;
    const codecs::Lzma *lzma_ = nullptr;
    codecs::lzma_stream ls_;
    const codecs::Bz2 *bz2_ = nullptr;
    codecs::bz_stream bs_;
};

// ---- output: plain, gzip, zstd or xz by extension (get_writer, src/local_filter.rs:110-151) ---------------------------
// Compressed outputs are written as a sequence of complete members (gzip members, zstd frames, xz streams), one per
// batch: concatenations of those are valid files of their formats (what pigz / bgzip / `cat a.zst b.zst` produce), and
// a member depends on nothing but its own bytes -- so the formatter threads compress their batches side by side and
// the writer only appends, where the reference's single encoder (get_writer, src/local_filter.rs:110-151) compresses
// on one thread.
class Output {
  public:
    enum Codec { PLAIN, GZIP, ZSTD, XZ };
    // the codec an output path asks for by its extension, with the level and the library checked (dies otherwise)
    static Codec check_level(const std::string &path, int level) {
        std::string why;
        if (ends_with(path, ".gz")) {
            if (level < 1 || level > 9) die("Invalid gzip compression level " + std::to_string(level) + ". Must be between 1 and 9.");
            return GZIP;
        } else if (ends_with(path, ".zst")) {
            if (level < 1 || level > 22) die("Invalid zstd compression level " + std::to_string(level) + ". Must be between 1 and 22.");
            if (!codecs::Zstd::get(&why)) die("zstd output " + path + ": " + why);
            return ZSTD;
        } else if (ends_with(path, ".xz")) {
            if (level < 0 || level > 9) die("Invalid xz compression level " + std::to_string(level) + ". Must be between 0 and 9.");
            if (!codecs::Lzma::get(&why)) die("xz output " + path + ": " + why);
            return XZ;
        }
        return PLAIN;
    }
    Output(const std::string &path, int level) : level_(level) {
        codec_ = check_level(path, level);
        if (path == "-") {
            f_ = stdout;
        } else {
            f_ = std::fopen(path.c_str(), "wb");
            if (!f_) die("Failed to create output file: " + path);
            own_ = true;
        }
        std::setvbuf(f_, nullptr, _IONBF, 0);  // batches arrive as multi-megabyte buffers: no second copy
    }
    ~Output() { close(); }
    Codec codec() const { return codec_; }
    int level() const { return level_; }
    bool plain() const { return codec_ == PLAIN; }

    // one complete member of `codec` holding `in` (thread-safe: every call has its own encoder)
    static void compress_member(Codec codec, int level, const char *in, size_t n, std::vector<char> &out) {
        out.clear();
        if (codec == GZIP) {
            // BGZF: members of <= 65,280 input bytes, each carrying its own compressed size in the 'BC' extra subfield (SAM
            // spec 4.1).  Still a gzip file for every reader (zcat, python gzip, the reference's niffler), and one whose
            // members a reader can find without inflating them: bgzip / htslib index it, and this tool's own input side
            // inflates it on several threads (Input::fill_bgzf).  Costs 26 bytes and a fresh window per 64 KB (~2-3 % of size).
            // DCN_CLI_GZIP_ONE_MEMBER=1 keeps the one member per batch of rounds 2-3.
            static const bool one_member = std::getenv("DCN_CLI_GZIP_ONE_MEMBER") != nullptr;
            // levels 1-3 (the default is 2): the tool's own compressor (fast_deflate.hpp: 2-2.6 x zlib's pace at those levels
            // and no larger); higher levels are zlib's longer searches.  DCN_CLI_ZLIB_DEFLATE=1: zlib for every level.
            static const bool zlib_deflate = std::getenv("DCN_CLI_ZLIB_DEFLATE") != nullptr;
            if (!one_member && level <= 3 && !zlib_deflate) {
                static thread_local std::unique_ptr<fastgz::FastDeflate> enc;
                if (!enc) enc.reset(new fastgz::FastDeflate());
                constexpr size_t BLOCK = 65280;
                out.reserve(n / 3 + (n / BLOCK + 1) * 32 + 64);
                size_t pos = 0;
                do {  // (n == 0: one empty member, which is BGZF's end-of-file marker)
                    const size_t take = std::min(BLOCK, n - pos);
                    const size_t at = out.size();
                    out.resize(at + 18 + fastgz::FastDeflate::bound(take) + 8);
                    unsigned char *h = (unsigned char *)out.data() + at;
                    static const unsigned char head[16] = {0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 'B', 'C', 2, 0};
                    std::memcpy(h, head, 16);
                    const size_t clen = enc->compress((const unsigned char *)in + pos, take, h + 18), total = 18 + clen + 8;
                    if (total > 65536) die("write error: gzip member too large");
                    h[16] = (unsigned char)((total - 1) & 0xFF);
                    h[17] = (unsigned char)((total - 1) >> 8);
                    const uint32_t crc = fastgz::crc32_fast(0, (const unsigned char *)in + pos, take);
                    unsigned char *t = h + 18 + clen;
                    for (int i = 0; i < 4; ++i) t[i] = (unsigned char)(crc >> (8 * i)), t[4 + i] = (unsigned char)((uint32_t)take >> (8 * i));
                    out.resize(at + total);
                    pos += take;
                } while (pos < n);
                return;
            }
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, level, Z_DEFLATED, one_member ? 15 + 16 : -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("gzip initialisation failed");
            if (!one_member) {
                constexpr size_t BLOCK = 65280;
                out.reserve(n / 3 + (n / BLOCK + 1) * 32 + 64);
                size_t pos = 0;
                do {  // (n == 0: one empty member, which is BGZF's end-of-file marker)
                    const size_t take = std::min(BLOCK, n - pos);
                    const size_t at = out.size();
                    out.resize(at + 18 + deflateBound(&zs, (uLong)take) + 8);
                    unsigned char *h = (unsigned char *)out.data() + at;
                    static const unsigned char head[16] = {0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 'B', 'C', 2, 0};
                    std::memcpy(h, head, 16);
                    zs.next_in = (Bytef *)(in + pos);
                    zs.avail_in = (uInt)take;
                    zs.next_out = h + 18;
                    zs.avail_out = (uInt)(out.size() - at - 18 - 8);
                    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) die("write error: gzip");
                    const size_t clen = zs.total_out, total = 18 + clen + 8;
                    if (total > 65536) die("write error: gzip member too large");
                    h[16] = (unsigned char)((total - 1) & 0xFF);
                    h[17] = (unsigned char)((total - 1) >> 8);
                    const uLong crc = crc32(crc32(0L, Z_NULL, 0), (const Bytef *)(in + pos), (uInt)take);
                    unsigned char *t = h + 18 + clen;
                    for (int i = 0; i < 4; ++i) t[i] = (unsigned char)(crc >> (8 * i)), t[4 + i] = (unsigned char)((uint32_t)take >> (8 * i));
                    out.resize(at + total);
                    pos += take;
                    if (deflateReset(&zs) != Z_OK) die("write error: gzip");
                } while (pos < n);
                deflateEnd(&zs);
                return;
            }
            out.resize(deflateBound(&zs, (uLong)std::min<size_t>(n, 1u << 30)) + (n >> 10) + 64);  // grown below if short
            size_t in_pos = 0;
            for (;;) {
                const size_t take = std::min<size_t>(n - in_pos, 1u << 30);
                zs.next_in = (Bytef *)(in + in_pos);
                zs.avail_in = (uInt)take;
                const bool last = in_pos + take == n;
                int r = Z_OK;
                do {
                    if (zs.total_out == out.size()) out.resize(out.size() * 2 + 4096);
                    zs.next_out = (Bytef *)out.data() + zs.total_out;
                    zs.avail_out = (uInt)std::min<size_t>(out.size() - zs.total_out, 1u << 30);
                    r = deflate(&zs, last ? Z_FINISH : Z_NO_FLUSH);
                    if (r == Z_STREAM_ERROR) die("write error: gzip");
                } while (zs.avail_in || (last && r != Z_STREAM_END));
                in_pos += take;
                if (last) break;
            }
            out.resize(zs.total_out);
            deflateEnd(&zs);
        } else if (codec == ZSTD) {
            const codecs::Zstd *z = codecs::Zstd::get(nullptr);
            void *cs = z ? z->createCStream() : nullptr;
            if (!cs || z->isError(z->initCStream(cs, level))) die("zstd initialisation failed");
            out.resize(n / 2 + (1 << 16));
            size_t pos = 0;
            codecs::ZSTD_inBuffer ib = {in, n, 0};
            auto room = [&] {
                if (out.size() - pos < (1u << 16)) out.resize(out.size() * 2);
            };
            while (ib.pos < ib.size) {
                room();
                codecs::ZSTD_outBuffer ob = {out.data() + pos, out.size() - pos, 0};
                if (z->isError(z->compressStream(cs, &ob, &ib))) die("write error: zstd");
                pos += ob.pos;
            }
            for (size_t left = 1; left;) {
                room();
                codecs::ZSTD_outBuffer ob = {out.data() + pos, out.size() - pos, 0};
                left = z->endStream(cs, &ob);
                if (z->isError(left)) die("write error: zstd");
                pos += ob.pos;
            }
            z->freeCStream(cs);
            out.resize(pos);
        } else if (codec == XZ) {
            const codecs::Lzma *l = codecs::Lzma::get(nullptr);
            codecs::lzma_stream ls;
            std::memset(&ls, 0, sizeof ls);
            if (!l || l->easy_encoder(&ls, (uint32_t)level, codecs::LZMA_CHECK_CRC64) != codecs::LZMA_OK) die("xz initialisation failed");
            out.resize(n / 2 + (1 << 16));
            size_t pos = 0;
            ls.next_in = (const uint8_t *)in;
            ls.avail_in = n;
            int r = codecs::LZMA_OK;
            while (r == codecs::LZMA_OK) {
                if (out.size() - pos < (1u << 16)) out.resize(out.size() * 2);
                ls.next_out = (uint8_t *)out.data() + pos;
                ls.avail_out = out.size() - pos;
                r = l->code(&ls, ls.avail_in ? codecs::LZMA_RUN : codecs::LZMA_FINISH);
                pos = out.size() - ls.avail_out;
                if (r != codecs::LZMA_OK && r != codecs::LZMA_STREAM_END) die("write error: xz");
            }
            l->end(&ls);
            out.resize(pos);
        } else {
            out.assign(in, in + n);
        }
    }
    // formatted records: compressed here when the output is (callers that compressed the batch themselves use write_raw)
    void write(const std::vector<char> &buf) {
        if (buf.empty()) return;
        if (codec_ == PLAIN) {
            raw_write(buf.data(), buf.size());
        } else {
            compress_member(codec_, level_, buf.data(), buf.size(), cbuf_);
            write_raw(cbuf_);
        }
    }
    void write_raw(const std::vector<char> &bytes) {
        if (bytes.empty()) return;
        wrote_member_ = true;
        raw_write(bytes.data(), bytes.size());
    }
    // plain streams only: the pieces of a batch in one gather write per IOV_MAX entries
    void write_gather(std::vector<struct iovec> &iov) {
        const int fd = fileno(f_);
        size_t i = 0;
        while (i < iov.size()) {
            const int n = (int)std::min<size_t>(iov.size() - i, 1024);
            ssize_t w = ::writev(fd, iov.data() + i, n);
            if (w < 0) {
                if (errno == EINTR) continue;
                die("write error");
            }
            while (w > 0 && i < iov.size()) {  // a short write ends inside some entry
                if ((size_t)w >= iov[i].iov_len) {
                    w -= (ssize_t)iov[i].iov_len;
                    ++i;
                } else {
                    iov[i].iov_base = (char *)iov[i].iov_base + w;
                    iov[i].iov_len -= (size_t)w;
                    w = 0;
                }
            }
        }
    }
    // the last flush is where a full disk or a closed pipe shows: a failure here must not end in "Retained ..."
    void close() {
        FILE *f = f_;
        if (!f) return;
        // nothing was kept: still a valid (empty) file of its format; a gzip file of BGZF members ends with the empty member
        // that bgzip / htslib read as "not truncated"
        if (codec_ != PLAIN && (!wrote_member_ || (codec_ == GZIP && !std::getenv("DCN_CLI_GZIP_ONE_MEMBER")))) {
            compress_member(codec_, level_, "", 0, cbuf_);
            raw_write(cbuf_.data(), cbuf_.size());
        }
        f_ = nullptr;
        if (std::fflush(f) != 0) die("write error");
        if (own_ && std::fclose(f) != 0) die("write error");
    }

  private:
    void raw_write(const char *p, size_t n) {
        if (n && std::fwrite(p, 1, n, f_) != n) die("write error");
    }
    FILE *f_ = nullptr;
    bool own_ = false, wrote_member_ = false;
    Codec codec_ = PLAIN;
    int level_ = 0;
    std::vector<char> cbuf_;
};

// ---- FASTA / FASTQ records -------------------------------------------------------------------------------------
constexpr uint64_t NO_QUAL = ~0ull;
struct Rec {
    uint64_t id_off;   // header line without the leading '>' / '@', in Batch::chars()
    uint32_t id_len;
    uint32_t seq_len;
    uint64_t seq_off;  // newline-free sequence in Batch::bases
    uint64_t qual_off; // in Batch::chars(); NO_QUAL for FASTA
    // parallel reader only: the record's bytes in the mapped input, when they are exactly what format_record_to_buffer
    // would write for it ("@id\nseq\n+\nqual\n" / ">id\nseq\n": one sequence line, a bare '+', no '\r'); 0 = format it
    uint64_t rec_off = 0;
    uint32_t rec_len = 0;
};

// std::allocator value-initialises on resize(): a 12 MB zero fill per parsed chunk that the parser overwrites at once
template <typename T>
struct DefaultInitAllocator : std::allocator<T> {
    template <typename U>
    struct rebind {
        using other = DefaultInitAllocator<U>;
    };
    using std::allocator<T>::allocator;
    template <typename U>
    void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) {
        ::new (static_cast<void *>(p)) U;
    }
    template <typename U, typename... Args>
    void construct(U *p, Args &&...args) {
        ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
    }
};

struct Batch {
    uint64_t seq_no = 0;
    std::vector<char> text;      // ids and qualities (streaming reader) ...
    const char *ext = nullptr;   // ... or the memory-mapped input file they live in (parallel reader)
    const char *chars() const { return ext ? ext : text.data(); }
    std::vector<char> out1, out2;  // formatted kept records (filled by the format stage)
    std::vector<char> comp1, comp2;  // ... and, for a compressed output, their complete member (same stage)
    bool compressed = false;
    std::vector<struct iovec> iov1;  // plain single-file output: what to write, in order -- ranges of the mapped input
                                     // (records that already have their output form) and pieces of out1
    std::vector<uint8_t, DefaultInitAllocator<uint8_t>> bases;  // concatenated sequences (what dcn_filter_batch takes)
    // ... or, from the chunk parsers, the same stream already in the form dcn_filter_batch_packed takes (2 bits per base
    // + 1 invalid bit per base, whole 32-base groups): every Rec::seq_off then points at the record's one sequence line
    // in chars() instead of into `bases`
    std::vector<uint32_t, DefaultInitAllocator<uint32_t>> packed, invmask;
    bool seq_in_chars = false;
    const char *seq_ptr(const Rec &r) const;
    std::vector<uint64_t> offsets{0};
    std::vector<uint32_t> unit_id;
    std::vector<Rec> recs;
    std::vector<uint8_t> keep;
    std::vector<uint32_t> hits, total;
    // the calls this batch was cut into (normally one): sequence numbers handed out by the multi-GPU driver and the
    // rebased offsets / unit ids of the later pieces, alive until the calls are waited for
    std::vector<uint64_t> gpu_seqs;
    std::vector<std::vector<uint64_t>> sub_off;
    std::vector<std::vector<uint32_t>> sub_uid;
    std::vector<std::vector<uint32_t>> sub_packed, sub_mask;  // (packed batches: the later pieces' bits, moved to base 0)
    uint64_t out_off = 0, out_bytes = 0;  // mapped output: where this batch's kept records go, and how many bytes
    uint64_t out_off2 = 0, out_bytes2 = 0;  // ... and the second mates', when they have a mapped file of their own (-O)
    size_t raw_end = 0;     // chunk reader: `text` holds raw input, whole records in [0, raw_end) ...
    bool raw_fastq = false;  // ... of this format
    std::vector<char> text2;  // two streams of mates through the chunk readers: the second file's chunk, the same number of
    size_t raw_end2 = 0;      // records in [0, raw_end2) ...
    size_t n_records = 0;     // ... namely this many
    bool paired = false;
    void clear() {
        text.clear();
        text2.clear();
        bases.clear();
        packed.clear();
        invmask.clear();
        seq_in_chars = false;
        offsets.assign(1, 0);
        unit_id.clear();
        recs.clear();
    }
    void reset() {  // back to a fresh batch that keeps its vectors' memory
        clear();
        seq_no = 0;
        ext = nullptr;
        out1.clear();
        out2.clear();
        comp1.clear();
        comp2.clear();
        compressed = false;
        iov1.clear();
        keep.clear();
        hits.clear();
        total.clear();
        gpu_seqs.clear();
        sub_off.clear();
        sub_uid.clear();
        sub_packed.clear();
        sub_mask.clear();
        out_off = out_bytes = 0;
        out_off2 = out_bytes2 = 0;
        raw_end = 0;
        raw_end2 = 0;
        n_records = 0;
        raw_fastq = false;
        paired = false;
    }
};

inline const char *Batch::seq_ptr(const Rec &r) const {
    return seq_in_chars ? chars() + r.seq_off : reinterpret_cast<const char *>(bases.data()) + r.seq_off;
}

// Batches travel reader -> parsers -> GPU stage -> formatters -> writer and come back here: their vectors (12 MB of
// bases, 5 MB of records per chunk) are reused instead of being mapped, page-faulted and unmapped once per chunk
class BatchPool {
  public:
    std::unique_ptr<Batch> get() {
        {
            std::lock_guard<std::mutex> l(m_);
            if (!free_.empty()) {
                std::unique_ptr<Batch> b = std::move(free_.back());
                free_.pop_back();
                return b;
            }
        }
        return std::unique_ptr<Batch>(new Batch());
    }
    void put(std::unique_ptr<Batch> b) {
        if (!b) return;
        b->reset();
        std::lock_guard<std::mutex> l(m_);
        free_.push_back(std::move(b));
    }

  private:
    std::mutex m_;
    std::vector<std::unique_ptr<Batch>> free_;
};

// An Input whose bytes are produced (read + decompressed) by a thread of its own, a few 4 MB blocks ahead of the
// consumer: with two gzip files (paired reads) the two inflates then run beside each other and beside the record
// parser instead of taking turns on one thread.
class AsyncInput {
  public:
    explicit AsyncInput(const std::string &path) : in_(path), th_([this] { produce(); }) {}
    ~AsyncInput() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    size_t read(char *dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (pos_ == cur_.size()) {
                std::unique_lock<std::mutex> l(m_);
                if (!cur_.empty() || cur_.capacity()) spare_.push_back(std::move(cur_));
                cur_.clear();
                pos_ = 0;
                cv_.wait(l, [&] { return !full_.empty() || eof_; });
                if (full_.empty()) break;  // end of input
                cur_ = std::move(full_.front());
                full_.pop_front();
                cv_.notify_all();
                continue;
            }
            const size_t take = std::min(n - got, cur_.size() - pos_);
            std::memcpy(dst + got, cur_.data() + pos_, take);
            pos_ += take;
            got += take;
        }
        return got;
    }

  private:
    void produce() {
        for (;;) {
            std::vector<char> buf;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return full_.size() < 4 || stop_; });
                if (stop_) return;
                if (!spare_.empty()) {
                    buf = std::move(spare_.back());
                    spare_.pop_back();
                }
            }
            buf.resize(4u << 20);
            const size_t got = in_.read(buf.data(), buf.size());
            buf.resize(got);
            std::lock_guard<std::mutex> l(m_);
            if (got == 0) {
                eof_ = true;
                cv_.notify_all();
                return;
            }
            full_.push_back(std::move(buf));
            cv_.notify_all();
        }
    }
    Input in_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::vector<char>> full_;
    std::vector<std::vector<char>> spare_;
    std::vector<char> cur_;
    size_t pos_ = 0;
    bool eof_ = false, stop_ = false;
    std::thread th_;  // last: everything above exists before it starts
};

// streaming parser over a refillable window; one record at a time, appended to a Batch
class FastxReader {
  public:
    explicit FastxReader(const std::string &path) : in_(path), buf_(1 << 24) {}

    // appends the next record to b; false at end of input
    bool next(Batch &b) {
        std::string_view line;
        do {
            if (!getline(line)) return false;
        } while (line.empty());
        char marker = line[0];
        if (marker != '>' && marker != '@') die("Invalid FASTX record start: expected '>' or '@'");
        Rec r;
        r.id_off = b.text.size();
        r.id_len = (uint32_t)line.size() - 1;
        b.text.insert(b.text.end(), line.begin() + 1, line.end());
        r.seq_off = b.bases.size();
        if (marker == '>') {
            // FASTA: sequence lines until the next header (multi-line records are joined, src/local_filter.rs:347)
            while (peek() != '>' && peek() != 0) {
                if (!getline(line)) break;
                b.bases.insert(b.bases.end(), line.begin(), line.end());
            }
            r.qual_off = NO_QUAL;
        } else {
            if (!getline(line)) die("Truncated FASTQ record");
            b.bases.insert(b.bases.end(), line.begin(), line.end());
            if (!getline(line) || line.empty() || line[0] != '+') die("Invalid FASTQ record: missing '+' line");
            if (!getline(line)) die("Truncated FASTQ record");
            r.qual_off = b.text.size();
            b.text.insert(b.text.end(), line.begin(), line.end());
            if (line.size() != b.bases.size() - r.seq_off) die("FASTQ sequence and quality lengths differ");
        }
        r.seq_len = (uint32_t)(b.bases.size() - r.seq_off);
        b.recs.push_back(r);
        b.offsets.push_back(b.bases.size());
        return true;
    }

  private:
    void refill() {
        if (eof_) return;
        size_t rem = end_ - pos_;
        if (pos_ > 0) std::memmove(buf_.data(), buf_.data() + pos_, rem);
        pos_ = 0;
        end_ = rem;
        if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
        size_t got = in_.read(buf_.data() + end_, buf_.size() - end_);
        end_ += got;
        if (got == 0) eof_ = true;
    }
    char peek() {
        if (pos_ == end_) refill();
        return pos_ < end_ ? buf_[pos_] : 0;
    }
    bool getline(std::string_view &out) {
        for (;;) {
            char *nl = (char *)std::memchr(buf_.data() + pos_, '\n', end_ - pos_);
            if (nl) {
                size_t len = nl - (buf_.data() + pos_);
                out = std::string_view(buf_.data() + pos_, len);
                pos_ += len + 1;
                if (!out.empty() && out.back() == '\r') out.remove_suffix(1);
                return true;
            }
            if (eof_) {
                if (pos_ == end_) return false;
                out = std::string_view(buf_.data() + pos_, end_ - pos_);
                pos_ = end_;
                return true;
            }
            refill();
        }
    }
    AsyncInput in_;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
};

template <typename T>
class Queue {  // bounded hand-off between the pipeline threads
  public:
    explicit Queue(size_t cap) : cap_(cap) {}
    void push(T v) {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(v));
        cv_.notify_all();
    }
    bool pop(T &out) {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return !q_.empty() || done_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        cv_.notify_all();
        return true;
    }
    void finish() {
        std::lock_guard<std::mutex> l(m_);
        done_ = true;
        cv_.notify_all();
    }

  private:
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<T> q_;
    size_t cap_;
    bool done_ = false;
};


// A pool of workers applying fn to batches; results come out in push order (the GPU stage and the writer need order).
class OrderedStage {
  public:
    OrderedStage(size_t threads, size_t max_inflight, std::function<void(Batch &)> fn)
        : fn_(std::move(fn)), max_inflight_(max_inflight) {
        for (size_t i = 0; i < threads; ++i) workers_.emplace_back([this] { work(); });
    }
    ~OrderedStage() {
        finish();
        for (auto &t : workers_) t.join();
    }
    void push(std::unique_ptr<Batch> b) {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return inflight_ < max_inflight_; });
        todo_.emplace_back(next_push_++, std::move(b));
        ++inflight_;
        cv_.notify_all();
    }
    bool pop(std::unique_ptr<Batch> &out) {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return done_.count(next_pop_) || (finished_ && next_pop_ == next_push_); });
        auto it = done_.find(next_pop_);
        if (it == done_.end()) return false;
        out = std::move(it->second);
        done_.erase(it);
        ++next_pop_;
        --inflight_;
        cv_.notify_all();
        return true;
    }
    void finish() {
        std::lock_guard<std::mutex> l(m_);
        finished_ = true;
        cv_.notify_all();
    }

  private:
    void work() {
        for (;;) {
            std::pair<uint64_t, std::unique_ptr<Batch>> job;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return !todo_.empty() || finished_; });
                if (todo_.empty()) return;
                job = std::move(todo_.front());
                todo_.pop_front();
            }
            fn_(*job.second);
            std::lock_guard<std::mutex> l(m_);
            done_.emplace(job.first, std::move(job.second));
            cv_.notify_all();
        }
    }
    std::function<void(Batch &)> fn_;
    size_t max_inflight_, inflight_ = 0;
    uint64_t next_push_ = 0, next_pop_ = 0;
    bool finished_ = false;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::pair<uint64_t, std::unique_ptr<Batch>>> todo_;
    std::map<uint64_t, std::unique_ptr<Batch>> done_;
    std::vector<std::thread> workers_;
};

// ---- parallel ingest of a plain (uncompressed) regular file: mmap + record-aligned chunks -----------------------
struct MappedFile {
    const char *data = nullptr;
    size_t size = 0;
    bool open(const std::string &path) {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) {
            ::close(fd);
            return false;
        }
        void *p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) return false;
        data = (const char *)p;
        size = (size_t)st.st_size;
        const unsigned char *m = (const unsigned char *)data;
        if (codecs::is_gzip_magic(m, size) || codecs::is_zstd_magic(m, size) || codecs::is_xz_magic(m, size) || codecs::is_bz2_magic(m, size)) {  // compressed: the streaming reader
            munmap(p, size);
            data = nullptr;
            return false;
        }
        madvise(p, size, MADV_SEQUENTIAL);
        return true;
    }
    ~MappedFile() {
        if (data) munmap((void *)data, size);
    }
};

inline size_t line_end(const char *d, size_t size, size_t p) {  // index of the '\n' ending the line at p (or size)
    const void *nl = p < size ? std::memchr(d + p, '\n', size - p) : nullptr;
    return nl ? (size_t)((const char *)nl - d) : size;
}

// first record start at or after `from` (a line start): FASTA: a line beginning with '>'; FASTQ: a line beginning
// with '@' whose third line begins with '+' and whose second and fourth lines have equal length
size_t next_record_start(const char *d, size_t size, size_t from, bool fastq) {
    size_t p = from;
    if (p > 0 && p < size && d[p - 1] != '\n') p = line_end(d, size, p) + 1;  // align to a line start
    while (p < size) {
        if (!fastq) {
            if (d[p] == '>') return p;
        } else if (d[p] == '@') {
            size_t e0 = line_end(d, size, p), e1 = line_end(d, size, e0 + 1), e2 = line_end(d, size, e1 + 1);
            size_t e3 = line_end(d, size, e2 + 1);
            if (e1 + 1 < size && d[e1 + 1] == '+' && e2 < size) {
                size_t l1 = e1 - (e0 + 1), l3 = e3 - (e2 + 1);
                if (l1 && d[e1 - 1] == '\r') --l1;
                if (l3 && e3 > e2 + 1 && d[e3 - 1] == '\r') --l3;
                if (l1 == l3) return p;
            }
        }
        p = line_end(d, size, p) + 1;
    }
    return size;
}

// Largest b <= n such that d[0, b) holds whole records only, given that more input follows d[0, n) (0: not even one
// record is complete yet).  FASTQ: a verified record start in the last megabytes, then record by record while all four
// lines end inside the buffer; FASTA: the last header line (its record may still grow).
size_t last_record_boundary(const char *d, size_t n, bool fastq) {
    if (!fastq) {
        for (size_t i = n; i-- > 1;)
            if (d[i] == '>' && d[i - 1] == '\n') return i;
        return 0;
    }
    for (size_t window = 4u << 20;; window *= 4) {
        const size_t from = n > window ? n - window : 0;
        size_t p = next_record_start(d, n, from, true);
        if (p < n) {
            for (;;) {
                while (p < n && d[p] == '\n') ++p;  // blank lines between records
                const size_t e0 = line_end(d, n, p), e1 = line_end(d, n, e0 + 1), e2 = line_end(d, n, e1 + 1);
                const size_t e3 = line_end(d, n, e2 + 1);
                if (e3 >= n) return p;  // this record's last line does not end here: it stays for the next chunk
                p = e3 + 1;
            }
        }
        if (from == 0) return 0;
    }
}

// index of the first '\n' in [p, size), or size.  Lines of a short-read file are 10-300 bytes: a libc call per line
// costs more than the search, so the first 128 bytes are looked at here, 32 at a time
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline size_t line_end_avx2(const char *d, size_t size, size_t p) {
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t q = p;
    for (int i = 0; i < 4 && q + 32 <= size; ++i, q += 32) {
        const unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(d + q)), nl));
        if (m) return q + (size_t)__builtin_ctz(m);
    }
    return line_end(d, size, q);
}
#endif

#if defined(__x86_64__)
// the same with 64 bytes per compare where the host has AVX-512BW (a 150-byte line: three steps instead of five); bytes at
// or past `size` are never loaded (masked load)
__attribute__((target("avx512f,avx512bw"))) inline size_t line_end_avx512(const char *d, size_t size, size_t p) {
    const __m512i nl = _mm512_set1_epi8('\n');
    size_t q = p;
    for (int i = 0; i < 3; ++i, q += 64) {
        if (q + 64 <= size) {
            const uint64_t m = _mm512_cmpeq_epi8_mask(_mm512_loadu_si512((const void *)(d + q)), nl);
            if (m) return q + (size_t)__builtin_ctzll(m);
        } else {
            if (q >= size) return size;
            const __mmask64 k = (~0ull) >> (64 - (size - q));
            const uint64_t m = _mm512_cmpeq_epi8_mask(_mm512_maskz_loadu_epi8(k, (const void *)(d + q)), nl) & k;
            return m ? q + (size_t)__builtin_ctzll(m) : size;
        }
    }
    return line_end(d, size, q);
}
#endif

inline bool cli_avx512() {
#if defined(__x86_64__)
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && !std::getenv("DCN_CLI_NO_AVX2") &&
                           !std::getenv("DCN_CLI_NO_AVX512");
    return ok;
#else
    return false;
#endif
}

// The batch stream written in the library's packed form while the records are parsed (dcn_filter_batch_packed,
// include/deacon_hip.h: base i = bits [2(i%32), +2) of the i/32-th 64-bit word, code (c >> 1) & 3; invalid bit i%32 of
// mask word i/32 set iff the byte is not one of ACGTacgt -- PackedSeqVec::from_ascii and the mask loop of
// src/filter_common.rs:238-258).  Sequence lines arrive at any base offset: a 64 + 32-bit accumulator takes 32 bases per
// step and every output word is stored once.  The ASCII never gets a copy of its own: one read of the mapped input,
// 0.375 bytes written per base (the ASCII form writes 1 and has the library's host threads read it again to pack it).
struct PackSink {
    uint32_t *P = nullptr;  // two u32 words per 32-base group, stored as one u64 (little endian)
    uint32_t *M = nullptr;
    uint64_t accP = 0;
    uint32_t accM = 0;
    unsigned fill = 0;  // bases in the accumulator, 0..31
    size_t g = 0;       // groups stored so far
    bool multi = false; // a record whose sequence spans several lines was met (its formatter needs contiguous bases)
    bool wide = cli_avx512();

    inline void put(uint64_t v, uint32_t m, unsigned n) {  // n <= 32 bases, bits above them zero
        accP |= v << (2 * fill);
        accM |= m << fill;
        if (fill + n >= 32) {
            std::memcpy(P + 2 * g, &accP, 8);
            M[g] = accM;
            ++g;
            accP = fill ? v >> (2 * (32 - fill)) : 0;
            accM = fill ? m >> (32 - fill) : 0;
            fill = fill + n - 32;
        } else {
            fill += n;
        }
    }
#if defined(__x86_64__)
    // n bytes at s; bytes up to `limit` may be read (the chunk's end: whatever follows a line there is input as well)
    __attribute__((target("avx2,bmi2"))) void push(const char *s, size_t n, const char *limit) {
        const __m256i lower = _mm256_set1_epi8(0x20);
        const __m256i ca = _mm256_set1_epi8('a'), cc = _mm256_set1_epi8('c'), cg = _mm256_set1_epi8('g'), ct = _mm256_set1_epi8('t');
        const uint64_t sel = 0x0606060606060606ull;  // bits 1..2 of every byte = (c >> 1) & 3
        for (size_t i = 0; i < n; i += 32) {
            const unsigned m = n - i < 32 ? (unsigned)(n - i) : 32u;
            __m256i v;
            if (s + i + 32 <= limit) {
                v = _mm256_loadu_si256((const __m256i *)(s + i));
            } else {
                char buf[32] = {0};
                std::memcpy(buf, s + i, m);
                v = _mm256_loadu_si256((const __m256i *)buf);
            }
            uint64_t code = _pext_u64((uint64_t)_mm256_extract_epi64(v, 0), sel) | (_pext_u64((uint64_t)_mm256_extract_epi64(v, 1), sel) << 16) |
                            (_pext_u64((uint64_t)_mm256_extract_epi64(v, 2), sel) << 32) | (_pext_u64((uint64_t)_mm256_extract_epi64(v, 3), sel) << 48);
            const __m256i l = _mm256_or_si256(v, lower);
            const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(l, ca), _mm256_cmpeq_epi8(l, cc)),
                                               _mm256_or_si256(_mm256_cmpeq_epi8(l, cg), _mm256_cmpeq_epi8(l, ct)));
            uint32_t inv = ~(uint32_t)_mm256_movemask_epi8(ok);
            if (m < 32) {
                code &= (1ull << (2 * m)) - 1;
                inv &= (1u << m) - 1;
            }
            put(code, inv, m);
        }
    }
#else
    void push(const char *, size_t, const char *) {}
#endif
#if defined(__x86_64__)
    // 64 bases per step (AVX-512BW): the codes folded by two multiply-adds and narrowed out of their dwords, the invalid
    // bits out of four byte compares as one 64-bit mask (csrc/host_pack.cpp has the same arithmetic for whole buffers);
    // a masked load takes the last, partial step, so nothing behind the line is touched
    __attribute__((target("avx512f,avx512bw"))) void push512(const char *s, size_t n) {
        const __m512i three = _mm512_set1_epi8(3), lower = _mm512_set1_epi8(0x20);
        const __m512i m14 = _mm512_set1_epi16(0x0401), m116 = _mm512_set1_epi32(0x00100001);
        const __m512i ca = _mm512_set1_epi8('a'), cc = _mm512_set1_epi8('c'), cg = _mm512_set1_epi8('g'), ct = _mm512_set1_epi8('t');
        for (size_t i = 0; i < n; i += 64) {
            const unsigned m = n - i < 64 ? (unsigned)(n - i) : 64u;
            const __mmask64 k = (~0ull) >> (64 - m);
            const __m512i v = m == 64 ? _mm512_loadu_si512((const void *)(s + i)) : _mm512_maskz_loadu_epi8(k, (const void *)(s + i));
            const __m512i code = _mm512_and_si512(_mm512_srli_epi16(v, 1), three);  // (zero bytes give code 0)
            const __m128i q = _mm512_cvtepi32_epi8(_mm512_madd_epi16(_mm512_maddubs_epi16(code, m14), m116));
            const __m512i l = _mm512_or_si512(v, lower);
            const uint64_t ok = _mm512_cmpeq_epi8_mask(l, ca) | _mm512_cmpeq_epi8_mask(l, cc) | _mm512_cmpeq_epi8_mask(l, cg) |
                                _mm512_cmpeq_epi8_mask(l, ct);
            const uint64_t inv = ~ok & k;
            put((uint64_t)_mm_cvtsi128_si64(q), (uint32_t)inv, m < 32 ? m : 32u);
            if (m > 32) put((uint64_t)_mm_extract_epi64(q, 1), (uint32_t)(inv >> 32), m - 32);
        }
    }
#else
    void push512(const char *, size_t) {}
#endif
    size_t finish() {  // stores the last, partial group (bits behind the stream's end stay zero: 'A', valid); groups written
        if (fill) {
            std::memcpy(P + 2 * g, &accP, 8);
            M[g] = accM;
            ++g;
            accP = 0, accM = 0, fill = 0;
        }
        return g;
    }
};

inline bool cli_can_pack() {  // the packing parser needs AVX2 + BMI2 (pext); other hosts parse to ASCII
#if defined(__x86_64__)
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !std::getenv("DCN_CLI_NO_AVX2") &&
                           !std::getenv("DCN_CLI_NO_PACKED_PARSE");
    return ok;
#else
    return false;
#endif
}

// One record of a mapped file: the record starting at d[p] (p < b, not a blank line) -> its Rec, the position behind it
// returned.  MODE 1: its sequence appended at bases + nb; MODE 2: packed into *ps, Rec::seq_off = the sequence line in the
// mapping; MODE 0: a counting pass, the sequence is not touched.  `shift` is added to the offsets the Rec keeps into the
// mapping (two mapped files share one base pointer: paired inputs).
template <bool AVX2, int MODE>
inline size_t parse_mapped_record(const char *d, size_t p, size_t b, bool fastq, uint8_t *bases, size_t &nb, Rec &r, uint64_t shift,
                                  PackSink *ps = nullptr) {
    auto eol = [&](size_t q) {
#if defined(__x86_64__)
        if (AVX2) return cli_avx512() ? line_end_avx512(d, b, q) : line_end_avx2(d, b, q);
#endif
        return line_end(d, b, q);
    };
    auto trim = [&](size_t s0, size_t e) { return (e > s0 && d[e - 1] == '\r') ? e - 1 : e; };
    const size_t e0 = eol(p);
    if (d[p] != (fastq ? '@' : '>')) die("Invalid FASTX record start: expected '>' or '@'");
    r.id_off = shift + p + 1;
    r.id_len = (uint32_t)(trim(p + 1, e0) - (p + 1));
    const size_t nb0 = nb;
    r.seq_off = nb;
    r.rec_off = 0;
    r.rec_len = 0;
    if (fastq) {
        // (the '+' line of a well-formed record is "+\n": look there before searching)
        size_t s1 = e0 + 1, e1 = eol(s1), s2 = e1 + 1;
        size_t e2 = (s2 + 1 < b && d[s2 + 1] == '\n') ? s2 + 1 : eol(s2);
        size_t s3 = e2 + 1, e3 = eol(s3);
        if (s1 >= b) die("Truncated FASTQ record");  // (the words of the record reader, FastxReader::next)
        if (s2 >= b || d[s2] != '+') die("Invalid FASTQ record: missing '+' line");
        if (s3 >= b) die("Truncated FASTQ record");  // the input ends behind the '+' line: no quality line at all
        size_t t1 = trim(s1, e1), t3 = trim(s3, e3);
        if (t3 - s3 != t1 - s1) die("FASTQ sequence and quality lengths differ");
        if (MODE == 1) std::memcpy(bases + nb, d + s1, t1 - s1);  // sequence = quality length: at most half of the chunk's bytes
        if (MODE == 2) (ps->wide ? ps->push512(d + s1, t1 - s1) : ps->push(d + s1, t1 - s1, d + b)), r.seq_off = shift + s1;
        nb += t1 - s1;
        r.qual_off = shift + s3;
        if (t1 == e1 && t3 == e3 && e2 == s2 + 1 && e3 < b && r.id_len == e0 - (p + 1) && e3 + 1 - p < (1ull << 32)) {
            r.rec_off = shift + p;
            r.rec_len = (uint32_t)(e3 + 1 - p);
        }
        p = e3 + 1;
    } else {
        size_t q = e0 + 1, lines = 0, last_e = 0;
        bool cr = false;
        const size_t first_line = q;
        while (q < b && d[q] != '>') {
            size_t e = eol(q);
            cr = cr || trim(q, e) != e;
            if (MODE == 1) std::memcpy(bases + nb, d + q, trim(q, e) - q);
            if (MODE == 2) (ps->wide ? ps->push512(d + q, trim(q, e) - q) : ps->push(d + q, trim(q, e) - q, d + b));
            nb += trim(q, e) - q;
            q = e + 1;
            last_e = e;
            ++lines;
        }
        r.qual_off = NO_QUAL;
        if (lines == 1 && !cr && last_e < b && r.id_len == e0 - (p + 1) && last_e + 1 - p < (1ull << 32)) {
            r.rec_off = shift + p;
            r.rec_len = (uint32_t)(last_e + 1 - p);
        }
        if (MODE == 2) {
            if (lines > 1) ps->multi = true;
            r.seq_off = shift + first_line;
        }
        p = q;
    }
    r.seq_len = (uint32_t)(nb - nb0);
    return p;
}

inline bool cli_avx2() {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2") && !std::getenv("DCN_CLI_NO_AVX2");
    return avx2;
#else
    return false;
#endif
}

// parse the records in [a, b) of the mapped file into batch (ids / qualities stay in the mapping).  PACKED: the batch's
// stream is written 2-bit packed and the sequences stay in the mapping too; false = a record with several sequence lines
// was met (the formatters want a record's bases in one piece): the caller parses the chunk again in the ASCII form
template <bool AVX2, bool PACKED>
bool parse_mapped_chunk_impl(const char *d, size_t a, size_t b, bool fastq, Batch &out) {
    out.ext = d;
    // the sequences of a chunk are at most its own size: one allocation, written through a raw pointer (insert() /
    // push_back() per record cost more than the copy itself)
    const size_t max_bases = (b - a) / (fastq ? 2 : 1) + 64;
    PackSink ps;
    uint8_t *bases = nullptr;
    if (PACKED) {
        out.bases.clear();
        out.packed.resize(2 * (max_bases / 32 + 2));
        out.invmask.resize(max_bases / 32 + 2);
        ps.P = out.packed.data();
        ps.M = out.invmask.data();
    } else {
        out.bases.resize(max_bases);
        bases = out.bases.data();
    }
    size_t nb = 0;
    out.recs.reserve((b - a) / 192 + 16);
    out.offsets.reserve((b - a) / 192 + 17);
    size_t p = a;
    while (p < b) {
        if (d[p] == '\n') {  // blank line
            ++p;
            continue;
        }
        Rec r;
        p = parse_mapped_record<AVX2, PACKED ? 2 : 1>(d, p, b, fastq, bases, nb, r, 0, &ps);
        out.recs.push_back(r);
        out.offsets.push_back(nb);
        if (PACKED && ps.multi) return false;
    }
    if (PACKED) {
        size_t groups = ps.finish();
        if (groups == 0) ps.P[0] = ps.P[1] = 0, ps.M[0] = 0, groups = 1;  // (only empty sequences: the arrays still have to exist)
        out.packed.resize(2 * groups);
        out.invmask.resize(groups);
        out.seq_in_chars = true;
    } else {
        out.bases.resize(nb);
    }
    return true;
}

void parse_mapped_chunk(const char *d, size_t a, size_t b, bool fastq, Batch &out, bool packed = false) {
    if (packed && cli_can_pack()) {
        if (parse_mapped_chunk_impl<true, true>(d, a, b, fastq, out)) return;
        out.recs.clear();
        out.offsets.assign(1, 0);
        out.packed.clear();
        out.invmask.clear();
    }
    if (cli_avx2()) parse_mapped_chunk_impl<true, false>(d, a, b, fastq, out);
    else parse_mapped_chunk_impl<false, false>(d, a, b, fastq, out);
}

// ---- two mapped files, mates in step (the parallel form of the paired reader) ------------------------------------------
// records in [a, b), walked like the parser walks them but without touching the sequences
size_t count_mapped_records(const char *d, size_t a, size_t b, bool fastq) {
    size_t n = 0, nb = 0;
    Rec r;
    for (size_t p = a; p < b;) {
        if (d[p] == '\n') {
            ++p;
            continue;
        }
        p = cli_avx2() ? parse_mapped_record<true, 0>(d, p, b, fastq, nullptr, nb, r, 0)
                       : parse_mapped_record<false, 0>(d, p, b, fastq, nullptr, nb, r, 0);
        ++n;
    }
    return n;
}

// position behind the first `skip` records at or after a (a is a record start or a blank line)
size_t skip_mapped_records(const char *d, size_t a, size_t size, bool fastq, size_t skip, size_t *left = nullptr) {
    size_t nb = 0, p = a;
    Rec r;
    while (skip && p < size) {
        if (d[p] == '\n') {
            ++p;
            continue;
        }
        p = cli_avx2() ? parse_mapped_record<true, 0>(d, p, size, fastq, nullptr, nb, r, 0)
                       : parse_mapped_record<false, 0>(d, p, size, fastq, nullptr, nb, r, 0);
        --skip;
    }
    if (left) *left = skip;  // records still wanted when [a, size) ran out
    return p;
}

// mates of [a1, b1) in file 1 and of [a2, b2) in file 2 (the same number of records), interleaved into one batch
// (mate 1, mate 2, ...).  Offsets into the mappings are relative to the lower of the two base pointers.
template <bool AVX2, bool PACKED>
bool parse_mapped_pairs_impl(const char *d1, size_t a1, size_t b1, const char *d2, size_t a2, size_t b2, bool fastq, Batch &out) {
    const char *base = d1 < d2 ? d1 : d2;
    const uint64_t sh1 = (uint64_t)(d1 - base), sh2 = (uint64_t)(d2 - base);
    out.ext = base;
    out.paired = true;
    const size_t max_bases = ((b1 - a1) + (b2 - a2)) / (fastq ? 2 : 1) + 128;  // a sequence is at most (half of) its record's bytes
    PackSink ps;
    uint8_t *bases = nullptr;
    if (PACKED) {
        out.bases.clear();
        out.packed.resize(2 * (max_bases / 32 + 2));
        out.invmask.resize(max_bases / 32 + 2);
        ps.P = out.packed.data();
        ps.M = out.invmask.data();
    } else {
        out.bases.resize(max_bases);
        bases = out.bases.data();
    }
    size_t nb = 0;
    out.recs.reserve((b1 - a1) / 96 + 32);
    out.offsets.reserve((b1 - a1) / 96 + 33);
    out.unit_id.reserve((b1 - a1) / 96 + 32);
    size_t p1 = a1, p2 = a2;
    uint32_t unit = 0;
    for (;;) {
        while (p1 < b1 && d1[p1] == '\n') ++p1;
        while (p2 < b2 && d2[p2] == '\n') ++p2;
        if (p1 >= b1 || p2 >= b2) break;
        Rec r;
        p1 = parse_mapped_record<AVX2, PACKED ? 2 : 1>(d1, p1, b1, fastq, bases, nb, r, sh1, &ps);
        out.recs.push_back(r);
        out.offsets.push_back(nb);
        p2 = parse_mapped_record<AVX2, PACKED ? 2 : 1>(d2, p2, b2, fastq, bases, nb, r, sh2, &ps);
        out.recs.push_back(r);
        out.offsets.push_back(nb);
        out.unit_id.push_back(unit);
        out.unit_id.push_back(unit);
        ++unit;
        if (PACKED && ps.multi) return false;
    }
    if (p1 < b1 || p2 < b2) die("internal: the two inputs' chunks do not hold the same number of records");
    if (PACKED) {
        size_t groups = ps.finish();
        if (groups == 0) ps.P[0] = ps.P[1] = 0, ps.M[0] = 0, groups = 1;  // (only empty sequences: the arrays still have to exist)
        out.packed.resize(2 * groups);
        out.invmask.resize(groups);
        out.seq_in_chars = true;
    } else {
        out.bases.resize(nb);
    }
    return true;
}

void parse_mapped_pairs(const char *d1, size_t a1, size_t b1, const char *d2, size_t a2, size_t b2, bool fastq, Batch &out, bool packed = false) {
    if (packed && cli_can_pack()) {
        if (parse_mapped_pairs_impl<true, true>(d1, a1, b1, d2, a2, b2, fastq, out)) return;
        out.recs.clear();
        out.offsets.assign(1, 0);
        out.unit_id.clear();
        out.packed.clear();
        out.invmask.clear();
    }
    if (cli_avx2()) parse_mapped_pairs_impl<true, false>(d1, a1, b1, d2, a2, b2, fastq, out);
    else parse_mapped_pairs_impl<false, false>(d1, a1, b1, d2, a2, b2, fastq, out);
}

// fn(i) for i in [0, n) on `threads` threads
template <typename F>
void parallel_for(size_t n, size_t threads, F fn) {
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (size_t t = 0; t < std::min(threads, n); ++t)
        pool.emplace_back([&] {
            for (size_t i; (i = next.fetch_add(1)) < n;) fn(i);
        });
    for (auto &t : pool) t.join();
}

struct BatchStats {
    uint64_t total_seqs = 0, filtered_seqs = 0, total_bp = 0, output_bp = 0, filtered_bp = 0, kept_records = 0;
};

// format_record_to_buffer (src/local_filter.rs:60-92) for every kept record of the batch; mate 2 goes to out2 when
// split_mates; rename_base = records written before this batch (global numbering, see the file header)
BatchStats format_batch(Batch &b, bool rename, bool split_mates, uint64_t rename_base) {
    BatchStats st;
    b.out1.clear();
    b.out2.clear();
    const char *chars = b.chars();
    size_t per_unit = b.paired ? 2 : 1;
    uint64_t counter = rename_base;
    for (size_t i = 0; i < b.recs.size(); ++i) {
        const Rec &r = b.recs[i];
        size_t u = i / per_unit;
        st.total_seqs++;
        st.total_bp += r.seq_len;
        if (!b.keep[u]) {
            st.filtered_seqs++;
            st.filtered_bp += r.seq_len;
            continue;
        }
        st.output_bp += r.seq_len;
        st.kept_records++;
        counter++;
        std::vector<char> &dst = (split_mates && (i & 1)) ? b.out2 : b.out1;
        bool fasta = r.qual_off == NO_QUAL;
        dst.push_back(fasta ? '>' : '@');
        if (rename) {
            char num[24];
            int n = std::snprintf(num, sizeof num, "%llu", (unsigned long long)counter);
            dst.insert(dst.end(), num, num + n);
        } else {
            dst.insert(dst.end(), chars + r.id_off, chars + r.id_off + r.id_len);
        }
        dst.push_back('\n');
        dst.insert(dst.end(), b.seq_ptr(r), b.seq_ptr(r) + r.seq_len);
        if (fasta) {
            dst.push_back('\n');
        } else {
            dst.insert(dst.end(), {'\n', '+', '\n'});
            dst.insert(dst.end(), chars + r.qual_off, chars + r.qual_off + r.seq_len);
            dst.push_back('\n');
        }
    }
    return st;
}

inline unsigned decimal_digits(uint64_t v) {
    unsigned n = 1;
    while (v >= 10) {
        v /= 10;
        ++n;
    }
    return n;
}

// bytes format_batch would produce for the kept records of a batch; mate >= 0: only for that mate of every pair
// (two output files, -o / -O), numbered as in the single stream
uint64_t formatted_size(const Batch &b, bool rename, uint64_t rename_base, int mate = -1) {
    uint64_t n = 0, counter = rename_base;
    const size_t per_unit = b.paired ? 2 : 1;
    for (size_t i = 0; i < b.recs.size(); ++i) {
        if (!b.keep[i / per_unit]) continue;
        const Rec &r = b.recs[i];
        ++counter;
        if (mate >= 0 && (int)(i & 1) != mate) continue;
        if (r.rec_len && !rename) n += r.rec_len;
        else n += 1 + (rename ? decimal_digits(counter) : r.id_len) + 1 + r.seq_len + (r.qual_off == NO_QUAL ? 1 : 4 + (uint64_t)r.seq_len);
    }
    return n;
}

// the same records written straight to `dst` (the output file's mapping at this batch's offset): one memcpy per
// run of adjacent kept records whose input bytes can be taken as they are, field by field otherwise
BatchStats format_batch_mapped(const Batch &b, bool rename, uint64_t rename_base, char *dst, uint64_t expect, int mate = -1) {
    BatchStats st;
    const char *chars = b.chars();
    const size_t per_unit = b.paired ? 2 : 1;
    uint64_t counter = rename_base;
    char *o = dst;
    // (MADV_POPULATE_WRITE over the batch's range before copying was measured slower than plain first-touch faults:
    // 0.96 vs 0.80 s for 2.5 GB on tmpfs, profiles/r02_cli_bench.txt)
    uint64_t run_off = 0, run_len = 0;  // pending verbatim run in the input mapping
    auto flush_run = [&] {
        if (run_len) std::memcpy(o, chars + run_off, run_len), o += run_len, run_len = 0;
    };
    for (size_t i = 0; i < b.recs.size(); ++i) {
        const Rec &r = b.recs[i];
        st.total_seqs++;
        st.total_bp += r.seq_len;
        if (!b.keep[i / per_unit]) {
            st.filtered_seqs++;
            st.filtered_bp += r.seq_len;
            continue;
        }
        st.output_bp += r.seq_len;
        st.kept_records++;
        counter++;
        if (mate >= 0 && (int)(i & 1) != mate) continue;  // the other file's mate (the statistics count both: take one call's)
        if (r.rec_len && !rename) {
            if (run_len && run_off + run_len == r.rec_off) run_len += r.rec_len;
            else flush_run(), run_off = r.rec_off, run_len = r.rec_len;
            continue;
        }
        flush_run();
        const bool fasta = r.qual_off == NO_QUAL;
        *o++ = fasta ? '>' : '@';
        if (rename) o += std::snprintf(o, 24, "%llu", (unsigned long long)counter);  // the '\0' is overwritten below
        else std::memcpy(o, chars + r.id_off, r.id_len), o += r.id_len;
        *o++ = '\n';
        std::memcpy(o, b.seq_ptr(r), r.seq_len), o += r.seq_len;
        if (fasta) {
            *o++ = '\n';
        } else {
            std::memcpy(o, "\n+\n", 3), o += 3;
            std::memcpy(o, chars + r.qual_off, r.seq_len), o += r.seq_len;
            *o++ = '\n';
        }
    }
    flush_run();
    if ((uint64_t)(o - dst) != expect) die("internal error: formatted size mismatch");
    return st;
}

// Gather form for a plain output stream: nothing is copied for records whose input bytes already are their output
// (adjacent ones coalesce into one range of the mapped input); the rest is formatted into b.out1 and referenced from
// there.  The writer hands the list to writev(2): one copy, made by the kernel.
BatchStats format_batch_gather(Batch &b, bool rename, uint64_t rename_base) {
    BatchStats st;
    const char *chars = b.chars();
    const size_t per_unit = b.paired ? 2 : 1;
    uint64_t counter = rename_base;
    b.iov1.clear();
    b.out1.clear();
    // formatted pieces are appended to out1; their iovecs hold OFFSETS until out1 stops growing
    std::vector<std::pair<size_t, size_t>> fixups;  // (iovec index, offset in out1)
    bool last_formatted = false;                    // the last iovec is a formatted piece (extendable in out1)
    uint64_t need = 0;
    for (size_t i = 0; i < b.recs.size(); ++i)
        if (b.keep[i / per_unit] && !(b.recs[i].rec_len && !rename)) need += 2 * (uint64_t)b.recs[i].seq_len + b.recs[i].id_len + 32;
    b.out1.reserve(need);
    for (size_t i = 0; i < b.recs.size(); ++i) {
        const Rec &r = b.recs[i];
        st.total_seqs++;
        st.total_bp += r.seq_len;
        if (!b.keep[i / per_unit]) {
            st.filtered_seqs++;
            st.filtered_bp += r.seq_len;
            continue;
        }
        st.output_bp += r.seq_len;
        st.kept_records++;
        counter++;
        if (r.rec_len && !rename) {
            const char *p = chars + r.rec_off;
            if (!b.iov1.empty() && !last_formatted && (const char *)b.iov1.back().iov_base + b.iov1.back().iov_len == p)
                b.iov1.back().iov_len += r.rec_len;
            else b.iov1.push_back({(void *)p, r.rec_len});
            last_formatted = false;
            continue;
        }
        const size_t at = b.out1.size();
        const bool fasta = r.qual_off == NO_QUAL;
        b.out1.push_back(fasta ? '>' : '@');
        if (rename) {
            char num[24];
            int n = std::snprintf(num, sizeof num, "%llu", (unsigned long long)counter);
            b.out1.insert(b.out1.end(), num, num + n);
        } else {
            b.out1.insert(b.out1.end(), chars + r.id_off, chars + r.id_off + r.id_len);
        }
        b.out1.push_back('\n');
        b.out1.insert(b.out1.end(), b.seq_ptr(r), b.seq_ptr(r) + r.seq_len);
        if (fasta) {
            b.out1.push_back('\n');
        } else {
            b.out1.insert(b.out1.end(), {'\n', '+', '\n'});
            b.out1.insert(b.out1.end(), chars + r.qual_off, chars + r.qual_off + r.seq_len);
            b.out1.push_back('\n');
        }
        if (last_formatted) {
            b.iov1.back().iov_len += b.out1.size() - at;  // extends the previous formatted piece
        } else {
            fixups.emplace_back(b.iov1.size(), at);
            b.iov1.push_back({nullptr, b.out1.size() - at});
        }
        last_formatted = true;
    }
    for (auto &f : fixups) b.iov1[f.first].iov_base = b.out1.data() + f.second;
    return st;
}

// Output file written through a shared mapping: the formatter threads copy kept records to their final place in
// parallel (a single write(2) stream moves ~6 GB/s and was the last serial stage).  The file is first sized to an
// upper bound (sparse), and cut to the bytes really written at the end.
class MappedOutput {
  public:
    bool open(const std::string &path, uint64_t reserve) {
        fd_ = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd_ < 0) die("Failed to create output file: " + path);
        struct stat st;
        // Only filesystems that report a failed allocation when the page is touched (SIGBUS, handled below) are written
        // through a mapping.  Anything else -- NFS, FUSE, CIFS, a quota enforced at writeback -- would let the run print
        // "Retained ..." and exit 0 over a short file; there the gather writer's write(2) reports the error as the
        // reference's writer does.  DCN_CLI_MMAP_ANY_FS=1 maps regardless and pays for an msync at the end instead.
        struct statfs sfs;
        const bool local = fstatfs(fd_, &sfs) == 0 &&
                           (sfs.f_type == 0x01021994 /* tmpfs */ || sfs.f_type == 0xEF53 /* ext2/3/4 */ ||
                            sfs.f_type == 0x58465342 /* xfs */ || sfs.f_type == 0x9123683E /* btrfs */ ||
                            sfs.f_type == 0x794C7630 /* overlayfs */ || sfs.f_type == 0xF2F52010 /* f2fs */ ||
                            sfs.f_type == 0x2FC12FC1 /* zfs */ || sfs.f_type == 0x858458F6 /* ramfs */);
        sync_at_end_ = !local;
        if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode) || (!local && !std::getenv("DCN_CLI_MMAP_ANY_FS")) ||
            ftruncate(fd_, (off_t)reserve) != 0) {  // a pipe, a device, a remote filesystem ...
            ::close(fd_);
            fd_ = -1;
            return false;
        }
        void *p = mmap(nullptr, reserve, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
        if (p == MAP_FAILED) {
            if (ftruncate(fd_, 0) != 0) die("write error");
            ::close(fd_);
            fd_ = -1;
            return false;
        }
        data_ = (char *)p;
        reserve_ = reserve;
        slot_ = g_sparse_out_fds[0].load() < 0 ? 0 : 1;
        g_sparse_out_fds[slot_] = fd_;
#ifdef MADV_HUGEPAGE
        (void)madvise(p, reserve, MADV_HUGEPAGE);  // where the filesystem honours it: 512x fewer first-touch faults
#endif
        // a full disk shows up as SIGBUS on a store into the mapping, not as an error code: report it like any other
        // write error instead of dying silently
        struct sigaction sa;
        std::memset(&sa, 0, sizeof sa);
        sa.sa_handler = [](int) {
            static const char msg[] = "Error: write error (output file could not grow)\n";
            ssize_t ignored = ::write(2, msg, sizeof msg - 1);
            (void)ignored;
            cut_sparse_outputs();
            _exit(1);
        };
        sigaction(SIGBUS, &sa, nullptr);
        // an interrupted run must not leave the sparse reservation (several times the input's size) behind
        struct sigaction si;
        std::memset(&si, 0, sizeof si);
        si.sa_handler = [](int sig) {
            cut_sparse_outputs();
            _exit(128 + sig);
        };
        for (int sig : {SIGINT, SIGTERM, SIGHUP}) sigaction(sig, &si, nullptr);
        return true;
    }
    char *at(uint64_t off, uint64_t len) {
        if (off + len > reserve_) die("internal error: output larger than its reservation");
        return data_ + off;
    }
    void finish(uint64_t bytes) {
        if (fd_ < 0) return;
        // a filesystem that defers allocation errors to writeback only shows them to msync / fsync
        if (sync_at_end_ && bytes && msync(data_, bytes, MS_SYNC) != 0) die("write error");
        g_sparse_out_fds[slot_] = -1;
        if (munmap(data_, reserve_) != 0 || ftruncate(fd_, (off_t)bytes) != 0) die("write error");
        if (sync_at_end_ && fdatasync(fd_) != 0) die("write error");
        if (::close(fd_) != 0) die("write error");
        fd_ = -1;
    }
    bool active() const { return fd_ >= 0; }

  private:
    int fd_ = -1;
    char *data_ = nullptr;
    uint64_t reserve_ = 0;
    int slot_ = 0;
    bool sync_at_end_ = false;
};

std::string fmt_duration(double s) {  // like Rust's {:.2?} for Duration
    char b[64];
    if (s >= 1.0) std::snprintf(b, sizeof b, "%.2fs", s);
    else if (s >= 1e-3) std::snprintf(b, sizeof b, "%.2fms", s * 1e3);
    else std::snprintf(b, sizeof b, "%.2fµs", s * 1e6);
    return b;
}

std::string json_str(const std::string &s) {
    std::string o = "\"";
    for (char c : s) {
        if (c == '"' || c == '\\') o += '\\';
        o += c;
    }
    return o + "\"";
}

struct FilterArgs {
    std::string index, input = "-", output = "-";
    std::string input2, output2, summary;
    bool has_input2 = false, has_output2 = false, has_summary = false;
    unsigned abs_threshold = 2;
    double rel_threshold = 0.01;
    size_t prefix_length = 0;
    bool deplete = false, rename = false, debug = false, quiet = false;
    size_t threads = 8;          // the reference's default (src/main.rs:68) ...
    bool threads_given = false;  // ... which, when -t is not given, grows to three quarters of the CPUs the process may use
    int compression_level = 2;
    std::vector<int> devices{0};  // --gpus N / --devices a,b,...: one pipeline context per entry (repeats allowed)
};

// --debug lines of process_record / process_record_pair (src/local_filter.rs:354-363, 424-434).  Single reads list
// the k-mer under every first occurrence of a hit hash, in read order (sequence_matches, src/filter_common.rs:
// 143-153): positions come from dcn_minimizer_hashes_batch, membership from dcn_index_contains.  Pairs print
// "id1/id2" only when something hit, and their k-mer list is always empty in the reference, because
// get_paired_minimizer_hashes_and_positions never fills the sequence list pair_matches cuts from
// (src/filter_common.rs:326-345 extends it by hashes.len() - positions.len() = 0 copies).
void debug_lines(deacon::FilterProcessor &proc, const deacon::Index &index, const Batch &b, size_t n_units,
                 size_t prefix_length) {
    const char *chars = b.chars();
    if (b.paired) {
        for (size_t u = 0; u < n_units; ++u) {
            if (b.hits[u] == 0) continue;
            const Rec &r1 = b.recs[2 * u], &r2 = b.recs[2 * u + 1];
            std::fprintf(stderr, "DEBUG: %.*s/%.*s hits=%u/%u keep=%s kmers=[]\n", (int)r1.id_len, chars + r1.id_off,
                         (int)r2.id_len, chars + r2.id_off, b.hits[u], b.total[u], b.keep[u] ? "true" : "false");
        }
        return;
    }
    const uint32_t n_reads = (uint32_t)b.recs.size();
    const unsigned k = index.header().kmer_length, w = index.header().window_size;
    std::string kmers;
    std::vector<uint64_t> seen;
    auto line = [&](uint32_t i, const uint64_t *hs, const bool *in_set, size_t n, auto pos_of) {
        const Rec &r = b.recs[i];
        kmers.clear();
        seen.clear();
        for (size_t j = 0; j < n; ++j) {
            if (!in_set[j] || std::find(seen.begin(), seen.end(), hs[j]) != seen.end()) continue;
            seen.push_back(hs[j]);
            if (!kmers.empty()) kmers += ',';
            kmers.append(reinterpret_cast<const char *>(b.bases.data()) + r.seq_off + pos_of(j), k);
        }
        std::fprintf(stderr, "DEBUG: %.*s hits=%u/%u keep=%s kmers=[%s]\n", (int)r.id_len, chars + r.id_off, b.hits[i],
                     b.total[i], b.keep[i] ? "true" : "false", kmers.c_str());
    };
    // calls that fit the context; a read longer than any call goes piece by piece (deacon::append_minimizer_hashes_any_length)
    const uint64_t cap_bases = proc.config().max_batch_bases, cap_reads = proc.config().max_batch_reads;
    std::vector<uint64_t> off, hashes, sub, pos64;
    std::vector<uint32_t> pos;
    std::unique_ptr<bool[]> member;
    auto members_of = [&](const std::vector<uint64_t> &hs) {
        std::vector<bool> m = index.contains(hs);
        member.reset(new bool[std::max<size_t>(m.size(), 1)]);
        for (size_t j = 0; j < m.size(); ++j) member[j] = m[j];
    };
    for (uint32_t r0 = 0; r0 < n_reads;) {
        uint32_t r1 = r0;
        while (r1 < n_reads && r1 - r0 < cap_reads && b.offsets[r1 + 1] - b.offsets[r0] <= cap_bases) ++r1;
        if (r1 == r0) {
            hashes.clear();
            pos64.clear();
            const Rec &r = b.recs[r0];
            deacon::append_minimizer_hashes_any_length(proc, b.bases.data() + r.seq_off, r.seq_len, k, w, prefix_length,
                                                       std::max<uint64_t>(cap_bases / 2, 64), hashes, &pos64);
            members_of(hashes);
            line(r0, hashes.data(), member.get(), hashes.size(), [&](size_t j) { return pos64[j]; });
            ++r0;
            continue;
        }
        const uint64_t nb = b.offsets[r1] - b.offsets[r0];
        sub.resize(r1 - r0 + 1);
        for (uint32_t r = r0; r <= r1; ++r) sub[r - r0] = b.offsets[r] - b.offsets[r0];
        off.assign(r1 - r0 + 1, 0);
        hashes.resize(nb + 1);
        pos.resize(nb + 1);
        deacon::check(dcn_minimizer_hashes_batch(proc.raw(), b.bases.data() + b.offsets[r0], sub.data(), r1 - r0, prefix_length, off.data(),
                                                 hashes.data(), pos.data(), hashes.size()));
        hashes.resize(off[r1 - r0]);
        members_of(hashes);
        for (uint32_t i = r0; i < r1; ++i) {
            const uint64_t a0 = off[i - r0];
            line(i, hashes.data() + a0, member.get() + a0, off[i - r0 + 1] - a0, [&](size_t j) { return (uint64_t)pos[a0 + j]; });
        }
        r0 = r1;
    }
}

// optional stage accounting (DCN_CLI_TIMING=1): busy seconds summed over the threads of each stage
struct StageClock {
    std::atomic<uint64_t> ns{0};
    struct Scope {
        StageClock &c;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        explicit Scope(StageClock &c_) : c(c_) {}
        ~Scope() { c.ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
    };
    double seconds() const { return ns.load() * 1e-9; }
};

// CPUs this process may really use: the affinity mask capped by the cgroup's CPU quota (a container often sees every
// hardware thread of its host and is throttled to a share of them)
size_t usable_cpus() {
    size_t n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<size_t>(n, (size_t)std::max(1, CPU_COUNT(&set)));
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
            n = std::min<size_t>(n, (size_t)std::max<long long>(1, quota / period));
        std::fclose(f);
    }
    return n;
}

// ---- deacon filter (src/local_filter.rs:575-824) ---------------------------------------------------------------
int run_filter(const FilterArgs &a_in) {
    // -t counts the reference's filter workers (default 8); here the filtering is the GPU's and the host threads parse and
    // format around it, a job that profits from more of them up to about three quarters of the CPUs (the library's packers
    // and the runtime want the rest: -t 12 beat -t 8 and -t 16 on a 16-CPU share, profiles/r03 cli notes).  An explicit -t
    // is taken as given.
    FilterArgs a_eff = a_in;
    if (!a_eff.threads_given) a_eff.threads = std::max<size_t>(a_eff.threads, std::min<size_t>(usable_cpus() * 3 / 4, 32));
    const FilterArgs &a = a_eff;
    StageClock t_parse, t_gpu, t_gpu_wait, t_format, t_write, t_push_wait;
    double m_index = 0, m_ctx = 0, m_feeder_done = 0, m_gpu_done = 0, m_written = 0;  // milestones, seconds since start
    const bool cli_timing = std::getenv("DCN_CLI_TIMING") != nullptr;
    using clock = std::chrono::steady_clock;
    auto start = clock::now();
    bool quiet = a.quiet || a.debug;  // :581
    bool paired_stdin = a.input == "-" && a.has_input2 && a.input2 == "-";
    bool paired = a.has_input2;
    if (!quiet) {
        std::string opts = "abs_threshold=" + std::to_string(a.abs_threshold) + ", rel_threshold=" + std::to_string(a.rel_threshold);
        // Rust prints the shortest round-trip form of the f64; trim trailing zeros of the fixed form
        {
            char b[64];
            std::snprintf(b, sizeof b, "%.17g", a.rel_threshold);
            std::string shortest = b;
            for (int prec = 1; prec < 17; ++prec) {
                std::snprintf(b, sizeof b, "%.*g", prec, a.rel_threshold);
                if (std::strtod(b, nullptr) == a.rel_threshold) {
                    shortest = b;
                    break;
                }
            }
            opts = "abs_threshold=" + std::to_string(a.abs_threshold) + ", rel_threshold=" + shortest;
        }
        if (a.prefix_length > 0) opts += ", prefix_length=" + std::to_string(a.prefix_length);
        if (a.rename) opts += ", rename";
        if (a.threads > 0) opts += ", threads=" + std::to_string(a.threads);
        std::fprintf(stderr, "Deacon-hip v%s; mode: %s; input: %s; options: %s\n", VERSION, a.deplete ? "deplete" : "search",
                     paired_stdin ? "interleaved" : paired ? "paired" : "single", opts.c_str());
    }
    // plain regular single input: mmap + parallel parsing of record-aligned chunks (see stage 1)
    MappedFile mapped, mapped2;
    bool parallel_in = !paired && a.input != "-" && mapped.open(a.input);
    // (a file that begins with a blank line is read by the stream readers, which skip it as the reference's reader does)
    if (parallel_in && mapped.size > 0 && (mapped.data[0] == '\n' || mapped.data[0] == '\r')) parallel_in = false;
    // two plain regular files of mates: both mapped, cut at the same record numbers, parsed in parallel as well
    const bool pair_in = paired && !paired_stdin && a.input != "-" && a.input2 != "-" && !std::getenv("DCN_CLI_NO_PAIR_MMAP") &&
                         mapped.open(a.input) && mapped2.open(a.input2) && mapped.size > 0 && mapped2.size > 0 &&
                         (mapped.data[0] == '@' || mapped.data[0] == '>') && mapped.data[0] == mapped2.data[0];
    // ... and if the output is a plain file too, the formatter threads write it through a shared mapping
    MappedOutput mapped_out;
    const bool plain_out = a.output != "-" && !ends_with(a.output, ".gz") && !ends_with(a.output, ".zst") && !ends_with(a.output, ".xz");
    // (one write(2) / writev(2) stream moves 4.8-6.6 GB/s on tmpfs and stalls the stages in front of it: 0.90-0.95 s
    // against 0.68-0.76 s through the mapping for the same 5 GB input, profiles/r02_cli_bench.txt)
    // two plain files of mates written to two plain files (-o / -O, the usual shape of a paired run): each output is mapped
    // and takes its own mate of every kept pair
    MappedOutput mapped_out2;
    const bool plain_out2 = a.has_output2 && a.output2 != "-" && !ends_with(a.output2, ".gz") && !ends_with(a.output2, ".zst") && !ends_with(a.output2, ".xz");
    const bool split_mapped = pair_in && a.has_output2 && plain_out && plain_out2 && a.output2 != a.output;
    if ((parallel_in || (pair_in && (!a.has_output2 || split_mapped))) && plain_out && !std::getenv("DCN_CLI_NO_MMAP_OUT")) {
        const uint64_t in1 = (uint64_t)mapped.size, in2 = pair_in ? (uint64_t)mapped2.size : 0;
        // >= any formatted size (renamed ids: <= 20 digits)
        if (mapped_out.open(a.output, 5 * (split_mapped ? in1 : in1 + in2) + (1u << 20)) && split_mapped &&
            !mapped_out2.open(a.output2, 5 * in2 + (1u << 20)))
            mapped_out.finish(0);  // both files through mappings, or both through write(2) (the file is re-created below)
    }
    const bool map_out = mapped_out.active();
    const bool map_out2 = map_out && mapped_out2.active();
    std::unique_ptr<Output> out1_holder;
    if (!map_out) out1_holder.reset(new Output(a.output, a.compression_level));
    // plain single-file output: the formatter only lists what to write, the writer thread gathers it with writev
    const bool gather_out = !map_out && out1_holder->plain() && !(a.has_output2 && paired) && !std::getenv("DCN_CLI_NO_GATHER");
    std::unique_ptr<Output> out2;
    if (a.has_output2 && paired) {
        if (!map_out2) out2.reset(new Output(a.output2, a.compression_level));
    } else if (a.has_output2 && !quiet) std::fprintf(stderr, "Warning: --output2 specified but no second input file provided. --output2 will be ignored.\n");

    const uint64_t batch_bases = 1ull << 26;
    const uint32_t batch_reads = 1u << 20;
    deacon::FilterConfig cfg;
    cfg.abs_threshold = a.abs_threshold;
    cfg.rel_threshold = a.rel_threshold;
    cfg.prefix_length = a.prefix_length;
    cfg.deplete = a.deplete;
    cfg.max_batch_bases = batch_bases + (1 << 24);
    cfg.max_batch_reads = batch_reads + 2;
    if (const char *e = std::getenv("DCN_CLI_MAX_BATCH_READS"))  // test hook: force batches to be cut into several calls
        cfg.max_batch_reads = (uint32_t)std::max(2, std::atoi(e));
    if (const char *e = std::getenv("DCN_CLI_MAX_BATCH_BASES"))  // test hook: records longer than a call at test sizes
        cfg.max_batch_bases = (uint64_t)std::max(1024, std::atoi(e));

    // ---- stage 1: parsed batches, in input order --------------------------------------------------------------
    // plain regular file, single input: mmap + parallel parsing of record-aligned chunks; otherwise (stdin, gzip,
    // paired) one streaming reader thread
    size_t n_workers = a.threads ? a.threads : usable_cpus();  // -t 0: every CPU this process may really use
    n_workers = std::min<size_t>(std::max<size_t>(n_workers, 1), 64);
    bool fastq_in = parallel_in && mapped.data[0] == '@';
    if (parallel_in && mapped.data[0] != '@' && mapped.data[0] != '>') die("Invalid FASTX record start: expected '>' or '@'");
    // one stream that is not a mappable plain file (stdin, gzip / zstd / xz): the chunk reader below
    // (... or interleaved mates on stdin: the same reader, chunks of an even number of records)
    const bool chunk_in = (!paired || paired_stdin) && !parallel_in && !std::getenv("DCN_CLI_NO_CHUNK_READER");
    // two streams of mates of which at least one is not a mappable plain file (.fastq.gz pairs: the usual shape of a short-read
    // run): two chunk readers in step, see below
    const bool pair_chunk_in = paired && !paired_stdin && !pair_in && a.input != "-" && a.input2 != "-" && !std::getenv("DCN_CLI_NO_CHUNK_READER");
    const bool pool_in = parallel_in || chunk_in || pair_in || pair_chunk_in;  // batches come out of the parser pool
    // the chunk parsers write the batch stream 2-bit packed (no ASCII copy of the bases, no second pass of the library's
    // host threads over it); --debug prints k-mer strings and wants the ASCII
    const bool packed_parse = pool_in && cli_can_pack() && !a.debug;
    BatchPool pool;
    Queue<std::unique_ptr<Batch>> parsed(4);
    std::unique_ptr<OrderedStage> parse_stage;
    std::thread reader, reader2;
    Queue<std::unique_ptr<Batch>> half_read(4);  // (pair_chunk_in: batches that hold file 1's chunk and wait for file 2's)
    if (parallel_in) {
        const char *d = mapped.data;
        size_t size = mapped.size;
        // (enough chunks in flight, ~12 MB each, to keep the parsers busy while the GPU runtime starts and the index loads)
        parse_stage.reset(new OrderedStage(n_workers, 4 * n_workers + 8, [d, fastq_in, packed_parse, &t_parse](Batch &b) {
            StageClock::Scope sc(t_parse);
            size_t ca = (size_t)b.offsets[0], cb = (size_t)b.seq_no;  // chunk bounds travel in the empty batch
            b.offsets.assign(1, 0);
            parse_mapped_chunk(d, ca, cb, fastq_in, b, packed_parse);
        }));
        reader = std::thread([&, d, size] {
            // chunks stay well under glibc's 32 MB mmap threshold: the per-batch vectors are then recycled by malloc
            // instead of being mapped, page-faulted and unmapped every time (64 MB chunks ran 1.7x slower for that
            // reason); small files still spread over the workers
            size_t chunk_cap = 24u << 20;
            if (const char *e = std::getenv("DCN_CLI_CHUNK_MB")) chunk_cap = (size_t)std::max(1, std::atoi(e)) << 20;  // tuning hook
            const size_t chunk = std::min<size_t>(std::max<size_t>(size / (4 * n_workers), 4u << 20), chunk_cap);
            size_t pos = 0;
            while (pos < size) {
                size_t end = pos + chunk >= size ? size : next_record_start(d, size, pos + chunk, fastq_in);
                std::unique_ptr<Batch> b = pool.get();
                b->offsets[0] = pos;  // see the worker lambda
                b->seq_no = end;
                parse_stage->push(std::move(b));
                pos = end;
            }
            parse_stage->finish();
        });
    } else if (pair_in) {
        const char *d1 = mapped.data, *d2 = mapped2.data;
        const size_t size1 = mapped.size, size2 = mapped2.size;
        const bool fq = d1[0] == '@';
        parse_stage.reset(new OrderedStage(n_workers, 4 * n_workers + 8, [d1, d2, fq, packed_parse, &t_parse](Batch &b) {
            StageClock::Scope sc(t_parse);
            // chunk bounds travel in the empty batch: file 1 in offsets[0] / seq_no, file 2 in out_off / out_bytes
            const size_t a1 = (size_t)b.offsets[0], b1 = (size_t)b.seq_no, a2 = (size_t)b.out_off, b2 = (size_t)b.out_bytes;
            b.offsets.assign(1, 0);
            b.out_off = b.out_bytes = 0;
            parse_mapped_pairs(d1, a1, b1, d2, a2, b2, fq, b, packed_parse);
        }));
        reader = std::thread([&, d1, d2, size1, size2, fq] {
            // Mates sit at the same record NUMBER of their files, not at the same byte: both files are cut into chunks at
            // record boundaries and the chunks' records counted (a walk without the sequence copies, all workers), which
            // gives every cut of file 1 a record number; the byte in file 2 behind that many records is found by walking
            // from the nearest cut of file 2 (again on all workers).  The pairs of ranges then go to the parser pool.
            size_t chunk_cap = 12u << 20;
            if (const char *e = std::getenv("DCN_CLI_CHUNK_MB")) chunk_cap = (size_t)std::max(1, std::atoi(e)) << 20;  // tuning hook
            auto cuts_of = [&](const char *d, size_t size) {
                const size_t chunk = std::min<size_t>(std::max<size_t>(size / (4 * n_workers), 2u << 20), chunk_cap);
                std::vector<size_t> cuts{0};
                while (cuts.back() < size) cuts.push_back(cuts.back() + chunk >= size ? size : next_record_start(d, size, cuts.back() + chunk, fq));
                return cuts;
            };
            const std::vector<size_t> c1 = cuts_of(d1, size1), c2 = cuts_of(d2, size2);
            std::vector<size_t> n1(c1.size(), 0), n2(c2.size(), 0);  // records before each cut
            parallel_for(c1.size() - 1 + c2.size() - 1, n_workers, [&](size_t i) {
                if (i < c1.size() - 1) n1[i + 1] = count_mapped_records(d1, c1[i], c1[i + 1], fq);
                else n2[i - (c1.size() - 1) + 1] = count_mapped_records(d2, c2[i - (c1.size() - 1)], c2[i - (c1.size() - 1) + 1], fq);
            });
            for (size_t i = 1; i < n1.size(); ++i) n1[i] += n1[i - 1];
            for (size_t i = 1; i < n2.size(); ++i) n2[i] += n2[i - 1];
            if (n1.back() > n2.back()) die("Paired input ended with an unpaired record");
            if (n1.back() < n2.back()) die("Second input has more records than the first");
            std::vector<size_t> at2(c1.size(), 0);  // byte of file 2 behind n1[j] records
            at2.back() = size2;
            parallel_for(c1.size() - 1, n_workers, [&](size_t j) {
                if (j == 0) return;
                const size_t k = (size_t)(std::upper_bound(n2.begin(), n2.end(), n1[j]) - n2.begin()) - 1;  // n2[k] <= n1[j]
                at2[j] = skip_mapped_records(d2, c2[k], size2, fq, n1[j] - n2[k]);
            });
            for (size_t j = 0; j + 1 < c1.size(); ++j) {
                std::unique_ptr<Batch> b = pool.get();
                b->offsets[0] = c1[j];  // see the worker lambda
                b->seq_no = c1[j + 1];
                b->out_off = at2[j];
                b->out_bytes = at2[j + 1];
                parse_stage->push(std::move(b));
            }
            parse_stage->finish();
        });
    } else if (pair_chunk_in) {
        // Two streams of mates, at least one of them compressed (or otherwise not mappable).  The record reader below walks both
        // record by record on one thread (2.5 GB/s of FASTQ for the two together); the streams' own readers deliver 4+ GB/s
        // EACH since the gzip reader runs on several threads.  So: one thread per stream.  The first cuts its stream into chunks
        // of whole records, as the chunk reader of a single stream does, and counts each chunk's records (the parser's walk
        // without the sequence copies); the second takes from its stream exactly that many records for the same batch; the
        // parser pool then parses the two chunks into one batch of interleaved mates (parse_mapped_pairs, the parser of the
        // mapped pair path: its two ranges need not be mappings).
        parse_stage.reset(new OrderedStage(n_workers, 2 * n_workers + 4, [packed_parse, &t_parse](Batch &b) {
            StageClock::Scope sc(t_parse);
            b.offsets.assign(1, 0);
            parse_mapped_pairs(b.text.data(), 0, b.raw_end, b.text2.data(), 0, b.raw_end2, b.raw_fastq, b, packed_parse);
        }));
        size_t chunk = 12u << 20;
        if (const char *e = std::getenv("DCN_CLI_CHUNK_MB")) chunk = (size_t)std::max(1, std::atoi(e)) << 20;  // tuning / test hook
        reader = std::thread([&, chunk] {
            Input in(a.input);
            std::vector<char> carry;
            bool eof = false;
            int fastq = -1;
            while (!eof || !carry.empty()) {
                std::unique_ptr<Batch> b = pool.get();
                std::vector<char> &t = b->text;
                t.resize(std::max(chunk, 2 * carry.size()));
                std::memcpy(t.data(), carry.data(), carry.size());
                size_t n = carry.size();
                carry.clear();
                size_t cut = 0;
                for (;;) {
                    while (n < t.size() && !eof) {
                        const size_t got = in.read(t.data() + n, t.size() - n);
                        if (got == 0) eof = true;
                        n += got;
                    }
                    if (fastq < 0 && n) {
                        size_t f = 0;
                        while (f < n && (t[f] == '\n' || t[f] == '\r')) ++f;
                        if (f < n) {
                            if (t[f] != '@' && t[f] != '>') die("Invalid FASTX record start: expected '>' or '@'");
                            fastq = t[f] == '@';
                        }
                    }
                    cut = eof ? n : last_record_boundary(t.data(), n, fastq > 0);
                    if (cut || eof) break;
                    t.resize(2 * t.size());  // a record longer than the chunk
                }
                if (n == 0) break;
                carry.assign(t.begin() + cut, t.begin() + n);
                b->raw_end = cut;
                b->raw_fastq = fastq > 0;
                b->n_records = count_mapped_records(t.data(), 0, cut, fastq > 0);
                if (b->n_records) half_read.push(std::move(b));
            }
            half_read.finish();
        });
        reader2 = std::thread([&, chunk] {
            Input in(a.input2);
            std::vector<char> carry;
            bool eof = false;
            int fastq = -1;
            auto only_blank = [](const char *p, size_t n) {
                for (size_t i = 0; i < n; ++i)
                    if (p[i] != '\n' && p[i] != '\r') return false;
                return true;
            };
            std::unique_ptr<Batch> b;
            while (half_read.pop(b)) {
                std::vector<char> &t = b->text2;
                // (as many bytes as the first file's chunk is a good guess; what is read beyond the records wanted is carried over)
                t.resize(std::max<size_t>(std::max(b->raw_end + (256u << 10), 2 * carry.size()), 1u << 20));
                std::memcpy(t.data(), carry.data(), carry.size());
                size_t n = carry.size(), p = 0, need = b->n_records;
                carry.clear();
                for (bool first = true;; first = false) {
                    // more input: the whole guess at first, then a megabyte at a time
                    const size_t want = first ? t.size() : std::min<size_t>(t.size(), n + (1u << 20));
                    while (n < want && !eof) {
                        const size_t got = in.read(t.data() + n, want - n);
                        if (got == 0) eof = true;
                        n += got;
                    }
                    if (fastq < 0 && n) {
                        size_t f = 0;
                        while (f < n && (t[f] == '\n' || t[f] == '\r')) ++f;
                        if (f < n) {
                            if (t[f] != '@' && t[f] != '>') die("Invalid FASTX record start: expected '>' or '@'");
                            fastq = t[f] == '@';
                            if ((fastq > 0) != b->raw_fastq) die("The two inputs are not of the same format (FASTA and FASTQ)");
                        }
                    }
                    const size_t whole = eof ? n : (fastq < 0 ? 0 : last_record_boundary(t.data(), n, fastq > 0));
                    if (whole > p) p = skip_mapped_records(t.data(), p, whole, fastq > 0, need, &need);
                    if (need == 0) break;
                    if (eof) die("Paired input ended with an unpaired record");
                    if (n == t.size()) t.resize(t.size() + std::max<size_t>(t.size() / 2, 1u << 20));
                }
                carry.assign(t.begin() + p, t.begin() + n);
                b->raw_end2 = p;
                parse_stage->push(std::move(b));
            }
            // the first file has ended: so must the second
            bool more = !only_blank(carry.data(), carry.size());
            std::vector<char> rest(1u << 16);
            for (size_t got; !more && !eof && (got = in.read(rest.data(), rest.size())) > 0;) more = !only_blank(rest.data(), got);
            if (more) die("Second input has more records than the first");
            parse_stage->finish();
        });
    } else if (chunk_in) {
        // stdin or a compressed file, one stream: this thread only decompresses and cuts the stream into chunks of whole
        // records (each batch keeps its raw chunk: ids and qualities stay in it); the worker pool parses them with the
        // parser of the mapped path.  (One thread doing both ran at the parser's pace: ~1 GB/s against zstd's 1.5+.)
        parse_stage.reset(new OrderedStage(n_workers, 2 * n_workers + 4, [packed_parse, paired_stdin, &t_parse](Batch &b) {
            StageClock::Scope sc(t_parse);
            b.offsets.assign(1, 0);
            parse_mapped_chunk(b.text.data(), 0, b.raw_end, b.raw_fastq, b, packed_parse);
            if (paired_stdin) {  // interleaved mates: records 2u and 2u + 1 are unit u (the reader cut an even number of them)
                b.paired = true;
                b.unit_id.resize(b.recs.size());
                for (size_t i = 0; i < b.recs.size(); ++i) b.unit_id[i] = (uint32_t)(i / 2);
            }
        }));
        reader = std::thread([&] {
            Input in(a.input);
            size_t chunk = 16u << 20;
            if (const char *e = std::getenv("DCN_CLI_CHUNK_MB")) chunk = (size_t)std::max(1, std::atoi(e)) << 20;  // tuning / test hook
            std::vector<char> carry;
            bool eof = false;
            int fastq = -1;
            while (!eof || !carry.empty()) {
                std::unique_ptr<Batch> b = pool.get();
                std::vector<char> &t = b->text;
                t.resize(std::max(chunk, 2 * carry.size()));
                std::memcpy(t.data(), carry.data(), carry.size());
                size_t n = carry.size();
                carry.clear();
                size_t cut = 0;
                for (;;) {
                    while (n < t.size() && !eof) {
                        const size_t got = in.read(t.data() + n, t.size() - n);
                        if (got == 0) eof = true;
                        n += got;
                    }
                    if (fastq < 0 && n) {
                        size_t f = 0;
                        while (f < n && (t[f] == '\n' || t[f] == '\r')) ++f;
                        if (f < n) {
                            if (t[f] != '@' && t[f] != '>') die("Invalid FASTX record start: expected '>' or '@'");
                            fastq = t[f] == '@';
                        }
                    }
                    cut = eof ? n : last_record_boundary(t.data(), n, fastq > 0);
                    if (cut || eof) break;
                    t.resize(2 * t.size());  // a record longer than the chunk
                }
                if (n == 0) break;
                if (paired_stdin) {
                    // mates stay together: an odd record at the end of the chunk waits for its mate in the next one
                    const size_t recs = count_mapped_records(t.data(), 0, cut, fastq > 0);
                    if (recs & 1) {
                        if (eof) die("Paired input ended with an unpaired record");
                        cut = skip_mapped_records(t.data(), 0, cut, fastq > 0, recs - 1);
                    }
                    if (recs < 2 && !eof) {  // (one record longer than the chunk: read on)
                        carry.assign(t.begin(), t.begin() + n);
                        continue;
                    }
                }
                carry.assign(t.begin() + cut, t.begin() + n);
                b->raw_end = cut;
                b->raw_fastq = fastq > 0;
                if (cut) parse_stage->push(std::move(b));
            }
            parse_stage->finish();
        });
    } else {
        reader = std::thread([&] {
            FastxReader r1(a.input);
            std::unique_ptr<FastxReader> r2;
            if (paired && !paired_stdin) r2.reset(new FastxReader(a.input2));
            bool more = true;
            while (more) {
                std::unique_ptr<Batch> b = pool.get();
                b->paired = paired;
                while (b->bases.size() < batch_bases && b->recs.size() < batch_reads) {
                    if (!r1.next(*b)) {
                        more = false;
                        break;
                    }
                    if (paired) {
                        bool ok = paired_stdin ? r1.next(*b) : r2->next(*b);
                        if (!ok) die("Paired input ended with an unpaired record");
                        uint32_t u = (uint32_t)(b->recs.size() / 2 - 1);
                        b->unit_id.push_back(u);
                        b->unit_id.push_back(u);
                    }
                }
                if (!b->recs.empty()) parsed.push(std::move(b));
            }
            if (r2) {
                Batch extra;
                if (r2->next(extra)) die("Second input has more records than the first");
            }
            parsed.finish();
        });
    }
    auto next_parsed = [&](std::unique_ptr<Batch> &b) { return pool_in ? parse_stage->pop(b) : parsed.pop(b); };

    // The parsers are already running: HIP start-up (~0.25 s) and the index load happen behind them.
    std::unique_ptr<deacon::Index> index_holder;
    try {
        index_holder.reset(new deacon::Index(deacon::Index::load(a.index, a.devices[0])));
    } catch (const std::exception &e) {
        die(e.what());
    }
    deacon::Index &index = *index_holder;
    auto hd = index.header();
    if (!quiet)
        std::fprintf(stderr, "Loaded index (k=%u, w=%u) in %s\n", hd.kmer_length, hd.window_size,
                     fmt_duration(std::chrono::duration<double>(clock::now() - start).count()).c_str());
    m_index = std::chrono::duration<double>(clock::now() - start).count();

    // ---- stage 3: format kept records on the pool, write in order -----------------------------------------------
    const bool split_mates = (bool)out2;
    std::vector<BatchStats> stats_by_batch;
    std::mutex stats_m;
    BatchStats tot;
    OrderedStage format_stage(pool_in ? n_workers : 2, 2 * n_workers + 2, [&](Batch &b) {
        StageClock::Scope sc(t_format);
        if (map_out2) format_batch_mapped(b, a.rename, b.seq_no, mapped_out2.at(b.out_off2, b.out_bytes2), b.out_bytes2, 1);
        BatchStats st = map_out ? format_batch_mapped(b, a.rename, b.seq_no, mapped_out.at(b.out_off, b.out_bytes), b.out_bytes, map_out2 ? 0 : -1)
                        : gather_out ? format_batch_gather(b, a.rename, b.seq_no)
                                     : format_batch(b, a.rename, split_mates, b.seq_no /* rename base, set by the GPU stage */);
        if (!map_out && !gather_out && !out1_holder->plain()) {  // this batch's member(s), compressed on this worker
            Output::compress_member(out1_holder->codec(), out1_holder->level(), b.out1.data(), b.out1.size(), b.comp1);
            if (b.out1.empty()) b.comp1.clear();  // (an empty member per empty batch would only grow the file)
            b.compressed = true;
        }
        if (!map_out && out2 && !out2->plain()) {
            Output::compress_member(out2->codec(), out2->level(), b.out2.data(), b.out2.size(), b.comp2);
            if (b.out2.empty()) b.comp2.clear();
        }
        std::lock_guard<std::mutex> l(stats_m);
        tot.total_seqs += st.total_seqs;
        tot.filtered_seqs += st.filtered_seqs;
        tot.total_bp += st.total_bp;
        tot.output_bp += st.output_bp;
        tot.filtered_bp += st.filtered_bp;
    });
    // one writer: several threads pwrite()-ing one growing file serialise on its inode lock (measured slower on tmpfs)
    std::thread writer([&] {
        std::unique_ptr<Batch> b;
        while (format_stage.pop(b)) {
            if (!map_out) {  // (mapped output: already in place)
                StageClock::Scope sc(t_write);
                if (gather_out) {
                    out1_holder->write_gather(b->iov1);
                } else {
                    if (b->compressed) out1_holder->write_raw(b->comp1);
                    else out1_holder->write(b->out1);
                    if (out2) {
                        if (!out2->plain()) out2->write_raw(b->comp2);
                        else out2->write(b->out2);
                    }
                }
            }
            pool.put(std::move(b));
        }
    });

    // ---- stage 2: the GPUs.  deacon::MultiGpuFilter runs one host thread + pipeline context per entry of
    // a.devices (index replicated device to device), deals the calls round-robin and keeps two of them in flight
    // per context, so a call's PCIe copy overlaps its predecessor's kernels.  The feeder thread submits, this
    // thread waits for the calls in submission order: batches leave the stage in input order (the --rename
    // numbering is done here).  Counterpart of the worker pool of run(), src/local_filter.rs:696-709.
    std::unique_ptr<deacon::MultiGpuFilter> multi;
    try {
        multi.reset(new deacon::MultiGpuFilter(index, a.devices, cfg));
    } catch (const std::exception &e) {
        die(e.what());
    }
    m_ctx = std::chrono::duration<double>(clock::now() - start).count();
    Queue<std::unique_ptr<Batch>> submitted(2 * a.devices.size() + 2);
    // A unit (record, or pair of records) longer than the largest call.  The reference takes records of any length
    // (src/local_filter.rs:346-374); here the unit's minimizer hashes are collected piece by piece by the same kernels
    // (deacon::append_minimizer_hashes_any_length: pieces overlapping by one window, the seam's duplicate removed on the GPU's
    // own say-so), mate 1 then mate 2 (src/filter_common.rs:312-348), and the unit is decided ONCE over all of them by
    // dcn_should_keep_hashes -- the shape of the server seam (src/remote_filter.rs:230-301).  Runs on the feeder thread with
    // a context of its own; the ordinary calls of the batch go on beside it.
    std::unique_ptr<deacon::FilterProcessor> giant_proc;
    uint64_t giant_piece = 1ull << 25;  // bases per piece
    if (const char *e = std::getenv("DCN_CLI_GIANT_PIECE")) giant_piece = (uint64_t)std::max(64, std::atoi(e));  // test hook
    auto giant_unit = [&](Batch &b, size_t r0, size_t per) {
        if (!giant_proc) {
            deacon::FilterConfig gc = cfg;
            gc.max_batch_bases = giant_piece + 2 * (hd.kmer_length + hd.window_size) + 64;
            gc.max_batch_reads = 16;
            giant_proc.reset(new deacon::FilterProcessor(index, gc));
        }
        std::vector<uint64_t> hashes;
        for (size_t m = 0; m < per; ++m) {
            const Rec &r = b.recs[r0 + m];
            deacon::append_minimizer_hashes_any_length(*giant_proc, reinterpret_cast<const uint8_t *>(b.seq_ptr(r)), r.seq_len, hd.kmer_length,
                                                       hd.window_size, a.prefix_length, giant_piece - (hd.kmer_length + hd.window_size), hashes);
        }
        const uint64_t hoff[2] = {0, hashes.size()};
        dcn_params p = giant_proc->params();
        uint8_t keep = 0;
        uint32_t hits = 0, total = 0;
        deacon::check(dcn_should_keep_hashes(giant_proc->raw(), hashes.data(), hoff, 1, &p, &keep, &hits, &total));
        const size_t u = r0 / per;
        b.keep[u] = keep;
        if (a.debug) {
            b.hits[u] = hits;
            b.total[u] = total;
        }
    };
    auto submit_batch = [&](Batch &b) {
        b.gpu_seqs.clear();
        b.sub_off.clear();
        b.sub_uid.clear();
        if (b.recs.empty()) return;
        StageClock::Scope sc(t_gpu);
        size_t n_units = b.paired ? b.recs.size() / 2 : b.recs.size();
        b.keep.assign(n_units, 0);
        if (a.debug) {  // hit counts are only printed by --debug
            b.hits.assign(n_units, 0);
            b.total.assign(n_units, 0);
        }
        // a parsed chunk normally fits one call; chunks of very short or very long records are cut at unit
        // boundaries into calls that fit the context
        const size_t n = b.recs.size(), per = b.paired ? 2 : 1;
        for (size_t r0 = 0; r0 < n;) {
            size_t r1 = r0;
            while (r1 < n && (r1 - r0) + per <= cfg.max_batch_reads && b.offsets[r1 + per] - b.offsets[r0] <= cfg.max_batch_bases)
                r1 += per;
            if (r1 == r0) {  // a unit no call can hold (a chromosome among the reads): see giant_unit
                giant_unit(b, r0, per);
                r0 += per;
                continue;
            }
            const size_t u0 = r0 / per;
            deacon::MultiGpuFilter::Job job;
            if (b.seq_in_chars) {
                job.packed = b.packed.data();
                job.invmask = b.invmask.data();
                if (r0 != 0) {  // a later piece: its bits moved so that its first base is base 0 of a stream of its own
                    const uint64_t base0 = b.offsets[r0], nbp = b.offsets[r1] - base0, groups = (nbp + 31) / 32;
                    const uint64_t alloc = std::max<uint64_t>(groups, 1);
                    const uint64_t g0 = base0 / 32, have = b.invmask.size();
                    const unsigned sh = (unsigned)(base0 % 32);
                    b.sub_packed.emplace_back(2 * alloc, 0u);
                    b.sub_mask.emplace_back(alloc, 0u);
                    uint32_t *sp = b.sub_packed.back().data(), *sm = b.sub_mask.back().data();
                    auto P = [&](uint64_t g) {
                        uint64_t v = 0;
                        if (g < have) std::memcpy(&v, b.packed.data() + 2 * g, 8);
                        return v;
                    };
                    auto M = [&](uint64_t g) { return g < have ? b.invmask[g] : 0u; };
                    for (uint64_t j = 0; j < groups; ++j) {
                        uint64_t v = sh ? (P(g0 + j) >> (2 * sh)) | (P(g0 + j + 1) << (64 - 2 * sh)) : P(g0 + j);
                        uint32_t m = sh ? (M(g0 + j) >> sh) | (M(g0 + j + 1) << (32 - sh)) : M(g0 + j);
                        const uint64_t left = nbp - 32 * j;  // bases of this group that belong to the piece
                        if (left < 32) v &= (1ull << (2 * left)) - 1, m &= (1u << left) - 1;
                        std::memcpy(sp + 2 * j, &v, 8);
                        sm[j] = m;
                    }
                    job.packed = sp;
                    job.invmask = sm;
                }
            } else {
                job.bases = b.bases.data() + b.offsets[r0];
            }
            job.offsets = b.offsets.data();
            job.unit_id = b.paired ? b.unit_id.data() : nullptr;
            if (r0 != 0) {  // offsets and unit ids of a later piece start from zero again
                b.sub_off.emplace_back(r1 - r0 + 1);
                for (size_t r = r0; r <= r1; ++r) b.sub_off.back()[r - r0] = b.offsets[r] - b.offsets[r0];
                job.offsets = b.sub_off.back().data();
                if (b.paired) {
                    b.sub_uid.emplace_back(r1 - r0);
                    for (size_t r = r0; r < r1; ++r) b.sub_uid.back()[r - r0] = b.unit_id[r] - b.unit_id[r0];
                    job.unit_id = b.sub_uid.back().data();
                }
            }
            job.n_reads = (uint32_t)(r1 - r0);
            job.keep = b.keep.data() + u0;
            job.hits = a.debug ? b.hits.data() + u0 : nullptr;
            job.total = a.debug ? b.total.data() + u0 : nullptr;
            b.gpu_seqs.push_back(multi->submit(job));
            r0 = r1;
        }
    };
    uint64_t out_bytes_total = 0, out_bytes_total2 = 0;  // mapped outputs: bytes placed so far
    std::unique_ptr<deacon::FilterProcessor> debug_proc;  // only --debug re-scans batches (for the k-mer strings)
    std::thread feeder([&] {
        std::unique_ptr<Batch> b;
        try {
            while (next_parsed(b)) {
                submit_batch(*b);
                submitted.push(std::move(b));
            }
        } catch (const std::exception &e) {
            die(e.what());
        }
        submitted.finish();
        m_feeder_done = std::chrono::duration<double>(clock::now() - start).count();
    });
    {
        std::unique_ptr<Batch> b;
        uint64_t written_before = 0;
        for (;;) {
            {
                StageClock::Scope sc(t_gpu_wait);
                if (!submitted.pop(b)) break;
                try {
                    for (uint64_t seq : b->gpu_seqs) multi->wait(seq);
                } catch (const std::exception &e) {
                    die(e.what());
                }
            }
            if (b->recs.empty()) continue;
            size_t n_units = b->paired ? b->recs.size() / 2 : b->recs.size();
            if (a.debug) {
                if (!debug_proc) debug_proc.reset(new deacon::FilterProcessor(index, cfg));
                debug_lines(*debug_proc, index, *b, n_units, a.prefix_length);
            }
            b->seq_no = written_before;  // records written before this batch: base of --rename numbering
            size_t kept_units = 0;
            for (size_t u = 0; u < n_units; ++u) kept_units += b->keep[u] != 0;
            if (map_out) {  // this batch's place in the output file
                b->out_off = out_bytes_total;
                b->out_bytes = formatted_size(*b, a.rename, written_before, map_out2 ? 0 : -1);
                out_bytes_total += b->out_bytes;
                if (map_out2) {
                    b->out_off2 = out_bytes_total2;
                    b->out_bytes2 = formatted_size(*b, a.rename, written_before, 1);
                    out_bytes_total2 += b->out_bytes2;
                }
            }
            written_before += kept_units * (b->paired ? 2 : 1);
            StageClock::Scope sc(t_push_wait);
            format_stage.push(std::move(b));
        }
        format_stage.finish();
        m_gpu_done = std::chrono::duration<double>(clock::now() - start).count();
    }
    feeder.join();
    reader.join();
    if (reader2.joinable()) reader2.join();
    writer.join();
    m_written = std::chrono::duration<double>(clock::now() - start).count();
    if (map_out) mapped_out.finish(out_bytes_total);
    else out1_holder->close();
    if (map_out2) mapped_out2.finish(out_bytes_total2);
    if (out2) out2->close();
    uint64_t total_seqs = tot.total_seqs, filtered_seqs = tot.filtered_seqs, total_bp = tot.total_bp,
             output_bp = tot.output_bp, filtered_bp = tot.filtered_bp;

    double secs = std::chrono::duration<double>(clock::now() - start).count();
    if (cli_timing) {
        struct rusage ru;
        if (getrusage(RUSAGE_SELF, &ru) == 0)  // every thread of the process, the library's host threads and the runtime's included
            std::fprintf(stderr, "timing: process CPU %.3f s user + %.3f s system\n", ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6,
                         ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6);

    }
    if (cli_timing)
        std::fprintf(stderr, "timing: wall %.3f s; busy seconds: parse %.3f (all workers), GPU stage %.3f (main thread waited %.3f for it), "
                             "format %.3f (all workers), write %.3f; main thread blocked pushing to format %.3f\n"
                             "timing: milestones (s): index loaded %.3f, contexts ready %.3f, all input parsed+queued %.3f, "
                             "GPU stage drained %.3f, all written %.3f\n", secs, t_parse.seconds(), t_gpu.seconds(),
                     t_gpu_wait.seconds(), t_format.seconds(), t_write.seconds(), t_push_wait.seconds(), m_index, m_ctx,
                     m_feeder_done, m_gpu_done, m_written);
    uint64_t seqs_out = total_seqs - filtered_seqs;
    auto prop = [](uint64_t x, uint64_t y) { return y ? (double)x / (double)y : 0.0; };
    if (!quiet)
        std::fprintf(stderr,
                     "Retained %llu/%llu sequences (%.3f%%), %llu/%llu bp (%.3f%%) in %s. Speed: %.0f seqs/s (%.1f Mbp/s)\n",
                     (unsigned long long)seqs_out, (unsigned long long)total_seqs, prop(seqs_out, total_seqs) * 100.0,
                     (unsigned long long)output_bp, (unsigned long long)total_bp, prop(output_bp, total_bp) * 100.0,
                     fmt_duration(secs).c_str(), total_seqs / secs, total_bp / secs / 1e6);
    if (a.has_summary) {  // FilterSummary, src/filter_common.rs:11-38
        FILE *f = std::fopen(a.summary.c_str(), "w");
        if (!f) die("Failed to create summary: " + a.summary);
        auto opt = [&](bool has, const std::string &v) { return has ? json_str(v) : std::string("null"); };
        std::fprintf(f,
                     "{\n  \"version\": %s,\n  \"index\": %s,\n  \"input\": %s,\n  \"input2\": %s,\n  \"output\": %s,\n"
                     "  \"output2\": %s,\n  \"k\": %u,\n  \"w\": %u,\n  \"abs_threshold\": %u,\n  \"rel_threshold\": %.17g,\n"
                     "  \"prefix_length\": %zu,\n  \"deplete\": %s,\n  \"rename\": %s,\n  \"seqs_in\": %llu,\n  \"seqs_out\": %llu,\n"
                     "  \"seqs_out_proportion\": %.17g,\n  \"seqs_removed\": %llu,\n  \"seqs_removed_proportion\": %.17g,\n"
                     "  \"bp_in\": %llu,\n  \"bp_out\": %llu,\n  \"bp_out_proportion\": %.17g,\n  \"bp_removed\": %llu,\n"
                     "  \"bp_removed_proportion\": %.17g,\n  \"time\": %.17g,\n  \"seqs_per_second\": %llu,\n  \"bp_per_second\": %llu\n}",
                     json_str(std::string("deacon-hip ") + VERSION).c_str(), json_str(a.index).c_str(), json_str(a.input).c_str(),
                     opt(a.has_input2, a.input2).c_str(), json_str(a.output).c_str(), opt(a.has_output2, a.output2).c_str(),
                     hd.kmer_length, hd.window_size, a.abs_threshold, a.rel_threshold, a.prefix_length,
                     a.deplete ? "true" : "false", a.rename ? "true" : "false", (unsigned long long)total_seqs,
                     (unsigned long long)seqs_out, prop(seqs_out, total_seqs), (unsigned long long)filtered_seqs,
                     prop(filtered_seqs, total_seqs), (unsigned long long)total_bp, (unsigned long long)output_bp,
                     prop(output_bp, total_bp), (unsigned long long)filtered_bp, prop(filtered_bp, total_bp), secs,
                     (unsigned long long)(total_seqs / secs), (unsigned long long)(total_bp / secs));
        if (std::ferror(f) || std::fclose(f) != 0) die("Failed to write summary: " + a.summary);
        if (!quiet) std::fprintf(stderr, "Summary saved to \"%s\"\n", a.summary.c_str());
    }
    // Everything is written and closed.  Tearing the pipeline down in order (unmapping gigabytes of input, freeing
    // every batch, destroying the device table and the HIP runtime) costs ~0.25 s and changes nothing observable:
    // leave it to the OS unless asked (DCN_CLI_FULL_TEARDOWN, e.g. under a leak checker).
    if (!std::getenv("DCN_CLI_FULL_TEARDOWN")) {
        std::fflush(nullptr);
        _exit(0);
    }
    return 0;
}

// ---- deacon index build (src/index.rs:167-308) / info (:539-560) ----------------------------------------------
// A whole FASTA input as one batch, for `index build` (src/index.rs:167-308 reads the reference genome record by record the same
// way its filter does): the record reader joins an 80-column genome's lines at ~1 GB/s on one thread, which was 40 % of an index
// build.  Here the input is taken whole (mapped, or decompressed by the readers above), cut into slices at line starts, and every
// slice's lines are classified (header / sequence) and copied on a thread of their own; a record's bases may come from several
// slices.  false: not FASTA (the first record starts with '@') -- the caller falls back to the record reader.
bool read_whole_fasta(const std::string &path, Batch &all, std::vector<char, DefaultInitAllocator<char>> &raw_store, MappedFile &mapped) {
    const char *d = nullptr;
    size_t n = 0;
    if (mapped.open(path)) {
        d = mapped.data;
        n = mapped.size;
    } else {
        Input in(path);
        // (a compressed genome is 3.5-4.5 x its file: room for 5 x at once, so that the text is not moved when the array grows;
        // its pages are touched ahead of the reader by a thread of their own where the kernel offers that)
        struct stat st;
        size_t guess = 64u << 20;
        if (::stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode)) guess = std::max<size_t>(guess, (size_t)st.st_size * 5);
        raw_store.resize(guess);
        std::thread toucher;
        std::atomic<bool> read_done{false};
#ifdef MADV_POPULATE_WRITE
        {
            char *base = raw_store.data();
            const size_t len = raw_store.size();
            toucher = std::thread([base, len, &read_done] {
                const uintptr_t lo = ((uintptr_t)base + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)base + len) & ~(uintptr_t)4095;
                for (uintptr_t p = lo; p < hi && !read_done.load(); p += 64u << 20)
                    (void)madvise((void *)p, std::min<size_t>(64u << 20, hi - p), MADV_POPULATE_WRITE);
            });
        }
#endif
        for (;;) {
            if (n == raw_store.size()) {
                read_done = true;
                if (toucher.joinable()) toucher.join();
                raw_store.resize(raw_store.size() + raw_store.size() / 2);
            }
            const size_t got = in.read(raw_store.data() + n, raw_store.size() - n);
            if (got == 0) break;
            n += got;
        }
        read_done = true;
        if (toucher.joinable()) toucher.join();
        d = raw_store.data();
    }
    size_t first = 0;
    while (first < n && (d[first] == '\n' || d[first] == '\r')) ++first;
    if (first == n) return true;  // empty input: no records
    if (d[first] == '@') return false;
    if (d[first] != '>') die("Invalid FASTX record start: expected '>' or '@'");
    struct Slice {
        size_t a = 0, b = 0, n_bases = 0, base0 = 0;
        std::vector<std::pair<size_t, size_t>> headers;  // (start of the header line, bases of this slice in front of it)
    };
    const size_t n_slices = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(usable_cpus(), 32), n / (4u << 20)));
    std::vector<Slice> sl(n_slices);
    for (size_t t = 0; t < n_slices; ++t) {
        size_t a = t == 0 ? first : n * t / n_slices;
        if (t > 0) {  // forward to the next line start
            const char *nl = (const char *)std::memchr(d + a, '\n', n - a);
            a = nl ? (size_t)(nl - d) + 1 : n;
        }
        sl[t].a = a;
        if (t > 0) sl[t - 1].b = a;
    }
    sl.back().b = n;
    auto walk = [&](Slice &s, uint8_t *out) {  // out == nullptr: count only
        size_t nb = 0;
        for (size_t p = s.a; p < s.b;) {
            const char *nl = (const char *)std::memchr(d + p, '\n', s.b - p);
            size_t e = nl ? (size_t)(nl - d) : s.b, next = nl ? e + 1 : s.b;
            if (e > p && d[e - 1] == '\r') --e;
            if (e > p && d[p] == '>') {
                if (!out) s.headers.emplace_back(p, nb);
            } else {
                if (out) std::memcpy(out + nb, d + p, e - p);
                nb += e - p;
            }
            p = next;
        }
        s.n_bases = nb;
    };
    parallel_for(n_slices, n_slices, [&](size_t t) { walk(sl[t], nullptr); });
    size_t total = 0;
    for (auto &s : sl) s.base0 = total, total += s.n_bases;
    all.bases.resize(total);
    parallel_for(n_slices, n_slices, [&](size_t t) { walk(sl[t], all.bases.data() + sl[t].base0); });
    all.ext = d;
    all.offsets.clear();
    all.recs.clear();
    for (auto &s : sl)
        for (auto &h : s.headers) {
            Rec r;
            size_t e = line_end(d, n, h.first);
            if (e > h.first && d[e - 1] == '\r') --e;
            r.id_off = h.first + 1;
            r.id_len = (uint32_t)(e - h.first - 1);
            r.seq_off = s.base0 + h.second;
            r.qual_off = NO_QUAL;
            r.seq_len = 0;
            all.recs.push_back(r);
            all.offsets.push_back(r.seq_off);
        }
    all.offsets.push_back(total);
    for (size_t i = 0; i < all.recs.size(); ++i) all.recs[i].seq_len = (uint32_t)(all.offsets[i + 1] - all.offsets[i]);
    return true;
}

int run_index_build(const std::string &input, unsigned k, unsigned w, const std::string &output, size_t capacity_millions,
                    float entropy, bool quiet) {
    auto start = std::chrono::steady_clock::now();
    std::fprintf(stderr, "Deacon-hip v%s; mode: build; input: single; options: capacity=%zuM\n", VERSION, capacity_millions);
    if ((k + w - 1) % 2 == 0)
        die("Constraint violated: k + w - 1 must be odd (k=" + std::to_string(k) + ", w=" + std::to_string(w) + ")");
    std::fprintf(stderr, "Building index (k=%u, w=%u)\n", k, w);
    Batch all;
    std::vector<char, DefaultInitAllocator<char>> raw_store;
    MappedFile raw_map;
    // a FASTA file (plain or compressed) is taken whole and its lines joined on all threads; stdin, FASTQ and
    // DCN_CLI_NO_CHUNK_READER=1 go record by record
    if (input != "-" && !std::getenv("DCN_CLI_NO_CHUNK_READER") && read_whole_fasta(input, all, raw_store, raw_map)) {
        if (!quiet)
            for (const Rec &r : all.recs) std::fprintf(stderr, "  %.*s (%ubp)\n", (int)r.id_len, all.chars() + r.id_off, r.seq_len);
    } else {
        all.reset();
        FastxReader rd(input);
        while (rd.next(all)) {
            if (!quiet) {
                const Rec &r = all.recs.back();
                std::fprintf(stderr, "  %.*s (%ubp)\n", (int)r.id_len, all.chars() + r.id_off, r.seq_len);
            }
        }
    }
    const bool timing = std::getenv("DCN_CLI_TIMING") != nullptr;
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count(); };
    const double t_read = since();
    dcn_index *raw = nullptr;
    // the capacity flag is only a pre-allocation hint (the table grows as needed); cap it by what the input can hold
    uint64_t hint = std::min<uint64_t>((uint64_t)capacity_millions * 1000000ull, all.bases.size() / 4 + 1024);
    deacon::check(dcn_index_build(all.bases.data(), all.offsets.data(), (uint32_t)all.recs.size(), (uint8_t)k, (uint8_t)w,
                                  entropy, hint, 0, &raw));
    const double t_built = since();
    uint64_t n = 0;
    deacon::check(dcn_index_header(raw, nullptr, nullptr, &n));
    std::fprintf(stderr, "Indexed %llu minimizers from %zu sequence(s) (%zubp)\n", (unsigned long long)n, all.recs.size(),
                 all.bases.size());
    std::string path = output;
    if (output == "-") path = "/dev/stdout";
    int rc = dcn_index_write_file(raw, path.c_str());
    dcn_index_destroy(raw);
    deacon::check(rc);
    if (timing)
        std::fprintf(stderr, "timing: input read %.3f s, index built on the GPU %.3f s, index file written %.3f s\n", t_read, t_built - t_read,
                     since() - t_built);
    std::fprintf(stderr, "Completed in %s\n",
                 fmt_duration(std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count()).c_str());
    return 0;
}

int run_index_info(const std::string &path) {
    auto start = std::chrono::steady_clock::now();
    deacon::Index idx = deacon::Index::load(path);
    auto hd = idx.header();
    std::fprintf(stderr, "Index information:\n  Format version: %u\n  K-mer length (k): %u\n  Window size (w): %u\n"
                         "  Distinct minimizer count: %llu\n",
                 hd.format_version, hd.kmer_length, hd.window_size, (unsigned long long)idx.len());
    std::fprintf(stderr, "Retrieved index info in %s\n",
                 fmt_duration(std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count()).c_str());
    return 0;
}

void write_index(dcn_index *raw, const std::string &output) {
    std::string path = output == "-" ? "/dev/stdout" : output;
    deacon::check(dcn_index_write_file(raw, path.c_str()));
}

struct RawIndex {  // owns a dcn_index*
    dcn_index *p = nullptr;
    ~RawIndex() {
        if (p) dcn_index_destroy(p);
    }
};

// index::union (src/index.rs:563-664)
int run_index_union(const std::vector<std::string> &inputs, const std::string &output) {
    auto start = std::chrono::steady_clock::now();
    std::vector<RawIndex> idx(inputs.size());
    std::vector<const dcn_index *> ptrs;
    for (size_t i = 0; i < inputs.size(); ++i) {
        deacon::check(dcn_index_from_file(inputs[i].c_str(), 0, &idx[i].p));
        uint64_t n = 0;
        dcn_index_header(idx[i].p, nullptr, nullptr, &n);
        std::fprintf(stderr, "Index %zu: %llu minimizers\n", i + 1, (unsigned long long)n);
        ptrs.push_back(idx[i].p);
    }
    RawIndex out;
    deacon::check(dcn_index_union(ptrs.data(), (uint32_t)ptrs.size(), &out.p));
    uint64_t n = 0;
    dcn_index_header(out.p, nullptr, nullptr, &n);
    std::fprintf(stderr, "Union: %llu minimizers from %zu indexes\n", (unsigned long long)n, inputs.size());
    write_index(out.p, output);
    std::fprintf(stderr, "Completed union operation in %s\n",
                 fmt_duration(std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count()).c_str());
    return 0;
}

// index::diff (src/index.rs:421-536): second is an index file, or a FASTX file when -k/-w are given or when it does
// not parse as an index (then the first index's k, w are used)
int run_index_diff(const std::string &first, const std::string &second, int k_opt, int w_opt, const std::string &output) {
    auto start = std::chrono::steady_clock::now();
    RawIndex a, b, out;
    deacon::check(dcn_index_from_file(first.c_str(), 0, &a.p));
    uint8_t k = 0, w = 0;
    uint64_t na = 0;
    dcn_index_header(a.p, &k, &w, &na);
    std::fprintf(stderr, "First index: loaded %llu minimizers\n", (unsigned long long)na);
    bool fastx = k_opt > 0 && w_opt > 0;
    if (!fastx && dcn_index_from_file(second.c_str(), 0, &b.p) != DCN_OK) {
        fastx = true;  // not an index file: treat it as FASTX with the first index's parameters
        k_opt = k;
        w_opt = w;
    }
    if (fastx) {
        if (k_opt != k || w_opt != w)
            die("FASTX parameters (k=" + std::to_string(k_opt) + ", w=" + std::to_string(w_opt) + ") must match first index (k=" +
                std::to_string((int)k) + ", w=" + std::to_string((int)w) + ")");
        std::fprintf(stderr, "Second index: processing FASTX from %s (k=%d, w=%d)…\n", second == "-" ? "stdin" : "file", k_opt, w_opt);
        FastxReader rd(second);
        Batch all;
        while (rd.next(all)) {
        }
        deacon::check(dcn_index_build(all.bases.data(), all.offsets.data(), (uint32_t)all.recs.size(), k, w, 0.0f,
                                      all.bases.size() / 4 + 1024, 0, &b.p));
    } else {
        uint64_t nb = 0;
        dcn_index_header(b.p, nullptr, nullptr, &nb);
        std::fprintf(stderr, "Second index: loaded %llu minimizers\n", (unsigned long long)nb);
    }
    deacon::check(dcn_index_diff(a.p, b.p, &out.p));
    uint64_t n = 0;
    dcn_index_header(out.p, nullptr, nullptr, &n);
    std::fprintf(stderr, "Removed %llu minimizers, %llu remaining\n", (unsigned long long)(na - n), (unsigned long long)n);
    write_index(out.p, output);
    std::fprintf(stderr, "Completed difference operation in %s\n",
                 fmt_duration(std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count()).c_str());
    return 0;
}

void usage() {
    std::fprintf(stderr,
                 "Usage: deacon-hip <COMMAND>\n\nCommands:\n  index   Build and compose minimizer indexes (build, info, union, diff)\n"
                 "  filter  Keep or discard DNA fastx records with sufficient minimizer hits to an index\n"
                 "  server  Hold a pre-loaded minimizer index on the GPU for filtering with the client command\n"
                 "  client  Alternate version of filter: minimizers computed here, the index held by a server\n\n"
                 "Options:\n  -h, --help     Print help\n  -V, --version  Print version\n");
}

// `deacon-hip server IDX -p PORT` and `deacon-hip client ADDR [INPUT] [INPUT2] ...` (src/main.rs:86-157): the server surface is
// the package's Python (deacon_server_amd/server.py, client.py over the same library); the tool starts it as a CHILD process
// with the repository root on PYTHONPATH and returns its exit code (never an exec of this process: it is linked against the
// HIP runtime).
int run_python_module(const std::vector<std::string> &args) {
    char self[4096];
    const ssize_t n = ::readlink("/proc/self/exe", self, sizeof self - 1);
    if (n <= 0) die("cannot find the tool's own path");
    self[n] = 0;
    std::string root(self);  // <root>/deacon-server_amd/bin/deacon-hip
    for (int up = 0; up < 3; ++up) root.erase(root.find_last_of('/'));
    std::string pp = root;
    if (const char *e = std::getenv("PYTHONPATH")) pp += std::string(":") + e;
    ::setenv("PYTHONPATH", pp.c_str(), 1);
    const std::string module = "deacon_server_amd." + args[0];
    std::vector<std::string> owned = {"python3", "-m", module};
    owned.insert(owned.end(), args.begin() + 1, args.end());
    std::vector<char *> argv;
    for (auto &a : owned) argv.push_back(a.data());
    argv.push_back(nullptr);
    pid_t pid = 0;
    if (::posix_spawnp(&pid, "python3", nullptr, nullptr, argv.data(), environ) != 0) die("cannot start python3");
    int status = 0;
    while (::waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
    return WIFEXITED(status) ? WEXITSTATUS(status) : 128 + WTERMSIG(status);
}

// `deacon-hip <subcommand> --help`: the options of src/main.rs:17-235 with the reference's short / long names and defaults,
// plus the two this tool adds (--gpus, --devices).  Printed to stdout with exit code 0, as clap does.
bool subcommand_help(const std::vector<std::string> &args) {
    bool asked = false;
    for (size_t i = 1; i < args.size(); ++i) asked |= args[i] == "--help" || args[i] == "-h";
    if (!asked) return false;
    const std::string sub = args[0] == "index" && args.size() >= 2 && args[1][0] != '-' ? "index " + args[1] : args[0];
    const char *text = nullptr;
    if (sub == "filter")
        text = "Keep or discard DNA fastx records with sufficient minimizer hits to an index\n\n"
               "Usage: deacon-hip filter [OPTIONS] <INDEX> [INPUT] [INPUT2]\n\n"
               "Arguments:\n"
               "  <INDEX>   Path to minimizer index file\n"
               "  [INPUT]   Optional path to fastx file (or - for stdin; gz, bgzf, zst, xz and bz2 found by content) [default: -]\n"
               "  [INPUT2]  Optional path to second paired fastx file (or - for interleaved stdin)\n\n"
               "Options:\n"
               "  -o, --output <OUTPUT>          Path to output fastx file (or - for stdout; .gz, .zst and .xz by extension) [default: -]\n"
               "  -O, --output2 <OUTPUT2>        Optional path to second paired output fastx file\n"
               "  -a, --abs-threshold <N>        Minimum absolute number of minimizer hits for a match [default: 2]\n"
               "  -r, --rel-threshold <F>        Minimum relative proportion (0.0-1.0) of minimizer hits for a match [default: 0.01]\n"
               "  -p, --prefix-length <N>        Search only the first N nucleotides per sequence (0 = entire sequence) [default: 0]\n"
               "  -d, --deplete                  Discard matching sequences (invert filtering behaviour)\n"
               "  -R, --rename                   Replace sequence headers with incrementing numbers\n"
               "  -s, --summary <SUMMARY>        Path to JSON summary output file\n"
               "  -t, --threads <THREADS>        Number of host threads (0 = auto) [default: 3/4 of the usable CPUs, at least 8]\n"
               "      --compression-level <N>    Output compression level (1-9 for gz & xz; 1-22 for zstd) [default: 2]\n"
               "      --debug                    Output sequences with minimizer hits to stderr\n"
               "  -q, --quiet                    Suppress progress reporting\n"
               "      --gpus <N>                 Use GPUs 0..N-1, one pipeline context and one index replica each [default: 1]\n"
               "      --devices <LIST>           Explicit device list, e.g. 0,2,3 (a repeated id = another context on that GPU)\n"
               "  -h, --help                     Print help\n";
    else if (sub == "index build")
        text = "Index minimizers contained within a fastx file\n\n"
               "Usage: deacon-hip index build [OPTIONS] <INPUT>\n\n"
               "Arguments:\n  <INPUT>  Path to input fastx file (gz, bgzf, zst, xz and bz2 found by content)\n\n"
               "Options:\n"
               "  -k <K>                         K-mer length used for indexing (1-57) [default: 31]\n"
               "  -w <W>                         Minimizer window size used for indexing [default: 15]\n"
               "  -o, --output <OUTPUT>          Path to output file (- for stdout) [default: -]\n"
               "  -c, --capacity <CAPACITY>      Preallocated index capacity in millions of minimizers (accepted; the device table sizes itself) [default: 400]\n"
               "  -t, --threads <THREADS>        Accepted for compatibility (the scan runs on the GPU)\n"
               "  -q, --quiet                    Suppress sequence header output\n"
               "  -e, --entropy-threshold <F>    Minimum scaled entropy threshold for k-mer filtering (0.0-1.0) [default: 0.0]\n"
               "  -h, --help                     Print help\n";
    else if (sub == "index info")
        text = "Show index information\n\nUsage: deacon-hip index info <INDEX>\n\nArguments:\n  <INDEX>  Path to index file\n";
    else if (sub == "index union")
        text = "Combine multiple minimizer indexes (A u B...)\n\n"
               "Usage: deacon-hip index union [OPTIONS] <INPUTS>...\n\n"
               "Arguments:\n  <INPUTS>...  Path(s) to one or more index file(s)\n\n"
               "Options:\n"
               "  -o, --output <OUTPUT>      Path to output file (- for stdout) [default: -]\n"
               "  -c, --capacity <CAPACITY>  Accepted for compatibility (the device table sizes itself)\n"
               "  -h, --help                 Print help\n";
    else if (sub == "index diff")
        text = "Subtract minimizers in one index from another (A - B)\n\n"
               "Usage: deacon-hip index diff [OPTIONS] <FIRST> <SECOND>\n\n"
               "Arguments:\n"
               "  <FIRST>   Path to first index file\n"
               "  <SECOND>  Path to second index file or FASTX file (or - for stdin when using FASTX)\n\n"
               "Options:\n"
               "  -k, --kmer-length <K>    K-mer length (required if second argument is FASTX file, 1-32)\n"
               "  -w, --window-size <W>    Window size (required if second argument is FASTX file)\n"
               "  -o, --output <OUTPUT>    Path to output file (- for stdout) [default: -]\n"
               "  -h, --help               Print help\n";
    else if (sub == "index")
        text = "Build and compose minimizer indexes\n\n"
               "Usage: deacon-hip index <COMMAND>\n\n"
               "Commands:\n"
               "  build  Index minimizers contained within a fastx file\n"
               "  info   Show index information\n"
               "  union  Combine multiple minimizer indexes (A u B...)\n"
               "  diff   Subtract minimizers in one index from another (A - B)\n";
    if (!text) return false;
    std::fputs(text, stdout);
    return true;
}

}  // namespace

int main(int argc, char **argv) {
    std::vector<std::string> args(argv + 1, argv + argc);
    if (args.empty()) {
        usage();
        return 2;  // clap exits with 2 when a required subcommand is missing (tests/cli_tests.rs)
    }
    try {
        if (args[0] == "--version" || args[0] == "-V") {
            std::printf("deacon-hip %s\n", VERSION);
            return 0;
        }
        if (args[0] == "--help" || args[0] == "-h") {
            usage();
            return 0;
        }
        if (args[0] == "server" || args[0] == "client") return run_python_module(args);
        if (subcommand_help(args)) return 0;
        auto need = [&](size_t i) -> const std::string & {
            if (i >= args.size()) die("missing value for " + args[i - 1]);
            return args[i];
        };
        if (args[0] == "filter") {
            FilterArgs a;
            std::vector<std::string> pos;
            for (size_t i = 1; i < args.size(); ++i) {
                const std::string &s = args[i];
                if (s == "-o" || s == "--output") a.output = need(++i);
                else if (s == "-O" || s == "--output2") a.output2 = need(++i), a.has_output2 = true;
                else if (s == "-a" || s == "--abs-threshold") {
                    long v = std::atol(need(++i).c_str());
                    if (v < 1 || v > 65535) die("invalid value for --abs-threshold: must be 1..65535");
                    a.abs_threshold = (unsigned)v;
                } else if (s == "-r" || s == "--rel-threshold") a.rel_threshold = std::atof(need(++i).c_str());
                else if (s == "-p" || s == "--prefix-length") a.prefix_length = (size_t)std::atoll(need(++i).c_str());
                else if (s == "-d" || s == "--deplete") a.deplete = true;
                else if (s == "-R" || s == "--rename") a.rename = true;
                else if (s == "-s" || s == "--summary") a.summary = need(++i), a.has_summary = true;
                else if (s == "-t" || s == "--threads") a.threads = (size_t)std::atoll(need(++i).c_str()), a.threads_given = true;
                else if (s == "--compression-level") a.compression_level = std::atoi(need(++i).c_str());
                else if (s == "--debug") a.debug = true;
                else if (s == "-q" || s == "--quiet") a.quiet = true;
                else if (s == "--gpus") {  // GPUs 0..N-1, one pipeline context each
                    int n = std::atoi(need(++i).c_str());
                    if (n < 1 || n > 64) die("invalid value for --gpus: must be 1..64");
                    a.devices.clear();
                    for (int d = 0; d < n; ++d) a.devices.push_back(d);
                } else if (s == "--devices") {  // explicit list, repeats allowed: 0,0 = two contexts on GPU 0
                    a.devices.clear();
                    std::stringstream ss(need(++i));
                    for (std::string tok; std::getline(ss, tok, ',');) {
                        char *end = nullptr;
                        long d = std::strtol(tok.c_str(), &end, 10);
                        if (tok.empty() || *end || d < 0 || d > 1023) die("invalid value for --devices: '" + tok + "'");
                        a.devices.push_back((int)d);
                    }
                    if (a.devices.empty() || a.devices.size() > 64) die("invalid value for --devices");
                }
                else if (s.size() > 1 && s[0] == '-' && s != "-") die("unexpected argument '" + s + "'");
                else pos.push_back(s);
            }
            if (pos.empty()) die("the following required arguments were not provided: <INDEX>");
            a.index = pos[0];
            if (pos.size() > 1) a.input = pos[1];
            if (pos.size() > 2) a.input2 = pos[2], a.has_input2 = true;
            if (pos.size() > 3) die("unexpected argument '" + pos[3] + "'");
            return run_filter(a);
        }
        if (args[0] == "cat" && args.size() >= 2) {  // hidden: the input side alone (format found by content, decoded to stdout; no GPU)
            Input in(args[1]);
            std::vector<char> buf(8u << 20);
            const auto t0 = std::chrono::steady_clock::now();
            uint64_t total = 0;
            for (size_t got; (got = in.read(buf.data(), buf.size())) > 0;) {
                total += got;
                if (args.size() < 3 || args[2] != "--count") {
                    size_t w = 0;
                    while (w < got) {
                        const ssize_t r = ::write(1, buf.data() + w, got - w);
                        if (r < 0) die("write error");
                        w += (size_t)r;
                    }
                }
            }
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::fprintf(stderr, "decoded %llu bytes in %.3f s: %.2f GB/s\n", (unsigned long long)total, sec, total / sec / 1e9);
            return 0;
        }
        if (args[0] == "compress" && args.size() >= 2) {  // hidden: a writer alone, streaming (stdin -> members / frames of gz | zst | xz on
                                                          // stdout; no GPU): compress <codec> [level].  The Python client's .zst outputs.
            const int level = args.size() >= 3 ? std::atoi(args[2].c_str()) : 2;
            const std::string fake = "x." + args[1];
            Output::Codec codec;
            if (args[1] == "gz") codec = Output::GZIP;
            else if (args[1] == "zst") codec = Output::ZSTD;
            else if (args[1] == "xz") codec = Output::XZ;
            else die("compress: codec must be gz, zst or xz");
            Output::check_level(fake, level);
            std::vector<char> buf(8u << 20), piece;
            size_t have = 0, total = 0;
            auto flush_piece = [&](size_t n) {
                Output::compress_member(codec, level, buf.data(), n, piece);
                for (size_t w = 0; w < piece.size();) {
                    const ssize_t r = ::write(1, piece.data() + w, piece.size() - w);
                    if (r < 0) die("write error");
                    w += (size_t)r;
                }
            };
            for (ssize_t r; (r = ::read(0, buf.data() + have, buf.size() - have)) > 0;) {
                have += (size_t)r;
                total += (size_t)r;
                if (have == buf.size()) flush_piece(have), have = 0;
            }
            if (have || codec != Output::GZIP || total == 0) flush_piece(have);  // (an empty input is still one valid empty frame)
            if (codec == Output::GZIP && total != 0) flush_piece(0);          // BGZF's end-of-file member
            return 0;
        }
        if (args[0] == "gz" && args.size() >= 1) {  // hidden: the .gz writer alone (stdin -> BGZF members on stdout; no GPU): gz [level]
            const int level = args.size() >= 2 ? std::atoi(args[1].c_str()) : 2;
            std::vector<char> in, piece;
            std::vector<char> buf(1u << 20);
            for (ssize_t r; (r = ::read(0, buf.data(), buf.size())) > 0;) in.insert(in.end(), buf.begin(), buf.begin() + r);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<char> all;
            for (size_t pos = 0; pos < in.size(); pos += 8u << 20) {  // (a batch's worth per call, as the formatter threads make them)
                Output::compress_member(Output::GZIP, level, in.data() + pos, std::min<size_t>(8u << 20, in.size() - pos), piece);
                all.insert(all.end(), piece.begin(), piece.end());
            }
            Output::compress_member(Output::GZIP, level, "", 0, piece);  // the end-of-file member
            all.insert(all.end(), piece.begin(), piece.end());
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            for (size_t w = 0; w < all.size();) {
                const ssize_t r = ::write(1, all.data() + w, all.size() - w);
                if (r < 0) die("write error");
                w += (size_t)r;
            }
            std::fprintf(stderr, "compressed %zu -> %zu bytes in %.3f s: %.0f MB/s on one thread\n", in.size(), all.size(), sec, in.size() / sec / 1e6);
            return 0;
        }
        if (args[0] == "bench-parse" && args.size() >= 2) {  // hidden: the parser pool alone on a plain FASTX file (no GPU)
            size_t threads = 8;
            bool packed = cli_can_pack(), verify = false, count_only = false, prefault = false;
            for (size_t i = 2; i < args.size(); ++i) {
                if (args[i] == "-t" && i + 1 < args.size()) threads = (size_t)std::atoll(args[++i].c_str());
                else if (args[i] == "--ascii") packed = false;
                else if (args[i] == "--count") count_only = true;  // the record walk alone: no sequence copy, no Rec kept
                else if (args[i] == "--prefault") prefault = true;  // experiment: MADV_POPULATE_READ of the whole mapping on all threads, before the clock starts
                else if (args[i] == "--verify") verify = true;  // both forms of every chunk, compared (dcn_pack_ascii as the judge)
            }
            MappedFile mf;
            if (!mf.open(args[1])) die("cannot map " + args[1]);
            const bool fq = mf.data[0] == '@';
#ifdef MADV_POPULATE_READ
            if (prefault) {
                auto tp = std::chrono::steady_clock::now();
                const size_t slice = 8u << 20, n_slices = (mf.size + slice - 1) / slice;
                parallel_for(n_slices, threads, [&](size_t i) {
                    const uintptr_t lo = ((uintptr_t)mf.data + i * slice) & ~(uintptr_t)4095;
                    const uintptr_t hi = ((uintptr_t)mf.data + std::min(mf.size, (i + 1) * slice) + 4095) & ~(uintptr_t)4095;
                    if (madvise((void *)lo, hi - lo, MADV_POPULATE_READ) != 0) std::perror("madvise");
                });
                std::printf("populated %.2f GB in %.3f s on %zu threads\n", mf.size / 1e9,
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count(), threads);
            }
#endif
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::pair<size_t, size_t>> chunks;
            size_t chunk = std::min<size_t>(std::max<size_t>(mf.size / (4 * threads), 4u << 20), 24u << 20);
            if (const char *e = std::getenv("DCN_CLI_CHUNK_KB")) chunk = (size_t)std::max(1, std::atoi(e)) << 10;  // test hook
            for (size_t pos = 0; pos < mf.size;) {
                size_t end = pos + chunk >= mf.size ? mf.size : next_record_start(mf.data, mf.size, pos + chunk, fq);
                chunks.emplace_back(pos, end);
                pos = end;
            }
            if (verify) {
                if (!cli_can_pack()) die("bench-parse --verify: this host cannot run the packing parser");
                size_t n_packed = 0, n_recs = 0;
                for (auto &c : chunks) {
                    Batch pa, as;
                    parse_mapped_chunk(mf.data, c.first, c.second, fq, pa, true);
                    parse_mapped_chunk(mf.data, c.first, c.second, fq, as, false);
                    if (pa.recs.size() != as.recs.size() || pa.offsets != as.offsets) die("verify: records / offsets differ");
                    for (size_t i = 0; i < pa.recs.size(); ++i) {
                        const Rec &x = pa.recs[i], &y = as.recs[i];
                        if (x.id_off != y.id_off || x.id_len != y.id_len || x.seq_len != y.seq_len || x.qual_off != y.qual_off ||
                            x.rec_off != y.rec_off || x.rec_len != y.rec_len || std::memcmp(pa.seq_ptr(x), as.seq_ptr(y), x.seq_len) != 0)
                            die("verify: record " + std::to_string(i) + " differs");
                    }
                    n_recs += pa.recs.size();
                    if (!pa.seq_in_chars) continue;  // (a multi-line record: the chunk fell back to ASCII)
                    ++n_packed;
                    const uint64_t nb = as.bases.size(), groups = std::max<uint64_t>((nb + 31) / 32, 1);
                    std::vector<uint32_t> rp(2 * groups, 0), rm(groups, 0);
                    uint32_t nl = 0;
                    deacon::check(dcn_pack_ascii(as.bases.data(), nb, rp.data(), rm.data(), &nl));
                    if (pa.packed.size() != rp.size() || pa.invmask.size() != rm.size() ||
                        std::memcmp(pa.packed.data(), rp.data(), 4 * rp.size()) != 0 || std::memcmp(pa.invmask.data(), rm.data(), 4 * rm.size()) != 0)
                        die("verify: packed stream differs from dcn_pack_ascii of the ASCII form");
                }
                std::printf("verified %zu records in %zu chunks (%zu packed)\n", n_recs, chunks.size(), n_packed);
                return 0;
            }
            std::atomic<size_t> next{0}, recs{0}, bases{0};
            std::vector<std::thread> pool;
            for (size_t t = 0; t < threads; ++t)
                pool.emplace_back([&] {
                    Batch b;  // recycled, as the pipeline's pool does
                    for (size_t i; (i = next.fetch_add(1)) < chunks.size();) {
                        if (count_only) {
                            recs += count_mapped_records(mf.data, chunks[i].first, chunks[i].second, fq);
                            continue;
                        }
                        b.clear();
                        parse_mapped_chunk(mf.data, chunks[i].first, chunks[i].second, fq, b, packed);
                        recs += b.recs.size();
                        bases += b.offsets.back();
                    }
                });
            for (auto &t : pool) t.join();
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("parsed %zu records, %zu bases, %.2f GB in %.3f s with %zu threads (%s): %.2f GB/s, %.2f Gbp/s\n", recs.load(),
                        bases.load(), mf.size / 1e9, s, threads, packed ? "packed" : "ascii", mf.size / s / 1e9, bases.load() / s / 1e9);
            return 0;
        }
        if (args[0] == "index" && args.size() >= 2 && args[1] == "build") {
            std::string input, output = "-";
            unsigned k = deacon::DEFAULT_KMER_LENGTH, w = deacon::DEFAULT_WINDOW_SIZE;
            size_t cap = 400;
            float entropy = 0.0f;
            bool quiet = false;
            for (size_t i = 2; i < args.size(); ++i) {
                const std::string &s = args[i];
                if (s == "-k") k = (unsigned)std::atoi(need(++i).c_str());
                else if (s == "-w") w = (unsigned)std::atoi(need(++i).c_str());
                else if (s == "-o" || s == "--output") output = need(++i);
                else if (s == "-c" || s == "--capacity") cap = (size_t)std::atoll(need(++i).c_str());
                else if (s == "-t" || s == "--threads") ++i;
                else if (s == "-q" || s == "--quiet") quiet = true;
                else if (s == "-e" || s == "--entropy-threshold") entropy = (float)std::atof(need(++i).c_str());
                else if (s.size() > 1 && s[0] == '-') die("unexpected argument '" + s + "'");
                else input = s;
            }
            if (input.empty()) die("the following required arguments were not provided: <INPUT>");
            if (k < 1 || k > 57) die("invalid value for -k: 1..=57");
            return run_index_build(input, k, w, output, cap, entropy, quiet);
        }
        if (args[0] == "index" && args.size() >= 3 && args[1] == "info") return run_index_info(args[2]);
        if (args[0] == "index" && args.size() >= 3 && (args[1] == "union" || args[1] == "diff")) {
            std::vector<std::string> pos;
            std::string output = "-";
            int k_opt = 0, w_opt = 0;
            for (size_t i = 2; i < args.size(); ++i) {
                const std::string &s = args[i];
                if (s == "-o" || s == "--output") output = need(++i);
                else if (s == "-c" || s == "--capacity") ++i;  // pre-allocation hint only
                else if (s == "-k" || s == "--kmer-length") k_opt = std::atoi(need(++i).c_str());
                else if (s == "-w" || s == "--window-size") w_opt = std::atoi(need(++i).c_str());
                else if (s.size() > 1 && s[0] == '-') die("unexpected argument '" + s + "'");
                else pos.push_back(s);
            }
            if (args[1] == "union") return run_index_union(pos, output);
            if (pos.size() != 2) die("index diff needs <FIRST> <SECOND>");
            return run_index_diff(pos[0], pos[1], k_opt, w_opt, output);
        }
        usage();
        return 2;
    } catch (const deacon::Error &e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}
