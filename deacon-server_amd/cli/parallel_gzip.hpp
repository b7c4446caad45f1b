// parallel_gzip.hpp -- ONE gzip stream inflated on several threads, for the tool's input side.
//
// Why: a .fastq.gz file is one deflate stream; fast_inflate.hpp reads it at 1.3 GB/s of text on one thread while the GPU stage
// behind it takes 150.  BGZF files are inflated member by member (Input::fill_bgzf), but most files are not BGZF.  The
// reference reads through flate2 on one thread (src/local_filter.rs:41-55 of the reference).
//
// How (the two-pass scheme known from pugz and rapidgzip, restated for this decoder):
//   * the compressed bytes are cut into chunks of 2 MB; a worker looks for the first deflate block that starts inside its
//     chunk -- a bit position where a non-final dynamic-Huffman header parses as zlib would accept it (find_block) -- and decodes
//     from there to the first block boundary at or behind the next chunk's start.  What lies before its start is unknown, so
//     it decodes into 16-bit elements whose 32 K predecessors are numbered MARKERS (0x8000 + position in the window): a match
//     that reaches back before the start copies markers, everything else is a byte;
//   * one driver thread walks the stream in order.  It knows the true bit position and the true last 32 KB; a worker's result
//     is taken only if it starts at exactly that position (then it also ends at a true boundary, because decoding from a true
//     boundary is exact), its markers are replaced through the window (resolve, again on the workers), and the window moves on.
//     Where no result fits -- the first chunk, stored / fixed / final blocks the search does not look for, a false candidate,
//     a block longer than a chunk -- the driver decodes the stretch itself with the known window, up to the next boundary a
//     worker started from;
//   * the reader takes the resolved stretches in order, folds their CRC-32s together (crc32_combine) and checks every member's
//     trailer.
// Nothing is trusted that is not re-derived: a stretch is used only if it begins where the exact decode before it ended, and
// every member's CRC-32 and length are checked as in the one-thread reader.  Same bytes, same error texts.
#ifndef DCN_PARALLEL_GZIP_HPP
#define DCN_PARALLEL_GZIP_HPP

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "fast_inflate.hpp"

namespace fastgz {

// ---- the block search ---------------------------------------------------------------------------------------------------------
// First bit position in [from, to) of `base` at which a non-final dynamic block's header is well-formed as far as 74 bits show:
// BFINAL 0, BTYPE 2, HLIT <= 29, HDIST <= 29, and code-length code lengths that form a complete prefix code (zlib refuses any
// other).  ~1 position in 2000 passes; the caller parses the whole header (and decodes) to decide.  `base` is readable PAD
// bytes past `n`.
inline uint64_t find_block(const unsigned char *base, size_t n, uint64_t from, uint64_t to) {
    const uint64_t last = n * 8 > 80 ? n * 8 - 80 : 0;
    if (to > last) to = last;
    for (uint64_t pos = from; pos < to; ++pos) {
        uint64_t w;
        std::memcpy(&w, base + (pos >> 3), 8);
        w >>= pos & 7;  // >= 57 bits
        if ((w & 7) != 4) continue;
        if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;
        const unsigned hclen = (unsigned)((w >> 13) & 15) + 4;
        uint64_t c;
        std::memcpy(&c, base + ((pos + 17) >> 3), 8);
        c >>= (pos + 17) & 7;  // >= 57 bits = 19 lengths
        unsigned kraft = 0;
        for (unsigned i = 0; i < hclen; ++i) {
            const unsigned l = (unsigned)(c >> (3 * i)) & 7;
            kraft += l ? 128u >> l : 0;
        }
        if (kraft != 128) continue;
        return pos;
    }
    return UINT64_MAX;
}

// ---- a plain growable array without value initialisation (tens of MB per chunk) ------------------------------------------------
template <typename T>
struct RawBuf {
    T *p = nullptr;
    size_t cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { std::free(p); }
    void swap(RawBuf &o) {
        std::swap(p, o.p);
        std::swap(cap, o.cap);
    }
    void release() {
        std::free(p);
        p = nullptr;
        cap = 0;
    }
    bool reserve(size_t n) {
        if (n <= cap) return true;
        void *q = std::realloc(p, n * sizeof(T));
        if (!q) return false;
        p = (T *)q;
        cap = n;
        return true;
    }
};

class ParallelGzReader {
  public:
    using Source = GzReader::Source;
    static constexpr size_t WINDOW = 32768;

    ParallelGzReader(Source src, void *ctx, unsigned workers, size_t chunk_bytes = 2u << 20)
        : src_(src), ctx_(ctx), n_workers_(workers < 1 ? 1 : workers), C_(chunk_bytes), OVER_(chunk_bytes / 2) {}
    ~ParallelGzReader() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_task_.notify_all();
        cv_done_.notify_all();
        cv_space_.notify_all();
        if (driver_.joinable()) driver_.join();
        for (auto &t : pool_) t.join();
        if (std::getenv("DCN_CLI_GZ_STATS"))
            std::fprintf(stderr,
                         "gzip reader: %llu stretches of workers taken (%.1f MB), %llu without a block start, %llu not fitting; driver decoded %llu stretches "
                         "(%.1f MB) in %.3f s and waited %.3f s; workers: search %.3f s (%llu candidates), decode %.3f s, resolve %.3f s\n",
                         (unsigned long long)st_.accepted, st_.accepted_bytes / 1e6, (unsigned long long)st_.spec_none, (unsigned long long)st_.spec_wasted,
                         (unsigned long long)st_.direct_calls, st_.direct_bytes / 1e6, st_.direct_us / 1e6, st_.wait_us / 1e6, st_.find_us / 1e6,
                         (unsigned long long)st_.candidates, st_.spec_us / 1e6, st_.resolve_us / 1e6);
    }
    // decompressed bytes; 0 = end of input.  error() is set on a malformed or truncated stream (and 0 is returned).
    size_t read(char *dst, size_t n) {
        if (!started_) {
            started_ = true;
            driver_ = std::thread([this] { drive(); });
        }
        size_t got = 0;
        while (got < n) {
            std::shared_ptr<Segment> seg;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_out_.wait(lk, [&] { return !outq_.empty() && outq_.front()->ready; });
                seg = outq_.front();
            }
            if (seg->kind == Segment::DATA) {
                if (!seg->bad) {
                    if (seg->taken == 0) {
                        crc_ = (uint32_t)::crc32_combine(crc_, seg->crc, (z_off_t)seg->n);
                        isize_ += seg->n;
                    }
                    const size_t take = std::min(n - got, seg->n - seg->taken);
                    std::memcpy(dst + got, seg->bytes.p + seg->off + seg->taken, take);
                    seg->taken += take;
                    got += take;
                    if (seg->taken < seg->n) continue;
                } else {
                    err_ = "invalid gzip stream";
                    seg->kind = Segment::FAILED;
                    break;
                }
            } else if (seg->kind == Segment::MEMBER_END) {
                if (crc_ != seg->want_crc || (uint32_t)isize_ != seg->want_isize) {
                    err_ = "invalid gzip stream";
                    seg->kind = Segment::FAILED;
                    break;
                }
                crc_ = 0;
                isize_ = 0;
            } else if (seg->kind == Segment::FAILED) {
                if (err_.empty()) err_ = seg->err;
                break;  // (stays at the front: every later call ends here too)
            } else {
                break;  // END: stays at the front
            }
            {
                std::lock_guard<std::mutex> g(mu_);
                queued_bytes_ -= seg->n;
                outq_.pop_front();
            }
            byte_pool_.give(seg->bytes);
            cv_space_.notify_all();
        }
        return got;
    }
    const std::string &error() const { return err_; }

  private:
    static constexpr uint64_t NONE = UINT64_MAX;
    static constexpr uint16_t MARK = 0x8000;
    using Dec8 = BlockDecoderT<unsigned char>;
    using Dec16 = BlockDecoderT<uint16_t>;

    struct Chunk {  // compressed bytes [start, start + n) of the stream, n <= C + OVER (a view over several chunks: more)
        uint64_t start = 0;
        size_t n = 0;
        RawBuf<unsigned char> data;  // n + PAD
    };
    struct Spec {  // a worker's decode from the first block it found in its chunk
        std::shared_ptr<Chunk> chunk;
        bool done = false, found = false, final = false;
        uint64_t start_bit = 0, end_bit = 0;  // absolute
        RawBuf<uint16_t> syms;                // WINDOW markers, then n_out elements
        size_t n_out = 0;
    };
    struct Segment {
        enum Kind { DATA, MEMBER_END, END, FAILED } kind = DATA;
        bool ready = false, bad = false;
        RawBuf<unsigned char> bytes;
        size_t off = 0, n = 0, taken = 0;
        uint32_t crc = 0;
        uint32_t want_crc = 0, want_isize = 0;
        std::string err;
        // to be resolved:
        std::shared_ptr<Spec> spec;
        unsigned char window[WINDOW];
        size_t wlen = 0;
    };
    // Buffers go round: a fresh 20-40 MB array per chunk is 10 000 page faults per chunk, more than the decoding of it costs.
    template <typename T>
    struct Pool {
        std::mutex mu;
        std::vector<std::unique_ptr<RawBuf<T>>> free_list;
        void take(RawBuf<T> &into) {
            std::lock_guard<std::mutex> g(mu);
            if (free_list.empty()) return;
            size_t best = 0;
            for (size_t i = 1; i < free_list.size(); ++i)
                if (free_list[i]->cap > free_list[best]->cap) best = i;
            into.swap(*free_list[best]);
            free_list.erase(free_list.begin() + (long)best);
        }
        void give(RawBuf<T> &from) {
            if (!from.p) return;
            auto b = std::make_unique<RawBuf<T>>();
            b->swap(from);
            std::lock_guard<std::mutex> g(mu);
            if (free_list.size() < 32) free_list.push_back(std::move(b));
        }
    };
    Pool<uint16_t> sym_pool_;
    Pool<unsigned char> byte_pool_;
    struct Stop {};  // thrown inside the driver when the reader is being destroyed
    static uint64_t now_us() {
        return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

    // ---- workers -------------------------------------------------------------------------------------------------------------
    void submit(std::function<void()> f, bool urgent) {
        {
            std::lock_guard<std::mutex> g(mu_);
            (urgent ? urgent_ : tasks_).push_back(std::move(f));
            if (pool_.size() < n_workers_ && pool_.size() < urgent_.size() + tasks_.size() + busy_) pool_.emplace_back([this] { work(); });
        }
        cv_task_.notify_one();
    }
    void work() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_task_.wait(lk, [&] { return stop_ || !urgent_.empty() || !tasks_.empty(); });
                if (stop_) return;
                auto &q = urgent_.empty() ? tasks_ : urgent_;
                f = std::move(q.front());
                q.pop_front();
                ++busy_;
            }
            f();
            {
                std::lock_guard<std::mutex> g(mu_);
                --busy_;
            }
        }
    }

    // the speculative decode of one chunk (on a worker)
    void run_spec(const std::shared_ptr<Spec> &sp) {
        static thread_local std::unique_ptr<Dec16> dec;
        if (!dec) dec.reset(new Dec16());
        const Chunk &ck = *sp->chunk;
        const unsigned char *base = ck.data.p, *in_end = base + ck.n;
        const uint64_t nominal_end = (uint64_t)std::min(C_, ck.n) * 8;  // a block that starts behind it belongs to the next chunk
        const bool more_behind = ck.n > C_;
        int tries = 0;
        const uint64_t t0 = now_us();
        uint64_t t_find = 0;
        auto timed_find = [&](uint64_t from) {
            const uint64_t a = now_us();
            const uint64_t r = find_block(base, ck.n, from, nominal_end);
            t_find += now_us() - a;
            return r;
        };
        for (uint64_t cand = timed_find(0); cand != NONE && tries < 200; cand = timed_find(cand + 1), ++tries) {
            if (stopping()) break;
            ++st_.candidates;
            if (tries == 0) sym_pool_.take(sp->syms);
            size_t cap = std::max<size_t>(sp->syms.cap, WINDOW + 6 * C_ + Dec16::SLACK);
            if (!sp->syms.reserve(cap)) break;
            if (tries == 0)
                for (size_t i = 0; i < WINDOW; ++i) sp->syms.p[i] = (uint16_t)(MARK + i);
            uint16_t *o = sp->syms.p + WINDOW;
            Dec16::Bits b;
            Dec16::start_at_bit(b, base, cand);
            uint64_t pos = cand;
            size_t blocks = 0;
            bool bad = false, fin = false;
            for (;;) {
                bool f = false;
                const Status st = dec->block(b, in_end, sp->syms.p, o, sp->syms.p + sp->syms.cap, f);
                if (st == OK) {
                    ++blocks;
                    pos = Dec16::bit_position(b, base);
                    if (f) {
                        fin = true;
                        break;
                    }
                    if (more_behind && pos >= nominal_end) break;
                    continue;
                }
                if (st == NEED_ROOM) {
                    const size_t at = (size_t)(o - sp->syms.p);
                    if (sp->syms.cap >= WINDOW + 24 * C_ || !sp->syms.reserve(sp->syms.cap * 2)) break;  // (what there is is handed over; the driver goes on from its end)
                    o = sp->syms.p + at;
                    continue;
                }
                if (st == BAD) bad = true;
                break;  // NEED_INPUT: a block that runs past the chunk's overlap (or the end of the stream); up to its start the result stands
            }
            if (bad) continue;  // not a block start after all: look further
            if (blocks == 0) break;
            sp->found = true;
            sp->final = fin;
            sp->start_bit = ck.start * 8 + cand;
            sp->end_bit = ck.start * 8 + pos;
            sp->n_out = (size_t)(o - (sp->syms.p + WINDOW));
            break;
        }
        st_.find_us += t_find;
        st_.spec_us += now_us() - t0 - t_find;
        if (!sp->found) {
            ++st_.spec_none;
            sym_pool_.give(sp->syms);
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            sp->done = true;
        }
        cv_done_.notify_all();
    }

    // markers -> bytes through the window that was in front of the stretch (on a worker)
    void run_resolve(const std::shared_ptr<Segment> &seg) {
        Spec &sp = *seg->spec;
        const size_t n = sp.n_out;
        const uint64_t t0 = now_us();
        byte_pool_.take(seg->bytes);
        bool bad = !seg->bytes.reserve(n + 16);
        if (!bad) {
            const uint16_t *in = sp.syms.p + WINDOW;
            unsigned char *out = seg->bytes.p;
            // marker MARK + j = the byte j of the 32 KB in front; of those only the last wlen exist
            const unsigned char *win = seg->window;  // right-aligned: window[WINDOW - wlen, WINDOW)
            const unsigned first_valid = (unsigned)(WINDOW - seg->wlen);
            unsigned lowest = WINDOW;
            size_t i = 0;
            if (first_valid == 0) {
                // the usual case, every marker has its byte: one table from element to byte (bytes map to themselves), no tests.
                // Sixteen elements that are all bytes -- where the data behind the start no longer reaches back -- are packed at once.
                std::unique_ptr<unsigned char[]> lut(new unsigned char[65536]);
                for (unsigned v = 0; v < 256; ++v) lut[v] = (unsigned char)v;
                std::memcpy(lut.get() + MARK, win, WINDOW);
#if defined(__x86_64__)
                const __m128i hi = _mm_set1_epi16((short)0xFF00), ramp = _mm_setr_epi16(0, 1, 2, 3, 4, 5, 6, 7);
                for (; i + 16 <= n; i += 16) {
                    const __m128i a = _mm_loadu_si128((const __m128i *)(in + i)), b = _mm_loadu_si128((const __m128i *)(in + i + 8));
                    if (_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_and_si128(_mm_or_si128(a, b), hi), _mm_setzero_si128())) == 0xFFFF) {
                        _mm_storeu_si128((__m128i *)(out + i), _mm_packus_epi16(a, b));
                        continue;
                    }
                    // sixteen markers in a row that name sixteen bytes in a row -- a match copied from before the start, which is
                    // how markers come to be here at all (quality lines repeating the one above): sixteen bytes of the window
                    const unsigned m0 = in[i];
                    if (m0 >= MARK && m0 <= 0xFFFFu - 15) {
                        const __m128i e0 = _mm_add_epi16(_mm_set1_epi16((short)m0), ramp);
                        const __m128i e1 = _mm_add_epi16(e0, _mm_set1_epi16(8));
                        if (_mm_movemask_epi8(_mm_and_si128(_mm_cmpeq_epi16(a, e0), _mm_cmpeq_epi16(b, e1))) == 0xFFFF) {
                            _mm_storeu_si128((__m128i *)(out + i), _mm_loadu_si128((const __m128i *)(win + (m0 - MARK))));
                            continue;
                        }
                    }
                    for (size_t q = i; q < i + 16; ++q) out[q] = lut[in[q]];
                }
#endif
                for (; i < n; ++i) out[i] = lut[in[i]];
            }
            for (; i < n; ++i) {  // near the start of a member: a marker may name a byte that does not exist
                const unsigned s = in[i];
                if (s < 256) out[i] = (unsigned char)s;
                else {
                    const unsigned j = s & (WINDOW - 1);
                    if (j < lowest) lowest = j;
                    out[i] = win[j];
                }
            }
            if (lowest < first_valid) bad = true;  // a match that reaches back before the start of its member
            seg->crc = crc32_fast(0, out, n);
        }
        seg->off = 0;
        seg->n = n;
        sym_pool_.give(sp.syms);
        seg->spec.reset();
        st_.resolve_us += now_us() - t0;
        {
            std::lock_guard<std::mutex> g(mu_);
            seg->bad = bad;
            seg->ready = true;
        }
        cv_out_.notify_all();
    }

    bool stopping() {
        std::lock_guard<std::mutex> g(mu_);
        return stop_;
    }

    // ---- the compressed stream, chunk by chunk (driver only) -------------------------------------------------------------------
    // reads chunk `idx` (and every chunk before it); false: the stream ends before that chunk
    bool ensure_read(size_t idx) {
        while (next_read_ <= idx) {
            if (eof_ && tail_.empty()) return false;
            auto ck = std::make_shared<Chunk>();
            ck->start = (uint64_t)next_read_ * C_;
            if (!ck->data.reserve(C_ + OVER_ + PAD)) throw std::bad_alloc();
            if (!tail_.empty()) std::memcpy(ck->data.p, tail_.data(), tail_.size());
            size_t n = tail_.size();
            while (!eof_ && n < C_ + OVER_) {
                const size_t r = src_(ctx_, ck->data.p + n, C_ + OVER_ - n);
                if (r == 0) eof_ = true;
                else n += r;
            }
            ck->n = n;
            std::memset(ck->data.p + n, 0, PAD);
            tail_.assign(ck->data.p + std::min(n, C_), ck->data.p + n);  // what belongs to the next chunk as well
            if (eof_) stream_len_ = ck->start + n;
            if (n == 0) return false;  // (the stream ended exactly at a chunk boundary)
            chunks_[next_read_++] = ck;
        }
        return chunks_.count(idx) != 0;
    }
    // chunks idx .. idx + count - 1 as one piece of memory (count == 1: the chunk itself)
    std::shared_ptr<Chunk> view(size_t idx, size_t count) {
        if (!ensure_read(idx)) return nullptr;
        if (count == 1) return chunks_[idx];
        auto v = std::make_shared<Chunk>();
        v->start = (uint64_t)idx * C_;
        if (!v->data.reserve(count * C_ + OVER_ + PAD)) throw std::bad_alloc();
        size_t n = 0;
        for (size_t k = 0; k < count; ++k) {
            if (!ensure_read(idx + k)) break;
            const Chunk &c = *chunks_[idx + k];
            n = k * C_;
            std::memcpy(v->data.p + n, c.data.p, c.n);
            n += c.n;
        }
        v->n = n;
        std::memset(v->data.p + n, 0, PAD);
        return v;
    }
    bool known_end(uint64_t byte_off) { return eof_ && byte_off >= stream_len_; }

    // ---- the driver ------------------------------------------------------------------------------------------------------------
    void push(const std::shared_ptr<Segment> &seg) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_space_.wait(lk, [&] { return stop_ || queued_bytes_ < OUT_LIMIT || outq_.empty(); });
        if (stop_) throw Stop();
        queued_bytes_ += seg->n;
        outq_.push_back(seg);
        lk.unlock();
        cv_out_.notify_all();
    }
    void push_mark(typename Segment::Kind kind, const std::string &err = std::string(), uint32_t crc = 0, uint32_t isize = 0) {
        auto seg = std::make_shared<Segment>();
        seg->kind = kind;
        seg->ready = true;
        seg->err = err;
        seg->want_crc = crc;
        seg->want_isize = isize;
        push(seg);
    }
    void lookahead(size_t j) {
        const size_t ahead = 2 * n_workers_ + 1;
        while (next_read_ <= j + ahead && !(eof_ && tail_.empty())) {
            const size_t idx = next_read_;
            if (!ensure_read(idx)) break;
            if (idx == 0) continue;
            auto sp = std::make_shared<Spec>();
            sp->chunk = chunks_[idx];
            specs_[idx] = sp;
            submit([this, sp] { run_spec(sp); }, false);
        }
        while (!chunks_.empty() && chunks_.begin()->first + 1 < j) chunks_.erase(chunks_.begin());
        while (!specs_.empty() && specs_.begin()->first < j) specs_.erase(specs_.begin());
    }
    void wait_done(const std::shared_ptr<Spec> &sp) {
        const uint64_t t0 = now_us();
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return stop_ || sp->done; });
        st_.wait_us += now_us() - t0;
        if (stop_) throw Stop();
    }
    void slide_window(const unsigned char *bytes, size_t n) {  // the last 32 KB of (window ++ bytes)
        if (n >= WINDOW) {
            std::memcpy(window_, bytes + n - WINDOW, WINDOW);
            wlen_ = WINDOW;
        } else {
            const size_t keep = std::min(wlen_, WINDOW - n);
            std::memmove(window_ + WINDOW - n - keep, window_ + WINDOW - keep, keep);
            std::memcpy(window_ + WINDOW - n, bytes, n);
            wlen_ = keep + n;
        }
    }

    // the driver's own decode from the true position `pos` with the true window: up to the first block boundary at or behind
    // `stop`, or equal to `alt` (where a worker started), or the end of the member.  Returns false on end of input inside a block.
    bool direct(uint64_t &pos, uint64_t stop, uint64_t alt, bool &final, const char *&why) {
        const size_t j = (size_t)((pos >> 3) / C_);
        const uint64_t t0 = now_us();
        for (size_t count = 1;; count *= 2) {
            std::shared_ptr<Chunk> ck = view(j, count);
            if (!ck) {
                why = "truncated gzip stream";
                return false;
            }
            const unsigned char *base = ck->data.p, *in_end = base + ck->n;
            auto seg = std::make_shared<Segment>();
            byte_pool_.take(seg->bytes);
            size_t cap = WINDOW + 8 * (size_t)((std::min<uint64_t>(stop, (ck->start + ck->n) * 8) - pos) / 8 + 8192) + Dec8::SLACK;
            if (!seg->bytes.reserve(cap)) throw std::bad_alloc();
            std::memcpy(seg->bytes.p + WINDOW - wlen_, window_ + WINDOW - wlen_, wlen_);
            unsigned char *o = seg->bytes.p + WINDOW;
            Dec8::Bits b;
            Dec8::start_at_bit(b, base, pos - ck->start * 8);
            size_t blocks = 0;
            uint64_t at = pos;
            Status st = OK;
            bool fin = false;
            for (;;) {
                bool f = false;
                st = dec8_->block(b, in_end, seg->bytes.p + WINDOW - wlen_, o, seg->bytes.p + seg->bytes.cap, f);
                if (st == OK) {
                    ++blocks;
                    at = ck->start * 8 + Dec8::bit_position(b, base);
                    if (f) {
                        fin = true;
                        break;
                    }
                    if (at >= stop || at == alt || (size_t)(o - seg->bytes.p) > DIRECT_SEGMENT) break;
                    continue;
                }
                if (st == NEED_ROOM) {
                    const size_t k = (size_t)(o - seg->bytes.p);
                    if (!seg->bytes.reserve(seg->bytes.cap * 2)) throw std::bad_alloc();
                    o = seg->bytes.p + k;
                    continue;
                }
                break;
            }
            if (st == BAD) {
                why = "invalid gzip stream";
                return false;
            }
            if (st == NEED_INPUT && blocks == 0) {
                if (known_end(ck->start + ck->n)) {
                    why = "truncated gzip stream";
                    return false;
                }
                continue;  // a block longer than what the view holds: take more chunks together
            }
            seg->off = WINDOW;
            seg->n = (size_t)(o - (seg->bytes.p + WINDOW));
            seg->crc = crc32_fast(0, seg->bytes.p + WINDOW, seg->n);
            seg->ready = true;
            slide_window(seg->bytes.p + WINDOW, seg->n);
            pos = at;
            final = fin;
            ++st_.direct_calls;
            st_.direct_bytes += seg->n;
            st_.direct_us += now_us() - t0;
            if (seg->n) push(seg);
            return true;
        }
    }

    // a worker's stretch that starts at the true position: hand it to the resolvers, move the window over it
    void accept(const std::shared_ptr<Spec> &sp) {
        auto seg = std::make_shared<Segment>();
        seg->spec = sp;
        seg->wlen = wlen_;
        std::memcpy(seg->window, window_, WINDOW);
        seg->n = sp->n_out;  // (for the queue's accounting; run_resolve sets it again)
        // the window behind the stretch: its last 32 K elements through the window in front of it
        const uint16_t *in = sp->syms.p + WINDOW;
        const size_t n = sp->n_out, tail = std::min(n, WINDOW);
        unsigned char last[WINDOW];
        for (size_t i = 0; i < tail; ++i) {
            const unsigned s = in[n - tail + i];
            last[i] = s < 256 ? (unsigned char)s : window_[s & (WINDOW - 1)];
        }
        slide_window(last, tail);
        push(seg);
        submit([this, seg] { run_resolve(seg); }, true);
    }

    void drive() {
        try {
            dec8_.reset(new Dec8());
            uint64_t cursor = 0;  // byte position of the next member's header
            for (;;) {
                // ---- header
                size_t at = 0;
                {
                    Status hs = NEED_INPUT;
                    const size_t j = (size_t)(cursor / C_);
                    for (size_t count = 1; hs == NEED_INPUT; count *= 2) {
                        std::shared_ptr<Chunk> ck = view(j, count);
                        if (!ck || cursor >= ck->start + ck->n) {
                            if (cursor == 0 || !known_end(cursor)) return push_mark(Segment::FAILED, "truncated gzip stream");
                            return push_mark(Segment::END);
                        }
                        hs = gzip_header(ck->data.p + (cursor - ck->start), (size_t)(ck->start + ck->n - cursor), at);
                        if (hs == NEED_INPUT && known_end(ck->start + ck->n)) return push_mark(Segment::FAILED, "truncated gzip stream");
                    }
                    if (hs == BAD) return push_mark(Segment::FAILED, "invalid gzip stream");
                }
                uint64_t pos = (cursor + at) * 8;
                wlen_ = 0;
                // ---- blocks
                for (bool final = false; !final;) {
                    const size_t j = (size_t)((pos >> 3) / C_);
                    lookahead(j);
                    std::shared_ptr<Spec> sp;
                    auto it = specs_.find(j);
                    if (it != specs_.end()) {
                        sp = it->second;
                        wait_done(sp);
                        if (!sp->found || sp->start_bit < pos) {
                            if (sp->found) ++st_.spec_wasted;
                            specs_.erase(it);
                            sp.reset();
                        }
                    }
                    if (sp && sp->start_bit == pos) {
                        accept(sp);
                        ++st_.accepted;
                        st_.accepted_bytes += sp->n_out;
                        pos = sp->end_bit;
                        final = sp->final;
                        specs_.erase(j);
                        continue;
                    }
                    const char *why = "";
                    if (!direct(pos, (uint64_t)(j + 1) * C_ * 8, sp ? sp->start_bit : NONE, final, why)) return push_mark(Segment::FAILED, why);
                }
                // ---- trailer
                cursor = (pos + 7) >> 3;
                {
                    const size_t j = (size_t)(cursor / C_);
                    std::shared_ptr<Chunk> ck = view(j, 1);
                    if (ck && cursor + 8 > ck->start + ck->n && !known_end(ck->start + ck->n)) ck = view(j, 2);  // (the overlap is half a chunk: two hold it)
                    if (!ck || cursor + 8 > ck->start + ck->n) return push_mark(Segment::FAILED, "truncated gzip stream");
                    const unsigned char *p = ck->data.p + (cursor - ck->start);
                    const uint32_t crc = p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
                    const uint32_t len = p[4] | (uint32_t)p[5] << 8 | (uint32_t)p[6] << 16 | (uint32_t)p[7] << 24;
                    push_mark(Segment::MEMBER_END, std::string(), crc, len);
                    cursor += 8;
                    ensure_read((size_t)(cursor / C_));
                    if (known_end(cursor)) return push_mark(Segment::END);
                }
            }
        } catch (const Stop &) {
        } catch (const std::exception &e) {
            try {
                push_mark(Segment::FAILED, std::string("gzip reader: ") + e.what());
            } catch (...) {
            }
        }
    }

    static constexpr size_t OUT_LIMIT = 512u << 20, DIRECT_SEGMENT = 64u << 20;
    Source src_;
    void *ctx_;
    const size_t n_workers_, C_, OVER_;
    // driver state
    std::map<size_t, std::shared_ptr<Chunk>> chunks_;
    std::map<size_t, std::shared_ptr<Spec>> specs_;
    std::vector<unsigned char> tail_;
    size_t next_read_ = 0;
    bool eof_ = false;
    uint64_t stream_len_ = 0;
    unsigned char window_[WINDOW];
    size_t wlen_ = 0;
    std::unique_ptr<Dec8> dec8_;
    // shared
    std::mutex mu_;
    std::condition_variable cv_task_, cv_done_, cv_out_, cv_space_;
    std::deque<std::function<void()>> tasks_, urgent_;
    std::deque<std::shared_ptr<Segment>> outq_;
    size_t queued_bytes_ = 0, busy_ = 0;
    bool stop_ = false;
    std::vector<std::thread> pool_;
    std::thread driver_;
    // counters (DCN_CLI_GZ_STATS=1 prints them when the reader goes)
    struct Stats {
        std::atomic<uint64_t> accepted{0}, accepted_bytes{0}, direct_calls{0}, direct_bytes{0}, spec_none{0}, spec_wasted{0}, find_us{0}, spec_us{0}, resolve_us{0},
            direct_us{0}, wait_us{0}, candidates{0};
    } st_;
    // reader state
    bool started_ = false;
    uint32_t crc_ = 0;
    uint64_t isize_ = 0;
    std::string err_;
};

}  // namespace fastgz
#endif  // DCN_PARALLEL_GZIP_HPP
