// deacon_hip.hpp -- header-only C++ host layer over the C ABI of deacon_hip.h.
//
// Mirrors the reference's filter interface (crate deacon 0.10.0) so that host code reads like the Rust it
// replaces; names, argument meaning and error behaviour follow the cited items (paths under the reference's src/):
//
//   deacon::Index                       index::load_minimizer_hashes + IndexHeader      index.rs:17-54, 80-107
//   deacon::FilterConfig                FilterConfig defaults                            lib.rs:90-109
//   deacon::FilterProcessor             local_filter::FilterProcessor                    local_filter.rs:153-285
//     .should_keep_sequence(seq)        -> (keep, hit_count, num_minimizers)             local_filter.rs:221-252
//     .should_keep_pair(seq1, seq2)                                                      local_filter.rs:254-285
//     .filter_batch(reads[, paired])    the paraseq per-record loop over one batch       local_filter.rs:346-528
//     .stats()                          ProcessingStats                                  local_filter.rs:179-187
//   deacon::MultiGpuFilter              the worker pool of run(): N workers, one set,    local_filter.rs:376-405, 630-631,
//                                       outputs merged in order, counters summed           696-709
//   deacon::get_minimizer_hashes_and_positions                                           filter_common.rs:211-310
//   deacon::unpaired_should_keep / paired_should_keep                                    remote_filter.rs:230-301
//
// Errors: the reference returns anyhow::Result up to main; here every failing C call throws deacon::Error carrying
// the code and dcn_last_error().  Nothing in this header computes: all arithmetic happens in libdeacon_hip.so.
#ifndef DEACON_HIP_HPP
#define DEACON_HIP_HPP

#include <algorithm>
#include <array>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "deacon_hip.h"

namespace deacon {

constexpr uint8_t DEFAULT_KMER_LENGTH = 31;  // minimizers.rs:4
constexpr uint8_t DEFAULT_WINDOW_SIZE = 15;  // minimizers.rs:5

class Error : public std::runtime_error {
  public:
    Error(int code, const std::string &what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }

  private:
    int code_;
};

inline void check(int rc) {
    if (rc != DCN_OK) throw Error(rc, std::string("deacon_hip error ") + std::to_string(rc) + ": " + dcn_last_error());
}

// (keep, hit_count, num_minimizers): the tuple should_keep_* returns, minus the debug k-mer strings
using Decision = std::tuple<bool, std::size_t, std::size_t>;

struct IndexHeader {  // index.rs:17-22
    uint8_t format_version = 2;
    uint8_t kmer_length = DEFAULT_KMER_LENGTH;
    uint8_t window_size = DEFAULT_WINDOW_SIZE;
};

class Index {
  public:
    // index.rs:80-107: load a deacon index file into the device-resident set
    static Index load(const std::string &path, int device = 0) {
        dcn_index *h = nullptr;
        check(dcn_index_from_file(path.c_str(), device, &h));
        return Index(h);
    }
    // the same from minimizer hashes already in memory (what index::build holds before write_minimizers)
    static Index from_hashes(const std::vector<uint64_t> &hashes, uint8_t kmer_length = DEFAULT_KMER_LENGTH,
                             uint8_t window_size = DEFAULT_WINDOW_SIZE, int device = 0) {
        dcn_index *h = nullptr;
        check(dcn_index_from_keys(hashes.data(), hashes.size(), kmer_length, window_size, device, &h));
        return Index(h);
    }
    Index(Index &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    Index &operator=(Index &&o) noexcept {
        if (this != &o) {
            reset();
            h_ = std::exchange(o.h_, nullptr);
        }
        return *this;
    }
    Index(const Index &) = delete;
    Index &operator=(const Index &) = delete;
    ~Index() { reset(); }

    IndexHeader header() const {
        IndexHeader hd;
        check(dcn_index_header(h_, &hd.kmer_length, &hd.window_size, nullptr));
        return hd;
    }
    uint64_t len() const {  // number of distinct minimizers (`deacon index info`, index.rs:539-560)
        uint64_t n = 0;
        check(dcn_index_header(h_, nullptr, nullptr, &n));
        return n;
    }
    std::vector<bool> contains(const std::vector<uint64_t> &hashes) const {  // FxHashSet::contains
        std::vector<uint8_t> out(hashes.size());
        check(dcn_index_contains(h_, hashes.data(), hashes.size(), out.data()));
        return std::vector<bool>(out.begin(), out.end());
    }
    // replica on another (or the same) device, copied device to device: the reference shares ONE set between its
    // workers (local_filter.rs:630-631); a multi-GPU host loads it once and clones it
    Index clone(int device) const {
        dcn_index *h = nullptr;
        check(dcn_index_clone(h_, device, &h));
        return Index(h);
    }
    int device() const {
        int d = 0;
        check(dcn_index_device(h_, &d));
        return d;
    }
    uint64_t table_bytes() const {
        uint64_t b = 0;
        check(dcn_index_memory(h_, &b));
        return b;
    }
    const dcn_index *raw() const { return h_; }

  private:
    explicit Index(dcn_index *h) : h_(h) {}
    void reset() {
        if (h_) dcn_index_destroy(h_);
        h_ = nullptr;
    }
    dcn_index *h_ = nullptr;
};

struct FilterConfig {  // the decision-relevant fields of lib.rs:39-87 with their defaults (lib.rs:90-109)
    std::size_t abs_threshold = 2;
    double rel_threshold = 0.01;
    std::size_t prefix_length = 0;
    bool deplete = false;
    // sizing of the device pipeline (no counterpart in the reference)
    uint64_t max_batch_bases = 1ull << 26;
    uint32_t max_batch_reads = 1u << 20;
};

struct ProcessingStats {  // local_filter.rs:179-187
    uint64_t total_seqs = 0, filtered_seqs = 0, total_bp = 0, output_bp = 0, filtered_bp = 0, output_seq_counter = 0;
};

class FilterProcessor {
  public:
    FilterProcessor(const Index &index, const FilterConfig &config) : config_(config) {
        check(dcn_ctx_create(index.raw(), config.max_batch_bases, config.max_batch_reads, &ctx_));
    }
    FilterProcessor(const FilterProcessor &) = delete;
    FilterProcessor &operator=(const FilterProcessor &) = delete;
    ~FilterProcessor() {
        if (ctx_) dcn_ctx_destroy(ctx_);
    }

    // local_filter.rs:221-252
    Decision should_keep_sequence(std::string_view seq) {
        auto r = filter_batch({seq}, false);
        return r.at(0);
    }
    // local_filter.rs:254-285: mate 1 then mate 2, hits distinct across both mates, one decision
    Decision should_keep_pair(std::string_view seq1, std::string_view seq2) {
        auto r = filter_batch({seq1, seq2}, true);
        return r.at(0);
    }
    // One batch of records; paired: reads 2i and 2i+1 are the mates of pair i (a trailing single read is its own
    // unit).  Returns one Decision per unit, in input order.
    std::vector<Decision> filter_batch(const std::vector<std::string_view> &reads, bool paired) {
        bases_.clear();
        offsets_.assign(1, 0);
        unit_id_.clear();
        for (std::size_t i = 0; i < reads.size(); ++i) {
            bases_.insert(bases_.end(), reads[i].begin(), reads[i].end());
            offsets_.push_back(bases_.size());
            if (paired) unit_id_.push_back(static_cast<uint32_t>(i / 2));
        }
        std::size_t n_units = paired ? (reads.size() + 1) / 2 : reads.size();
        keep_.assign(n_units, 0);
        hits_.assign(n_units, 0);
        total_.assign(n_units, 0);
        dcn_params p = params();
        check(dcn_filter_batch(ctx_, bases_.data(), offsets_.data(), paired ? unit_id_.data() : nullptr,
                               static_cast<uint32_t>(reads.size()), &p, keep_.data(), hits_.data(), total_.data()));
        std::vector<Decision> out(n_units);
        for (std::size_t u = 0; u < n_units; ++u) out[u] = Decision(keep_[u] != 0, hits_[u], total_[u]);
        return out;
    }
    // Decisions only: what the filter loop consumes outside --debug (local_filter.rs:350-371).  No hit counts are
    // returned, so the kernels stop probing a unit once its decision is fixed.  One bool per unit, in input order.
    std::vector<bool> keep_batch(const std::vector<std::string_view> &reads, bool paired) {
        bases_.clear();
        offsets_.assign(1, 0);
        unit_id_.clear();
        for (std::size_t i = 0; i < reads.size(); ++i) {
            bases_.insert(bases_.end(), reads[i].begin(), reads[i].end());
            offsets_.push_back(bases_.size());
            if (paired) unit_id_.push_back(static_cast<uint32_t>(i / 2));
        }
        std::size_t n_units = paired ? (reads.size() + 1) / 2 : reads.size();
        keep_.assign(n_units, 0);
        dcn_params p = params();
        check(dcn_filter_batch(ctx_, bases_.data(), offsets_.data(), paired ? unit_id_.data() : nullptr,
                               static_cast<uint32_t>(reads.size()), &p, keep_.data(), nullptr, nullptr));
        return std::vector<bool>(keep_.begin(), keep_.end());
    }
    // Zero-copy form of the batch seam: concatenated bases + offsets (+ optional unit ids), outputs per unit.
    // hits and total may both be null (decisions only).
    void filter_batch(const uint8_t *bases, const uint64_t *offsets, const uint32_t *unit_id, uint32_t n_reads,
                      uint8_t *keep, uint32_t *hits, uint32_t *total) {
        dcn_params p = params();
        check(dcn_filter_batch(ctx_, bases, offsets, unit_id, n_reads, &p, keep, hits, total));
    }

    // ---- the shape of `impl ParallelProcessor for FilterProcessor` (local_filter.rs:345-405, pairs :482-573) ---------------
    // paraseq calls process_record for every record of a record set (1,024 records by default) and then
    // on_batch_complete; the record is only borrowed during the call.  A GPU call costs ~100 us whatever it carries
    // (INTEGRATION.md 3.2), so this processor copies what it is handed, lets record sets GATHER, decides `flush_reads`
    // of them in one call, and hands every record to the sink with its unit's decision, in input order -- where the
    // reference formats kept records into its thread-local buffer and writes it under the lock (:365-367, :376-405).
    struct Record {
        std::string id, seq, qual;  // (qual empty: FASTA)
    };
    void set_flush_reads(std::size_t n) { flush_reads_ = n < 1 ? 1 : n; }  // default 1 << 16
    void process_record(std::string_view id, std::string_view seq, std::string_view qual = {}) {  // :346-374
        gathered_.push_back(Record{std::string(id), std::string(seq), std::string(qual)});
        gathered_paired_ = false;
    }
    // :483-528 (two readers) and :409-448 (interleaved): mate 1 then mate 2 of one unit
    void process_record_pair(std::string_view id1, std::string_view seq1, std::string_view qual1, std::string_view id2, std::string_view seq2,
                             std::string_view qual2) {
        gathered_.push_back(Record{std::string(id1), std::string(seq1), std::string(qual1)});
        gathered_.push_back(Record{std::string(id2), std::string(seq2), std::string(qual2)});
        gathered_paired_ = true;
    }
    // sink(const Record &, bool keep): every gathered record in input order (both mates of a pair get the pair's decision)
    template <typename Sink>
    void on_batch_complete(Sink &&sink) {  // :376-405 -- here: only once enough has gathered
        if (gathered_.size() >= flush_reads_) flush(sink);
    }
    template <typename Sink>
    void on_thread_complete(Sink &&sink) {  // what is left when the reader is done
        flush(sink);
    }
    template <typename Sink>
    void flush(Sink &&sink) {
        if (gathered_.empty()) return;
        // (max_batch_reads / max_batch_bases of the context bound a call: larger gatherings go in several)
        std::size_t at = 0;
        while (at < gathered_.size()) {
            std::vector<std::string_view> views;
            uint64_t bases = 0;
            std::size_t end = at;
            const std::size_t step = gathered_paired_ ? 2 : 1;
            while (end < gathered_.size() && views.size() + step <= config_.max_batch_reads) {
                uint64_t add = 0;
                for (std::size_t j = 0; j < step; ++j) add += gathered_[end + j].seq.size();
                if (!views.empty() && bases + add > config_.max_batch_bases) break;
                for (std::size_t j = 0; j < step; ++j) views.emplace_back(gathered_[end + j].seq);
                bases += add;
                end += step;
            }
            const std::vector<bool> keep = keep_batch(views, gathered_paired_);
            for (std::size_t i = at; i < end; ++i) sink(gathered_[i], bool(keep[(i - at) / step]));
            at = end;
        }
        gathered_.clear();
    }

    ProcessingStats stats() {
        std::array<uint64_t, DCN_N_STATS> c{};
        check(dcn_ctx_stats(ctx_, c.data()));
        ProcessingStats s;
        s.total_seqs = c[DCN_STAT_TOTAL_SEQS];
        s.filtered_seqs = c[DCN_STAT_FILTERED_SEQS];
        s.total_bp = c[DCN_STAT_TOTAL_BP];
        s.output_bp = c[DCN_STAT_OUTPUT_BP];
        s.filtered_bp = c[DCN_STAT_FILTERED_BP];
        s.output_seq_counter = c[DCN_STAT_OUTPUT_SEQ_COUNTER];
        return s;
    }

    FilterConfig &config() { return config_; }
    dcn_ctx *raw() { return ctx_; }

    dcn_params params() const {
        dcn_params p;
        p.abs_threshold = config_.abs_threshold;
        p.rel_threshold = config_.rel_threshold;
        p.prefix_length = config_.prefix_length;
        p.deplete = config_.deplete ? 1u : 0u;
        p.reserved = 0;
        return p;
    }

  private:
    FilterConfig config_;
    dcn_ctx *ctx_ = nullptr;
    std::vector<uint8_t> bases_, keep_;
    std::vector<uint64_t> offsets_;
    std::vector<uint32_t> unit_id_, hits_, total_;
    std::vector<Record> gathered_;
    bool gathered_paired_ = false;
    std::size_t flush_reads_ = std::size_t(1) << 16;
};

// The reference's run() hands record batches to N worker threads that share one set, merges their output buffers
// under a mutex and sums their ProcessingStats (local_filter.rs:376-405, 630-631, 696-709).  Here a worker is a
// GPU pipeline context: one host thread + dcn_ctx per entry of `devices` (a device may be listed more than once:
// {0, 0} runs two contexts on GPU 0), the index loaded once and replicated device to device, batches dealt
// round-robin by sequence number -- a batch holds whole units, so pairs stay intact and any batch->worker map gives
// the decisions of the single-context run.  Each worker keeps two batches in flight (dcn_filter_batch_submit /
// _wait), so copies of one overlap kernels of the other.  Results land in the caller's arrays; wait(seq) in
// sequence order is the ordered merge.  No collective: all contexts live in this process, the counters are summed
// on the host (dcn_stats_allreduce); separate processes reduce them with RCCL instead (bench.py).
class MultiGpuFilter {
  public:
    struct Job {  // one batch: the arguments of dcn_filter_batch; every array must stay valid until wait(seq) returns
        const uint8_t *bases = nullptr;
        // ... or, instead of `bases`, the stream as dcn_filter_batch_packed takes it (both set, `bases` null)
        const uint32_t *packed = nullptr, *invmask = nullptr;
        const uint64_t *offsets = nullptr;
        const uint32_t *unit_id = nullptr;
        uint32_t n_reads = 0;
        uint8_t *keep = nullptr;
        uint32_t *hits = nullptr, *total = nullptr;
    };

    MultiGpuFilter(const Index &index, const std::vector<int> &devices, const FilterConfig &config,
                   std::size_t queue_depth = 4)
        : config_(config), queue_depth_(queue_depth) {
        if (devices.empty()) throw Error(DCN_ERR_ARG, "MultiGpuFilter: no devices");
        // one table per distinct device: the caller's index (which must outlive this object) serves its own device,
        // every other device gets a replica copied device to device
        std::map<int, const dcn_index *> by_device;
        by_device[index.device()] = index.raw();
        for (int d : devices) {
            if (by_device.count(d)) continue;
            replicas_.push_back(index.clone(d));
            by_device[d] = replicas_.back().raw();
        }
        for (std::size_t i = 0; i < devices.size(); ++i) workers_.emplace_back(new Worker());
        // every context first, the threads only once nothing can throw any more (a constructor that throws with a
        // thread already running would end in std::terminate)
        for (std::size_t i = 0; i < devices.size(); ++i) {
            int rc = dcn_ctx_create(by_device[devices[i]], config.max_batch_bases, config.max_batch_reads, &workers_[i]->ctx);
            if (rc != DCN_OK) {
                const std::string msg = std::string("deacon_hip error ") + std::to_string(rc) + ": " + dcn_last_error();
                for (auto &w : workers_)
                    if (w->ctx) dcn_ctx_destroy(w->ctx);
                throw Error(rc, msg);
            }
        }
        for (auto &w : workers_) {
            Worker *wp = w.get();
            wp->thread = std::thread([this, wp] { work(*wp); });
        }
    }
    MultiGpuFilter(const MultiGpuFilter &) = delete;
    MultiGpuFilter &operator=(const MultiGpuFilter &) = delete;
    ~MultiGpuFilter() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &w : workers_) {
            if (w->thread.joinable()) w->thread.join();
            if (w->ctx) dcn_ctx_destroy(w->ctx);
        }
    }

    std::size_t workers() const { return workers_.size(); }

    // hands the batch to worker seq % N and returns its sequence number; blocks while that worker's queue is full
    uint64_t submit(const Job &job) {
        std::unique_lock<std::mutex> l(m_);
        const uint64_t seq = next_seq_++;
        Worker &w = *workers_[seq % workers_.size()];
        cv_.wait(l, [&] { return w.queue.size() < queue_depth_; });
        w.queue.emplace_back(seq, job);
        cv_.notify_all();
        return seq;
    }
    // blocks until batch `seq` is done and its outputs are in the caller's arrays; throws what the worker caught
    void wait(uint64_t seq) {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return done_.count(seq) != 0; });
        std::pair<int, std::string> r = std::move(done_[seq]);
        done_.erase(seq);
        if (r.first != DCN_OK) throw Error(r.first, r.second);
    }
    // the six counters summed over every context (ProcessingStats merge, local_filter.rs:388-396); call once
    // everything submitted has been waited for
    ProcessingStats stats() {
        std::vector<dcn_ctx *> ctxs;
        for (auto &w : workers_) ctxs.push_back(w->ctx);
        std::array<uint64_t, DCN_N_STATS> c{};
        check(dcn_stats_allreduce(ctxs.data(), static_cast<int>(ctxs.size()), c.data()));
        ProcessingStats s;
        s.total_seqs = c[DCN_STAT_TOTAL_SEQS];
        s.filtered_seqs = c[DCN_STAT_FILTERED_SEQS];
        s.total_bp = c[DCN_STAT_TOTAL_BP];
        s.output_bp = c[DCN_STAT_OUTPUT_BP];
        s.filtered_bp = c[DCN_STAT_FILTERED_BP];
        s.output_seq_counter = c[DCN_STAT_OUTPUT_SEQ_COUNTER];
        return s;
    }

  private:
    struct Worker {
        dcn_ctx *ctx = nullptr;
        std::thread thread;
        std::deque<std::pair<uint64_t, Job>> queue;
    };
    void finish(uint64_t seq, int rc) {
        std::string msg = rc == DCN_OK ? std::string() : std::string("deacon_hip error ") + std::to_string(rc) + ": " + dcn_last_error();
        std::lock_guard<std::mutex> l(m_);
        done_[seq] = {rc, std::move(msg)};
        cv_.notify_all();
    }
    void work(Worker &w) {
        dcn_params p;
        p.abs_threshold = config_.abs_threshold;
        p.rel_threshold = config_.rel_threshold;
        p.prefix_length = config_.prefix_length;
        p.deplete = config_.deplete ? 1u : 0u;
        p.reserved = 0;
        std::deque<std::pair<uint64_t, uint64_t>> flying;  // (sequence number, ticket), oldest first
        for (;;) {
            std::pair<uint64_t, Job> job;
            bool have = false;
            {
                std::unique_lock<std::mutex> l(m_);
                // with a batch in flight do not sleep on an empty queue: go and collect it
                cv_.wait(l, [&] { return stop_ || !w.queue.empty() || !flying.empty(); });
                if (!w.queue.empty() && flying.size() < 2) {
                    job = w.queue.front();
                    w.queue.pop_front();
                    have = true;
                    cv_.notify_all();
                } else if (flying.empty() && stop_) {
                    return;
                }
            }
            if (have) {
                uint64_t ticket = 0;
                const Job &j = job.second;
                int rc = j.packed ? dcn_filter_batch_packed_submit(w.ctx, j.packed, j.invmask, j.offsets, j.unit_id, j.n_reads, &p,
                                                                   j.keep, j.hits, j.total, &ticket)
                                  : dcn_filter_batch_submit(w.ctx, j.bases, j.offsets, j.unit_id, j.n_reads, &p, j.keep, j.hits,
                                                            j.total, &ticket);
                if (rc != DCN_OK) finish(job.first, rc);
                else flying.emplace_back(job.first, ticket);
                if (flying.size() < 2) continue;  // look for a second batch before blocking on the first
            }
            if (!flying.empty()) {
                auto f = flying.front();
                flying.pop_front();
                finish(f.first, dcn_filter_batch_wait(w.ctx, f.second));
            }
        }
    }

    FilterConfig config_;
    std::size_t queue_depth_;
    std::vector<Index> replicas_;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    std::map<uint64_t, std::pair<int, std::string>> done_;
    uint64_t next_seq_ = 0;
    bool stop_ = false;
};

// filter_common.rs:211-310 for one read: (minimizer hashes, positions) after the ACGT filter
inline std::pair<std::vector<uint64_t>, std::vector<uint32_t>> get_minimizer_hashes_and_positions(
    FilterProcessor &proc, std::string_view seq, std::size_t prefix_length = 0) {
    uint64_t offsets[2] = {0, seq.size()};
    uint64_t out_off[2] = {0, 0};
    std::vector<uint64_t> h(seq.size() + 1);
    std::vector<uint32_t> p(seq.size() + 1);
    check(dcn_minimizer_hashes_batch(proc.raw(), reinterpret_cast<const uint8_t *>(seq.data()), offsets, 1,
                                     prefix_length, out_off, h.data(), p.data(), h.size()));
    h.resize(out_off[1]);
    p.resize(out_off[1]);
    return {std::move(h), std::move(p)};
}

// The same for a read of ANY length -- longer than the context's largest batch included (the reference has no such limit:
// src/local_filter.rs:346-374 takes whatever paraseq hands it, whole chromosomes too).  A window's choice depends only on
// its own l = k + w - 1 bases, so the read is cut into pieces of `piece_windows` windows, each piece one call of
// dcn_minimizer_hashes_batch; only the rule that drops CONSECUTIVE duplicate positions crosses a seam.  A later piece
// therefore starts one window early (the last window of its predecessor): that "carry" window's choice c is the last entry
// of the predecessor's list and the first of this piece's, and the read's list is the lists joined with that first entry
// removed.  After the ACGT filter (src/filter_common.rs:275-286) c is present in both lists or in neither; which, the GPU
// says itself: the carry window's l bases travel as a read of their own in the same call and yield one entry or none.
// Positions are read-relative, as u64 (a chromosome's do not fit the u32 of the per-read call).  Nothing is computed here:
// every hash and position comes out of the library's kernels.
inline void append_minimizer_hashes_any_length(FilterProcessor &proc, const uint8_t *seq, uint64_t len, unsigned k, unsigned w,
                                               uint64_t prefix_length, uint64_t piece_windows, std::vector<uint64_t> &hashes,
                                               std::vector<uint64_t> *positions = nullptr) {
    if (len < k) return;                                        // filter_common.rs:217-219, on the full read
    uint64_t n = (prefix_length > 0 && len > prefix_length) ? prefix_length : len;  // :221-226
    const uint64_t l = static_cast<uint64_t>(k) + w - 1;
    if (n < l) return;                                          // no window at all
    const uint64_t windows = n - l + 1;
    if (piece_windows < 2) piece_windows = 2;
    std::vector<uint8_t> buf;
    std::vector<uint64_t> h;
    std::vector<uint32_t> p;
    for (uint64_t s = 0; s < windows; s += piece_windows) {
        const uint64_t e = std::min(windows, s + piece_windows);
        const uint64_t first = s ? s - 1 : 0;                    // first window of the piece: the carry window for s > 0
        const uint64_t piece_len = (e - first) + l - 1;
        uint64_t offsets[3] = {0, s ? l : 0, (s ? l : 0) + piece_len}, out_off[3] = {0, 0, 0};
        buf.resize(offsets[2]);
        if (s) std::memcpy(buf.data(), seq + first, l);          // read 0: the carry window alone (empty for the first piece)
        std::memcpy(buf.data() + offsets[1], seq + first, piece_len);
        h.resize(piece_len + 2);
        p.resize(piece_len + 2);
        check(dcn_minimizer_hashes_batch(proc.raw(), buf.data(), offsets, 2, 0, out_off, h.data(), p.data(), h.size()));
        uint64_t from = out_off[1];
        if (s && out_off[1] == 1) {                              // the carry window's choice passed the ACGT filter:
            if (out_off[2] == out_off[1] || h[from] != h[0] || p[from] != p[0])
                throw std::runtime_error("a piece does not start with its carry window's minimizer");
            ++from;                                              // ... it is the predecessor's last entry, not a new one
        } else if (s && out_off[1] != 0) {
            throw std::runtime_error("one window gave more than one minimizer");
        }
        hashes.insert(hashes.end(), h.begin() + from, h.begin() + out_off[2]);
        if (positions)
            for (uint64_t j = from; j < out_off[2]; ++j) positions->push_back(first + p[j]);
    }
}

namespace detail {
inline std::vector<Decision> should_keep(FilterProcessor &proc, const std::vector<std::vector<uint64_t>> &input,
                                         std::size_t abs_threshold, double rel_threshold, bool deplete) {
    std::vector<uint64_t> flat, off(1, 0);
    for (const auto &v : input) {
        flat.insert(flat.end(), v.begin(), v.end());
        off.push_back(flat.size());
    }
    std::vector<uint8_t> keep(input.size());
    std::vector<uint32_t> hits(input.size()), total(input.size());
    dcn_params p = proc.params();
    p.abs_threshold = abs_threshold;
    p.rel_threshold = rel_threshold;
    p.deplete = deplete ? 1u : 0u;
    check(dcn_should_keep_hashes(proc.raw(), flat.data(), off.data(), static_cast<uint32_t>(input.size()), &p,
                                 keep.data(), hits.data(), total.data()));
    std::vector<Decision> out(input.size());
    for (std::size_t u = 0; u < input.size(); ++u) out[u] = Decision(keep[u] != 0, hits[u], total[u]);
    return out;
}
}  // namespace detail

// remote_filter.rs:230-264: one Vec<u64> of minimizer hashes per read
inline std::vector<Decision> unpaired_should_keep(FilterProcessor &proc,
                                                  const std::vector<std::vector<uint64_t>> &input_minimizers,
                                                  std::size_t abs_threshold, double rel_threshold, bool deplete) {
    return detail::should_keep(proc, input_minimizers, abs_threshold, rel_threshold, deplete);
}

// remote_filter.rs:266-301: one Vec<u64> per pair (both mates' hashes concatenated)
inline std::vector<Decision> paired_should_keep(FilterProcessor &proc,
                                                const std::vector<std::vector<uint64_t>> &input_minimizers,
                                                std::size_t abs_threshold, double rel_threshold, bool deplete) {
    return detail::should_keep(proc, input_minimizers, abs_threshold, rel_threshold, deplete);
}

}  // namespace deacon
#endif  // DEACON_HIP_HPP
