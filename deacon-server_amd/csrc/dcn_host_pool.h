// The library's host-side thread pool (plain C++: no HIP in here, so tests/cpp/host_pool_test.cpp can put it under
// ThreadSanitizer on the CPU).  Included by api.hip only.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

namespace dcn_host {

// A few host threads that split one large job (a copy or a pack) into slices: a single core moves ~10 GB/s into
// the pinned staging buffer, which is less than the PCIe link takes out of it.  Process-wide, created on first
// use; DCN_HOST_THREADS sets the width (default: the usable CPUs, at most 16; 1 = run inline).
//
// Several jobs at a time (round 4).  A job is `width()` slices; the caller's thread and the pool's workers CLAIM slices
// one by one, from whichever of the jobs in flight has any left, so the contexts of one process -- deacon::MultiGpuFilter,
// `deacon-hip filter --gpus N`, a Rust host with one worker thread per GPU -- pack their pageable input side by side
// instead of taking turns behind one GPU's packing (the one-job-at-a-time rule of rounds 2-3: DESIGN.md section 5).  The
// workers are shared and the scheme is work-conserving: one context alone still gets all of them.  With contexts on more
// than one device the pool grows (ensure_devices: up to 16 threads per device in use, never beyond the CPUs the process
// may use), which is what a per-device pool would give without idling one device's threads while another's pack runs.
class HostPool {
  public:
    static HostPool &get() {
        static HostPool pool;
        return pool;
    }
    int width() const { return slices_; } // slices per job: fn(i, width())
    int threads() {
        std::lock_guard<std::mutex> g(mu_);
        return (int)workers_.size() + 1;
    }
    // fn(i, n): slice i of n, each slice run exactly once by some thread
    void run(const std::function<void(int, int)> &fn, bool small = false) {
        if (slices_ <= 1 || small) {
            fn(0, 1);
            return;
        }
        Slot *sl = nullptr;
        for (int spin = 0; !sl; ++spin) { // a free slot: more jobs in flight than slots only with > kSlots caller threads
            for (auto &cand : slots_) {
                bool expect = false;
                if (cand.busy.compare_exchange_strong(expect, true, std::memory_order_acquire)) {
                    sl = &cand;
                    break;
                }
            }
            if (!sl) {
                if (spin > 64) {
                    fn(0, 1);
                    return;
                }
                std::this_thread::yield();
            }
        }
        sl->fn.store(&fn, std::memory_order_relaxed);
        sl->n.store(slices_, std::memory_order_relaxed);
        sl->done.store(0, std::memory_order_relaxed);
        const uint64_t epoch = (sl->state.load(std::memory_order_relaxed) >> 32) + 1;
        sl->state.store(epoch << 32, std::memory_order_release); // published: slice 0 of this epoch is up for claim
        {
            std::lock_guard<std::mutex> g(mu_); // (under the lock: a worker about to sleep re-checks the generation under it)
            generation_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        while (claim_and_run(*sl)) {
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (sl->done.load(std::memory_order_acquire) != slices_) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) {
                std::unique_lock<std::mutex> g(mu_);
                done_cv_.wait(g, [&] { return sl->done.load(std::memory_order_acquire) == slices_; });
                break;
            }
            cpu_relax();
        }
        sl->busy.store(false, std::memory_order_release);
    }
    void copy(void *dst, const void *src, size_t n) {
        static const size_t par_min = getenv("DCN_COPY_PAR_MIN") ? (size_t)atoll(getenv("DCN_COPY_PAR_MIN")) : (size_t)4 << 20;
        run([&](int i, int nt) {
            size_t per = ((n / nt) + 4095) & ~(size_t)4095;
            size_t lo = std::min(n, per * i), hi = i == nt - 1 ? n : std::min(n, per * (i + 1));
            if (hi > lo) memcpy((uint8_t *)dst + lo, (const uint8_t *)src + lo, hi - lo);
        }, n < par_min);
    }
    // contexts exist on `n_devices` devices of this process: up to 16 threads per device, within the CPUs we may use
    void ensure_devices(int n_devices) {
        if (fixed_ || slices_ <= 1) return;
        const int want = std::min(cpus_, 16 * std::max(1, n_devices));
        std::lock_guard<std::mutex> g(mu_);
        while ((int)workers_.size() + 1 < want) {
            const int i = (int)workers_.size() + 1;
            workers_.emplace_back([this, i] { worker(i); });
        }
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_.store(true);
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }

  private:
    static constexpr int kSlots = 16;
    struct Slot {
        std::atomic<bool> busy{false};
        std::atomic<uint64_t> state{0}; // epoch << 32 | next slice to claim
        std::atomic<int> n{0}, done{0};
        std::atomic<const std::function<void(int, int)> *> fn{nullptr};
    };
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    // One slice of this slot's job, if it has one left.  The claim is a compare-exchange on (epoch, next): it can only
    // succeed while that epoch's job is incomplete -- its caller is still inside run(), so fn and n are the ones read.
    bool claim_and_run(Slot &sl) {
        uint64_t st = sl.state.load(std::memory_order_acquire);
        for (;;) {
            const int n = sl.n.load(std::memory_order_relaxed);
            const std::function<void(int, int)> *fn = sl.fn.load(std::memory_order_relaxed);
            const int next = (int)(uint32_t)st;
            if (!sl.busy.load(std::memory_order_relaxed) || next >= n) return false;
            if (sl.state.compare_exchange_weak(st, st + 1, std::memory_order_acq_rel, std::memory_order_acquire)) {
                (*fn)(next, n);
                if (sl.done.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
                    std::lock_guard<std::mutex> g(mu_);
                    done_cv_.notify_all();
                }
                return true;
            }
        }
    }
    HostPool() {
        // CPUs this process may really use: the affinity mask, capped by the cgroup quota (a container often sees all of
        // the host's hardware threads but is throttled to a share of them); at most 16 of those per device in use
        unsigned hw = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::min<unsigned>(hw ? hw : 1024, (unsigned)CPU_COUNT(&set));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0;
            if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                hw = std::min<unsigned>(hw ? hw : 1024, (unsigned)std::max<long long>(1, quota / period));
            fclose(f);
        }
        cpus_ = (int)std::max(1u, hw);
        // (at most 16: same-box sweeps of the host legs, round 4: 16 threads 100-105 / 103-111 Gbp/s on pageable input against 88-99 /
        // 99-102 with 12, on a share of 16 CPUs -- the pool's threads claim slices, so an oversubscribed one is late, not idle)
        int want = std::min(16, cpus_);
        if (const char *e = getenv("DCN_HOST_THREADS")) {
            want = atoi(e);
            fixed_ = true;
        }
        slices_ = std::max(1, std::min(want, 64));
        // A batch is a few dozen jobs a fraction of a millisecond apart (a chunk's pack, its offsets, their check): a
        // worker that went to sleep on the condition variable after each of them paid ~0.05 ms to wake up again, three times
        // per chunk.  It now polls the generation for a short while first (DCN_HOST_SPIN_US, default 200; 0 = sleep at once).
        if (const char *e = getenv("DCN_HOST_SPIN_US")) spin_us_ = std::max(0, atoi(e));
        for (int i = 1; i < slices_; ++i) workers_.emplace_back([this, i] { worker(i); });
    }
    void worker(int i) {
        uint64_t seen = 0;
        int streak = 0; // jobs in a row that came within the polling window of their predecessor
        for (;;) {
            // polls only while jobs keep coming that closely (a submission: a chunk's pack, its check, the next chunk's
            // pack ...); the tool's staging copies, one or two per half millisecond, would only burn the parsers' CPUs
            // (measured: 2.3 core-seconds of a 0.9 s run)
            const auto t0 = std::chrono::steady_clock::now();
            const auto window = std::chrono::microseconds(streak >= 2 ? spin_us_ : 0);
            while (generation_.load(std::memory_order_acquire) == seen && !stop_.load(std::memory_order_relaxed)) {
                if (std::chrono::steady_clock::now() - t0 >= window) {
                    std::unique_lock<std::mutex> g(mu_);
                    cv_.wait(g, [&] { return stop_.load() || generation_.load(std::memory_order_acquire) != seen; });
                    break;
                }
                cpu_relax();
            }
            if (stop_.load()) return;
            streak = std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us_) ? std::min(streak + 1, 2) : 0;
            seen = generation_.load(std::memory_order_acquire);
            // slices of every job in flight, starting at another slot than the neighbour worker does, until none is left
            for (bool any = true; any;) {
                any = false;
                for (int k = 0; k < kSlots; ++k)
                    while (claim_and_run(slots_[(size_t)((i + k) % kSlots)])) any = true;
            }
        }
    }
    int slices_ = 1, cpus_ = 1;
    bool fixed_ = false;
    int spin_us_ = 200;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    Slot slots_[kSlots];
    std::atomic<uint64_t> generation_{0};
    std::atomic<bool> stop_{false};
};

} // namespace dcn_host
