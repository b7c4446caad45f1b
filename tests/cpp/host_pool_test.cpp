// The library's host pool (deacon-server_amd/csrc/dcn_host_pool.h) under ThreadSanitizer / AddressSanitizer on the CPU:
// tests/test_host_pack.py::test_host_pool_under_the_sanitizers.  Several caller threads (the contexts of one process) run
// jobs at the same time, large and small, with pauses that send the workers to sleep; every slice of every job must run
// exactly once, in the job it belongs to, and the pool must grow while jobs are in flight.  Exit code 0 = all counts right.
#include "dcn_host_pool.h"
#include <cstdio>
#include <random>

int main() {
    using dcn_host::HostPool;
    HostPool &pool = HostPool::get();
    const int W = pool.width();
    std::atomic<long> bad{0};
    std::atomic<bool> grow{false};
    auto caller = [&](int id, int jobs, bool crowded) {
        std::mt19937 rng(id * 7919 + 1);
        std::vector<int> hits((size_t)W);
        std::vector<unsigned char> a(1 << 20), b(1 << 20);
        for (int j = 0; j < jobs; ++j) {
            std::fill(hits.begin(), hits.end(), 0);
            long sum = 0;
            std::atomic<long> asum{0};
            const int tag = id * 100000 + j;
            std::atomic<int> seen_nt{0};
            pool.run([&](int i, int nt) {
                if (nt != W && nt != 1) bad.fetch_add(1);
                seen_nt.store(nt); // (W, or 1 when the job ran inline: a small job, or more callers than the pool has slots)
                hits[(size_t)i] += 1; // (each slice is one thread's: no atomics needed if the pool is right -- TSan checks)
                asum.fetch_add(tag + i);
            }, j % 11 == 0);
            const int nt = seen_nt.load();
            if (j % 11 == 0 && nt != 1) bad.fetch_add(1);
            if (j % 11 != 0 && nt != W && !crowded) bad.fetch_add(1);
            for (int i = 0; i < nt; ++i) {
                if (hits[(size_t)i] != 1) bad.fetch_add(1);
                sum += tag + i;
            }
            for (int i = nt; i < W; ++i)
                if (hits[(size_t)i] != 0) bad.fetch_add(1);
            if (sum != asum.load()) bad.fetch_add(1);
            if (j % 7 == 0) { // a threaded copy, small and large
                const size_t n = (j % 14 == 0) ? a.size() : 4096 + rng() % 5000;
                for (size_t q = 0; q < n; ++q) a[q] = (unsigned char)(q * 31 + tag);
                pool.copy(b.data(), a.data(), n);
                if (memcmp(a.data(), b.data(), n)) bad.fetch_add(1);
            }
            if (rng() % 50 == 0) std::this_thread::sleep_for(std::chrono::microseconds(300 + rng() % 700)); // workers go to sleep
            if (id == 0 && j == jobs / 2 && !grow.exchange(true)) pool.ensure_devices(2);
        }
    };
    std::vector<std::thread> ts;
    const int callers = 6, jobs = 1500;
    for (int t = 0; t < callers; ++t) ts.emplace_back(caller, t, jobs, false);
    for (auto &t : ts) t.join();
    // more callers than slots (kSlots = 16): the surplus runs its job inline
    ts.clear();
    for (int t = 0; t < 24; ++t) ts.emplace_back(caller, 100 + t, 60, true);
    for (auto &t : ts) t.join();
    printf("width %d threads %d bad %ld\n", W, pool.threads(), bad.load());
    return bad.load() != 0;
}
