// What the batch seam delivers at the batch sizes a host really passes: T caller threads, one dcn_ctx each (the shape of the
// reference's per-worker FilterProcessor, src/local_filter.rs:153-177, 696-709), blocking dcn_filter_batch on pageable ASCII with
// R reads of 150 bp per call -- from paraseq's default record set (1,024 records) to the bench's 10 M.
//   g++ -O2 -std=c++17 -pthread -I include profiles/small_calls.cpp -L deacon-server_amd/lib -ldeacon_hip -Wl,-rpath,$PWD/deacon-server_amd/lib -o /tmp/small_calls
//   /tmp/small_calls [index_keys=100000000] [seconds_per_point=1.0] [decisions_only=1] [only_reads_per_call] [only_threads]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "deacon_hip.h"

static void check(int rc, const char *what) {
    if (rc != DCN_OK) {
        std::fprintf(stderr, "%s: %s\n", what, dcn_last_error());
        std::exit(1);
    }
}

int main(int argc, char **argv) {
    const uint64_t n_keys = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100000000ull;
    const double secs = argc > 2 ? std::atof(argv[2]) : 1.0;
    const bool decisions_only = argc > 3 ? std::atoi(argv[3]) != 0 : true;
    const size_t only_R = argc > 4 ? (size_t)std::atol(argv[4]) : 0;
    const int only_T = argc > 5 ? std::atoi(argv[5]) : 0;
    std::vector<uint64_t> keys(n_keys);
    {
        uint64_t x = 88172645463325252ull;
        for (auto &k : keys) {
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            k = x;
        }
    }
    dcn_index *index = nullptr;
    check(dcn_index_from_keys(keys.data(), n_keys, 31, 15, 0, &index), "index");
    keys.clear();
    keys.shrink_to_fit();
    const uint32_t L = 150;
    const size_t max_reads = 1u << 20;
    std::vector<uint8_t> bases(max_reads * L);
    {
        std::mt19937_64 rng(1);
        for (size_t i = 0; i < bases.size(); i += 32) {
            uint64_t r = rng();
            for (size_t j = 0; j < 32 && i + j < bases.size(); ++j, r >>= 2) bases[i + j] = "ACGT"[r & 3];
        }
    }
    std::vector<uint64_t> offsets(max_reads + 1);
    for (size_t i = 0; i <= max_reads; ++i) offsets[i] = i * L;
    dcn_params params;
    std::memset(&params, 0, sizeof params);
    params.abs_threshold = 2;
    params.rel_threshold = 0.01;
    std::printf("# index %llu keys; blocking dcn_filter_batch, pageable ASCII, %s; Gbp/s summed over the caller threads (calls per second per thread)\n",
                (unsigned long long)n_keys, decisions_only ? "decisions only (hits/total NULL)" : "counting");
    std::printf("%-12s", "reads/call");
    const int Ts[] = {1, 2, 4, 8, 16};
    for (int T : Ts) std::printf("  T=%-18d", T);
    std::printf("\n");
    for (size_t R : {(size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)262144, (size_t)1048576}) {
        if (only_R && R != only_R) continue;
        std::printf("%-12zu", R);
        for (int T : Ts) {
            if (only_T && T != only_T) continue;
            std::vector<dcn_ctx *> ctx(T, nullptr);
            for (int t = 0; t < T; ++t) check(dcn_ctx_create(index, R * L + 1024, (uint32_t)R, &ctx[t]), "ctx");
            std::atomic<bool> go{false}, stop{false};
            std::vector<uint64_t> calls(T, 0);
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    std::vector<uint8_t> keep(R);
                    std::vector<uint32_t> hits(R), total(R);
                    // every thread its own stretch of the read buffer (as a worker's own record set would be)
                    const size_t first = ((size_t)t * 7919 * 64) % (max_reads - R + 1);
                    std::vector<uint64_t> off(R + 1);
                    for (size_t i = 0; i <= R; ++i) off[i] = i * L;
                    const uint8_t *b = bases.data() + first * L;
                    for (int warm = 0; warm < 3; ++warm)
                        check(dcn_filter_batch(ctx[t], b, off.data(), nullptr, (uint32_t)R, &params, keep.data(), decisions_only ? nullptr : hits.data(),
                                               decisions_only ? nullptr : total.data()),
                              "filter");
                    while (!go.load()) std::this_thread::yield();
                    while (!stop.load()) {
                        check(dcn_filter_batch(ctx[t], b, off.data(), nullptr, (uint32_t)R, &params, keep.data(), decisions_only ? nullptr : hits.data(),
                                               decisions_only ? nullptr : total.data()),
                              "filter");
                        ++calls[t];
                    }
                });
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
            const auto t0 = std::chrono::steady_clock::now();
            go = true;
            std::this_thread::sleep_for(std::chrono::duration<double>(secs));
            stop = true;
            for (auto &x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            uint64_t c = 0;
            for (auto v : calls) c += v;
            std::printf("  %7.2f (%8.0f/s)", (double)c * R * L / dt / 1e9, (double)c / dt / T);
            std::fflush(stdout);
            for (auto *x : ctx) dcn_ctx_destroy(x);
        }
        std::printf("\n");
    }
    dcn_index_destroy(index);
    return 0;
}
