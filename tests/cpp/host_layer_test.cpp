// Drives the C++ host layer (include/deacon_hip.hpp) the way the reference's Rust callers drive their filter:
// build/load an index, create a FilterProcessor, call should_keep_sequence / should_keep_pair / the batch seam.
// usage: host_layer_test <case file> ; prints one line per result for tests/test_cpp_host.py to compare with the
// oracle.  Exit code 3 = the library reported an error (printed on stderr), e.g. no GPU.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "deacon_hip.hpp"

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <case file>\n", argv[0]);
        return 2;
    }
    std::ifstream in(argv[1]);
    unsigned k, w, deplete, paired;
    std::size_t abs_t, prefix;
    double rel_t;
    std::size_t n_keys, n_reads;
    in >> k >> w >> abs_t >> rel_t >> prefix >> deplete >> paired >> n_keys;
    std::vector<uint64_t> keys(n_keys);
    for (auto &x : keys) in >> std::hex >> x;
    in >> std::dec >> n_reads;
    std::vector<std::string> reads(n_reads);
    for (auto &r : reads) {
        in >> r;
        if (r == "-") r.clear();  // "-" encodes an empty read
    }
    try {
        deacon::Index index = deacon::Index::from_hashes(keys, (uint8_t)k, (uint8_t)w);
        auto hd = index.header();
        std::printf("header %u %u %llu\n", hd.kmer_length, hd.window_size, (unsigned long long)index.len());
        deacon::FilterConfig cfg;
        cfg.abs_threshold = abs_t;
        cfg.rel_threshold = rel_t;
        cfg.prefix_length = prefix;
        cfg.deplete = deplete != 0;
        cfg.max_batch_bases = 1 << 22;
        cfg.max_batch_reads = 1 << 14;
        deacon::FilterProcessor proc(index, cfg);
        // batch seam
        std::vector<std::string_view> views(reads.begin(), reads.end());
        auto res = proc.filter_batch(views, paired != 0);
        for (auto &[keep, hits, total] : res) std::printf("unit %d %zu %zu\n", keep ? 1 : 0, hits, total);
        auto st = proc.stats();
        {   // decisions only: same answers; its counters are not part of the "stats" line printed below
            auto only = proc.keep_batch(views, paired != 0);
            std::printf("keeponly");
            for (bool b : only) std::printf(" %d", b ? 1 : 0);
            std::printf("\n");
        }
        std::printf("stats %llu %llu %llu %llu %llu %llu\n", (unsigned long long)st.total_seqs,
                    (unsigned long long)st.filtered_seqs, (unsigned long long)st.total_bp,
                    (unsigned long long)st.output_bp, (unsigned long long)st.filtered_bp,
                    (unsigned long long)st.output_seq_counter);
        // per-read seam on the first unit, and the minimizers of the first read
        if (!reads.empty()) {
            deacon::Decision d = (paired && reads.size() > 1) ? proc.should_keep_pair(reads[0], reads[1])
                                                              : proc.should_keep_sequence(reads[0]);
            std::printf("single %d %zu %zu\n", std::get<0>(d) ? 1 : 0, std::get<1>(d), std::get<2>(d));
            auto [h, p] = deacon::get_minimizer_hashes_and_positions(proc, reads[0], prefix);
            std::printf("minimizers");
            for (std::size_t i = 0; i < h.size(); ++i) std::printf(" %llx:%u", (unsigned long long)h[i], p[i]);
            std::printf("\n");
            // server seam: the first read's hashes as one unit
            auto sk = deacon::unpaired_should_keep(proc, {h}, abs_t, rel_t, deplete != 0);
            std::printf("hashes %d %zu %zu\n", std::get<0>(sk[0]) ? 1 : 0, std::get<1>(sk[0]), std::get<2>(sk[0]));
        }
    } catch (const deacon::Error &e) {
        std::fprintf(stderr, "deacon::Error %d: %s\n", e.code(), e.what());
        return 3;
    }
    return 0;
}
