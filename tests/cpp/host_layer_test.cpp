// Drives the C++ host layer (include/deacon_hip.hpp) the way the reference's Rust callers drive their filter:
// build/load an index, create a FilterProcessor, call should_keep_sequence / should_keep_pair / the batch seam.
// usage: host_layer_test <case file> [devices] ; prints one line per result for tests/test_cpp_host.py to compare with the
// oracle.  Exit code 3 = the library reported an error (printed on stderr), e.g. no GPU.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
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
        {   // the paraseq-shaped seam: record sets of 100 gather, 256 reads (or pairs' mates) decided per call, the rest at the end
            deacon::FilterProcessor gp(index, cfg);
            gp.set_flush_reads(256);
            std::string line = "gathered";
            std::size_t seen = 0, calls = 0;
            auto sink = [&](const deacon::FilterProcessor::Record &r, bool keep) {
                if (r.id != "r" + std::to_string(seen) || r.seq != reads[seen]) line += " OUT-OF-ORDER";
                if (!paired || seen % 2 == 0 || seen + 1 == reads.size()) line += keep ? " 1" : " 0";
                ++seen;
            };
            const std::size_t whole = paired ? reads.size() / 2 * 2 : reads.size();
            for (std::size_t i = 0; i < whole; i += paired ? 2 : 1) {
                if (paired) gp.process_record_pair("r" + std::to_string(i), reads[i], "", "r" + std::to_string(i + 1), reads[i + 1], "");
                else gp.process_record("r" + std::to_string(i), reads[i]);
                if ((i / (paired ? 2 : 1)) % 100 == 99) gp.on_batch_complete(sink), ++calls;
            }
            gp.on_thread_complete(sink);
            if (paired && whole < reads.size()) {  // (a trailing single read is its own unit in filter_batch's reading of the list)
                gp.process_record("r" + std::to_string(whole), reads[whole]);
                gp.on_thread_complete(sink);
            }
            std::printf("%s\n", line.c_str());
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
        // in-process multi-GPU driver: argv[2] = device list ("0,0" = two contexts on GPU 0); the reads are cut into
        // batches of whole units, dealt round-robin, merged in sequence order
        if (argc > 2) {
            std::vector<int> devices;
            std::stringstream ss(argv[2]);
            for (std::string tok; std::getline(ss, tok, ',');) devices.push_back(std::stoi(tok));
            deacon::MultiGpuFilter multi(index, devices, cfg);
            const std::size_t per = paired ? 2 : 1, batch_reads = 64 * per;
            struct Owned {
                std::vector<uint8_t> bases, keep;
                std::vector<uint64_t> offsets;
                std::vector<uint32_t> unit_id, hits, total;
            };
            std::vector<std::unique_ptr<Owned>> owned;
            std::vector<uint64_t> seqs;
            for (std::size_t r0 = 0; r0 < reads.size(); r0 += batch_reads) {
                std::size_t r1 = std::min(reads.size(), r0 + batch_reads);
                owned.emplace_back(new Owned());
                Owned &o = *owned.back();
                o.offsets.push_back(0);
                for (std::size_t r = r0; r < r1; ++r) {
                    o.bases.insert(o.bases.end(), reads[r].begin(), reads[r].end());
                    o.offsets.push_back(o.bases.size());
                    if (paired) o.unit_id.push_back((uint32_t)((r - r0) / 2));
                }
                std::size_t nu = paired ? (r1 - r0 + 1) / 2 : r1 - r0;
                o.keep.assign(nu, 0);
                o.hits.assign(nu, 0);
                o.total.assign(nu, 0);
                deacon::MultiGpuFilter::Job job;
                job.bases = o.bases.data();
                job.offsets = o.offsets.data();
                job.unit_id = paired ? o.unit_id.data() : nullptr;
                job.n_reads = (uint32_t)(r1 - r0);
                job.keep = o.keep.data();
                job.hits = o.hits.data();
                job.total = o.total.data();
                seqs.push_back(multi.submit(job));
            }
            std::printf("multi %zu", multi.workers());
            for (std::size_t i = 0; i < seqs.size(); ++i) {
                multi.wait(seqs[i]);  // ordered merge: by batch sequence number
                for (std::size_t u = 0; u < owned[i]->keep.size(); ++u)
                    std::printf(" %d:%u:%u", owned[i]->keep[u], owned[i]->hits[u], owned[i]->total[u]);
            }
            std::printf("\n");
            auto ms = multi.stats();
            std::printf("multistats %llu %llu %llu %llu %llu %llu\n", (unsigned long long)ms.total_seqs,
                        (unsigned long long)ms.filtered_seqs, (unsigned long long)ms.total_bp,
                        (unsigned long long)ms.output_bp, (unsigned long long)ms.filtered_bp,
                        (unsigned long long)ms.output_seq_counter);
        }
    } catch (const deacon::Error &e) {
        std::fprintf(stderr, "deacon::Error %d: %s\n", e.code(), e.what());
        return 3;
    }
    return 0;
}
