"""CPU checks of the measurement code itself (bench.py / bench_cli.py): the synthetic streams have the shape the configs
name, the timing lines of the tool are parsed as printed, and a `roofline.traffic` figure is only taken from a committed
PMC file when workload, batch size and table all match."""
import json
import os

import numpy as np
import pytest
import torch

import bench
import bench_cli


def test_mixed_stream_is_half_long_half_short_and_interleaved():
    dev = torch.device("cpu")
    g = bench.make_host_genome(1_000_000, 3, dev)
    b, o = bench.make_mixed_reads(g, 2_000_000, 9, dev)
    o = o.numpy()
    ln = np.diff(o)
    assert int(o[-1]) == b.numel() and o[0] == 0
    long_bases, short_bases = int(ln[ln != 150].sum()), int(ln[ln == 150].sum())
    assert abs(long_bases - short_bases) < 0.02 * b.numel()             # half and half by bases
    assert ln[0] != 150 and (ln[1:1 + int(round(ln[0] / 150))] == 150).all()  # every long read is followed by its share of short ones
    assert set(np.unique(b.numpy()).tolist()) <= set(b"ACGTN")
    lb, lo = bench.make_long_reads(g, 1_000_000, 9, dev)                  # the long reads are configs[2]'s generator, same seed
    assert bytes(b[o[0]:o[1]].numpy()) == bytes(lb[lo[0]:lo[1]].numpy())


def test_tool_timing_lines_are_parsed():
    err = ("timing: process CPU 4.095 s user + 1.544 s system\n"
           "timing: wall 0.782 s; busy seconds: parse 3.742 (all workers), GPU stage 0.043 (main thread waited 0.012 for it), "
           "format 1.295 (all workers), write 0.000; main thread blocked pushing to format 0.001\n"
           "timing: milestones (s): index loaded 0.298, contexts ready 0.312, all input parsed+queued 0.762, GPU stage drained 0.764, all written 0.765\n")
    t = bench_cli.parse_timing(err)
    assert t["run_wall_s"] == 0.782 and t["busy_core_s"] == {"parse": 3.742, "gpu_stage": 0.043, "format": 1.295, "write": 0.0}
    assert t["milestones_s"]["index_loaded"] == 0.298 and t["milestones_s"]["all_written"] == 0.765
    assert t["milestones_s"]["contexts_ready"] == 0.312 and t["process_cpu_s"] == {"user": 4.095, "system": 1.544}
    assert bench_cli.parse_timing("Retained 1/2 sequences") == {}


def test_committed_traffic_needs_a_matching_workload(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    json.dump({"workload": {"workload": "long", "reads_per_batch": 150_000, "bases_per_batch": 1_500_000_000, "index_keys": 409_913_780,
                            "host_genome_bases": 64_000_000},
               "scan_kernel": {"hbm_bytes_per_launch": 1.3e10}}, open(prof / "r03_traffic_long.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rf = {"avg_launch_ms": 5.0, "traffic": None}
    assert bench.committed_traffic(rf, "long", 1_499_000_000, 409_913_780, 64_000_000) == 1.3e10     # within 1 % of the batch size
    assert rf["traffic_source"] == "profiles/r03_traffic_long.json" and abs(rf["traffic_rate_GBps"] - 2600.0) < 1e-6
    for args in (("long", 1_200_000_000, 409_913_780, 64_000_000), ("long", 1_500_000_000, 950_000_000, 64_000_000),
                 ("mixed", 1_500_000_000, 409_913_780, 64_000_000), ("long", 1_500_000_000, 409_913_780, 1_000_000_000)):
        rf = {"avg_launch_ms": 5.0, "traffic": None}
        assert bench.committed_traffic(rf, *args) is None and rf["traffic"] is None


def test_mix64_keys_have_decidable_membership():
    i = np.arange(1, 1000, dtype=np.uint64)
    from conftest import mix64
    h = mix64(i)
    assert bench.unmix64(h).tolist() == i.tolist()                       # bench.py's inverse of the key generator
    assert len(set(h.tolist())) == len(i)
