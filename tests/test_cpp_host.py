"""The C++ host layer (include/deacon_hip.hpp) compiled with g++ against libdeacon_hip.so and driven like the
reference's Rust callers; results compared with the oracle (GPU) / loud failure checked (CPU)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import mutate, random_reads, revcomp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory, dcn):
    out = tmp_path_factory.mktemp("cpp") / "host_layer_test"
    libdir = os.path.dirname(dcn._native.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_layer_test.cpp"), "-o", str(out),
                           "-L", libdir, "-ldeacon_hip", "-lpthread", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
    return str(out)


def write_case(path, k, w, abs_t, rel_t, prefix, deplete, paired, keys, reads):
    with open(path, "w") as f:
        f.write(f"{k} {w} {abs_t} {rel_t} {prefix} {int(deplete)} {int(paired)} {len(keys)}\n")
        f.write(" ".join(f"{int(x):x}" for x in keys) + "\n")
        f.write(f"{len(reads)}\n")
        for r in reads:
            f.write((r.decode() if r else "-") + "\n")


def test_cpp_layer_compiles_and_fails_loudly_without_gpu(driver, dcn, tmp_path):
    import ctypes
    n = ctypes.c_int()
    if dcn._native.lib().dcn_device_count(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    case = tmp_path / "case.txt"
    write_case(case, 31, 15, 2, 0.01, 0, False, False, [1, 2, 3], [b"ACGT" * 20])
    p = subprocess.run([driver, str(case)], capture_output=True, text=True)
    assert p.returncode == 3 and "deacon::Error" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("paired,deplete", [(False, False), (True, True)])
def test_cpp_layer_matches_oracle(driver, oracle, tmp_path, paired, deplete):
    rng = np.random.default_rng(77)
    genome = random_reads(rng, 1, 40_000, 40_000)[0]
    oidx = oracle.Index.build([genome])
    reads = []
    for i in range(401):
        ln = int(rng.integers(20, 260))
        if i % 2:
            s = int(rng.integers(0, len(genome) - ln))
            r = mutate(rng, genome[s:s + ln], 0.01)
            reads.append(revcomp(r) if i % 4 == 1 else r)
        else:
            reads.append(random_reads(rng, 1, ln, ln, p_n=0.01)[0])
    reads[7] = b""
    case = tmp_path / "case.txt"
    write_case(case, 31, 15, 2, 0.01, 0, deplete, paired, oidx.keys(), reads)
    p = subprocess.run([driver, str(case), "0,0,0"], capture_output=True, text=True, check=True)
    lines = p.stdout.strip().split("\n")
    assert lines[0] == f"header 31 15 {len(oidx)}"
    uid = (np.arange(len(reads)) // 2).astype(np.uint32) if paired else None
    b, o = oracle.concat_reads(reads)
    keep, hits, total = oracle.filter_batch(oidx, b, o, uid, deplete=deplete)
    units = [l.split()[1:] for l in lines if l.startswith("unit ")]
    assert [int(u[0]) for u in units] == keep.astype(int).tolist()
    assert [int(u[1]) for u in units] == hits.tolist()
    assert [int(u[2]) for u in units] == total.tolist()
    lens = np.array([len(r) for r in reads])
    ulen = np.bincount(uid, weights=lens).astype(np.int64) if paired else lens
    ucnt = np.bincount(uid) if paired else np.ones(len(reads), np.int64)
    assert [int(x) for x in next(l for l in lines if l.startswith("keeponly")).split()[1:]] == [int(x) for x in keep]
    # process_record / on_batch_complete / on_thread_complete: record sets gather, 256 reads per call, input order kept
    assert next(l for l in lines if l.startswith("gathered")).split()[1:] == [str(int(x)) for x in keep]
    st = [int(x) for x in next(l for l in lines if l.startswith("stats ")).split()[1:]]
    assert st == [len(reads), int(ucnt[~keep].sum()), int(lens.sum()), int(ulen[keep].sum()),
                  int(ulen[~keep].sum()), int(ucnt[keep].sum())]
    # deacon::MultiGpuFilter over three contexts on GPU 0 (index cloned device to device, batches of 64 units dealt
    # round-robin, merged by sequence number): identical to the single-context run, counters summed
    multi = next(l for l in lines if l.startswith("multi ")).split()[1:]
    assert multi[0] == "3"
    # batches hold whole units and 128 reads (64 units when paired), so pairing is the same as in the one-batch run
    assert [tuple(int(x) for x in m.split(":")) for m in multi[1:]] == \
        list(zip(keep.astype(int).tolist(), hits.tolist(), total.tolist()))
    mst = [int(x) for x in next(l for l in lines if l.startswith("multistats ")).split()[1:]]
    assert mst == st
    single = [int(x) for x in next(l for l in lines if l.startswith("single ")).split()[1:]]
    assert single == [int(keep[0]), int(hits[0]), int(total[0])]
    wh, wp = oracle.minimizer_hashes_and_positions(reads[0], 31, 15)
    got = next(l for l in lines if l.startswith("minimizers")).split()[1:]
    assert got == [f"{int(h):x}:{int(q)}" for h, q in zip(wh, wp)]
    hk = oracle.should_keep_hashes(oidx, wh, np.array([0, len(wh)], np.uint64), 2, 0.01, deplete)
    assert [int(x) for x in next(l for l in lines if l.startswith("hashes ")).split()[1:]] == \
        [int(hk[0][0]), int(hk[1][0]), int(hk[2][0])]
