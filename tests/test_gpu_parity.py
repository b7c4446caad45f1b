"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs, bit-exact.

Everything here runs through lib/libdeacon_hip.so; the oracle is only the checker."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, mutate, random_reads, revcomp

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------------
def make_index_pair(oracle, dcn, seqs, k=31, w=15, extra_keys=()):
    """Build the same index for both sides: oracle set + device table from the oracle's key list."""
    oidx = oracle.Index.build(seqs, k=k, w=w)
    keys = oidx.keys()
    if len(extra_keys):
        keys = np.concatenate([keys, np.asarray(extra_keys, dtype=np.uint64)])
        oidx = oracle.Index(keys, k, w)
    gidx = dcn.Index.from_keys(keys, k, w)
    assert gidx.n_keys == len(oidx)
    return oidx, gidx


def check_batch(oracle, proc, oidx, reads, unit_id=None, **kw):
    b, o = oracle.concat_reads(reads)
    want = oracle.filter_batch(oidx, b, o, unit_id, abs_threshold=proc.abs_threshold,
                               rel_threshold=proc.rel_threshold, prefix_length=proc.prefix_length,
                               deplete=proc.deplete, threads=4)
    got = proc.filter_batch(b, o, unit_id)
    assert got[2].tolist() == want[2].tolist(), "total minimizers differ"
    assert got[1].tolist() == want[1].tolist(), "distinct hit counts differ"
    assert got[0].tolist() == want[0].tolist(), "keep decisions differ"
    # decisions-only mode (lanes stop probing once a read's decision is fixed): same decisions, same counters
    s0 = proc.stats()
    keep_only = proc.filter_batch(b, o, unit_id, counts=False)
    assert keep_only.tolist() == want[0].tolist(), "decisions-only mode differs"
    s1 = proc.stats()
    proc.filter_batch(b, o, unit_id)
    s2 = proc.stats()
    assert {n: s1[n] - s0[n] for n in s0} == {n: s2[n] - s1[n] for n in s0}, "counters differ between the two modes"
    return got


@pytest.fixture(scope="module")
def genome():
    return random_reads(np.random.default_rng(1), 1, 200_000, 200_000)[0]


@pytest.fixture(scope="module")
def index_pair(oracle, dcn, genome):
    return make_index_pair(oracle, dcn, [genome])


def sample_reads(rng, genome, n, min_len, max_len, host_frac=0.5, sub=0.005, p_n=0.001):
    reads = []
    for _ in range(n):
        ln = int(rng.integers(min_len, max_len + 1))
        if rng.random() < host_frac and ln < len(genome):
            s = int(rng.integers(0, len(genome) - ln))
            r = mutate(rng, genome[s:s + ln], sub)
            if rng.random() < 0.5:
                r = revcomp(r)
        else:
            r = random_reads(rng, 1, ln, ln)[0]
        if p_n > 0 and ln:
            a = np.frombuffer(r, dtype=np.uint8).copy()
            a[rng.random(ln) < p_n] = ord("N")
            r = a.tobytes()
        reads.append(r)
    return reads


# ------------------------------------------------------------------------------------------------------
# K4: the device set
# ------------------------------------------------------------------------------------------------------
def test_table_membership(dcn):
    rng = np.random.default_rng(2)
    keys = rng.integers(0, 2**63, 300_000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 300_000).astype(np.uint64)
    keys = np.concatenate([keys, keys[:1000], np.array([0, 1, 2**64 - 1], np.uint64)])  # duplicates, zero, max
    idx = dcn.Index.from_keys(keys, 31, 15)
    assert idx.header() == (31, 15, len(set(keys.tolist())))
    probe = np.concatenate([keys[::7], rng.integers(0, 2**63, 100_000, dtype=np.uint64), np.array([0, 3], np.uint64)])
    want = np.isin(probe, keys)
    assert idx.contains(probe).tolist() == want.tolist()
    empty = dcn.Index.from_keys(np.zeros(0, np.uint64), 31, 15)
    assert empty.n_keys == 0 and not empty.contains(probe[:100]).any()
    only_zero = dcn.Index.from_keys(np.zeros(3, np.uint64), 31, 15)
    assert only_zero.n_keys == 1 and only_zero.contains(np.array([0, 1], np.uint64)).tolist() == [True, False]


def test_probe_ceiling_measurement(dcn):
    """dcn_index_probe_ceiling (measurement only): a positive rate for generated and for replayed key streams, nothing
    for an empty request, and no effect on the set."""
    import torch
    keys = np.unique(np.random.default_rng(8).integers(1, 2**62, 200_000, dtype=np.uint64))
    idx = dcn.Index.from_keys(keys, 31, 15)
    assert idx.probe_ceiling(None, 1 << 20, reps=1) > 1e8
    d = torch.from_numpy(keys.view(np.int64)).cuda()
    assert idx.probe_ceiling(d.data_ptr(), d.numel(), reps=2) > 1e8
    assert idx.probe_ceiling(None, 0, reps=1) == 0.0
    assert idx.contains(keys[:1000]).all() and len(idx) == len(keys)
    idx.close()


def test_table_adversarial_keys(dcn):
    # keys that collide in the low / high bits must still be exact
    a = (np.arange(1, 50_001, dtype=np.uint64) << np.uint64(40))
    b = np.arange(1, 50_001, dtype=np.uint64)
    keys = np.concatenate([a, b])
    idx = dcn.Index.from_keys(keys, 31, 15)
    assert idx.n_keys == 100_000
    probe = np.concatenate([a[:1000], b[:1000], a[:1000] + np.uint64(1), b[:1000] + np.uint64(50_000)])
    assert idx.contains(probe).tolist() == [True] * 2000 + [False] * 2000


def test_index_from_file(oracle, dcn, tmp_path, genome):
    oidx = oracle.Index.build([genome[:50_000]], k=31, w=15)
    path = tmp_path / "g.idx"
    oidx.write(path)
    gidx = dcn.Index.from_file(str(path))
    assert gidx.header() == (31, 15, len(oidx))
    keys = oidx.keys()
    assert gidx.contains(keys).all()
    assert not gidx.contains(keys ^ np.uint64(1)).all()


def _write_index(path, keys, k=31, w=15):
    """bincode-2 varint index file (src/index.rs:130-164), written here byte by byte"""
    def varint(v):
        v = int(v)
        if v < 251:
            return bytes([v])
        if v <= 0xFFFF:
            return b"\xfb" + v.to_bytes(2, "little")
        if v <= 0xFFFFFFFF:
            return b"\xfc" + v.to_bytes(4, "little")
        return b"\xfd" + v.to_bytes(8, "little")
    with open(path, "wb") as f:
        f.write(bytes([2, k, w]) + varint(len(keys)) + b"".join(varint(x) for x in keys))


def test_index_file_encodings(dcn, tmp_path):
    """A9: the streaming device decoder (all hashes 9-byte varints) and the host decoder (anything else)
    must load the same sets; several chunk shapes, duplicates, small values, broken files."""
    rng = np.random.default_rng(77)
    big = rng.integers(1 << 32, 1 << 63, 300_000, dtype=np.uint64) | np.uint64(1 << 63) * (rng.random(300_000) < 0.5)
    big[1000:1100] = big[0:100]  # duplicates in the file are merged by the set
    for n in (1, 7, 8, 9, 64, 250, 251, 70_000, 300_000):  # 251 and 70 000: longer count varints shift every record
        path = tmp_path / f"big{n}.idx"
        _write_index(path, big[:n])
        idx = dcn.Index.from_file(str(path))
        want = np.unique(big[:n])
        assert len(idx) == len(want), n
        assert np.array_equal(np.sort(idx.keys()), want), n
    # small values use 1/3/5-byte varints: host decoder; 0 is a legal hash
    mixed = np.concatenate([big[:5000], np.array([0, 1, 250, 251, 65535, 65536, (1 << 32) - 1, 1 << 32], np.uint64)])
    path = tmp_path / "mixed.idx"
    _write_index(path, mixed)
    idx = dcn.Index.from_file(str(path))
    assert np.array_equal(np.sort(idx.keys()), np.unique(mixed))
    assert idx.contains(np.array([0, 2, 251], np.uint64)).tolist() == [True, False, True]
    # same byte count as an all-9-byte file but one marker is not 0xFD -> rejected, not misread
    path = tmp_path / "badmarker.idx"
    _write_index(path, big[:1000])
    raw = bytearray(path.read_bytes())
    assert raw[3 + 3 + 9 * 500] == 0xFD
    raw[3 + 3 + 9 * 500] = 0xFC
    path.write_bytes(bytes(raw))
    with pytest.raises(dcn.DeaconHipError):
        dcn.Index.from_file(str(path))
    # truncated file
    path = tmp_path / "short.idx"
    _write_index(path, big[:1000])
    path.write_bytes(path.read_bytes()[:-5])
    with pytest.raises(dcn.DeaconHipError):
        dcn.Index.from_file(str(path))


# ------------------------------------------------------------------------------------------------------
# K1-K3: minimizer positions and hashes
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,w", [(31, 15), (15, 11), (41, 15), (5, 5), (31, 1), (21, 9), (32, 16), (33, 15), (56, 2)])
def test_minimizer_hashes_and_positions_parity(oracle, dcn, k, w):
    rng = np.random.default_rng(k * 1000 + w)
    reads = random_reads(rng, 300, 0, 700, p_n=0.003, p_lower=0.05)
    reads += [b"", b"A", b"ACGT" * 40, b"A" * 150, b"AT" * 80, b"GGGCCC" * 30, b"N" * 100,
              b"ACGTNACGT" * 20, random_reads(rng, 1, 149, 149)[0] + b"\n", b"acgtacgtggccaattgcat" * 8]
    reads += random_reads(rng, 5, 3000, 9000, p_n=0.001)  # multi-tile
    idx = dcn.Index.from_keys(np.arange(1, 10, dtype=np.uint64), k, w)
    proc = dcn.FilterProcessor(idx, max_batch_bases=1 << 21, max_batch_reads=4096)
    b, o = oracle.concat_reads(reads)
    off, h, p = proc.minimizer_hashes_batch(b, o)
    for r, s in enumerate(reads):
        wh, wp = oracle.minimizer_hashes_and_positions(s, k, w)
        lo, hi = int(off[r]), int(off[r + 1])
        assert p[lo:hi].tolist() == wp.tolist(), (r, len(s))
        assert h[lo:hi].tolist() == wh.tolist(), (r, len(s))


def test_prefix_length_parity(oracle, dcn):
    rng = np.random.default_rng(9)
    reads = random_reads(rng, 100, 20, 400, p_n=0.002)
    idx = dcn.Index.from_keys(np.arange(1, 10, dtype=np.uint64), 31, 15)
    proc = dcn.FilterProcessor(idx, max_batch_bases=1 << 20, max_batch_reads=1024)
    b, o = oracle.concat_reads(reads)
    for pl in (6, 45, 100, 1000):
        off, h, p = proc.minimizer_hashes_batch(b, o, prefix_length=pl)
        for r, s in enumerate(reads):
            wh, wp = oracle.minimizer_hashes_and_positions(s, 31, 15, prefix_length=pl)
            lo, hi = int(off[r]), int(off[r + 1])
            assert p[lo:hi].tolist() == wp.tolist() and h[lo:hi].tolist() == wh.tolist()


def test_golden_vectors_on_gpu(dcn):
    vec = json.load(open(os.path.join(GOLDEN, "oracle_vectors.json")))["vectors"]
    procs = {}
    for v in vec:
        key = (v["k"], v["w"])
        if key not in procs:
            idx = dcn.Index.from_keys(np.arange(1, 3, dtype=np.uint64), *key)
            procs[key] = dcn.FilterProcessor(idx, max_batch_bases=1 << 16, max_batch_reads=16)
        h, p = dcn.get_minimizer_hashes_and_positions(procs[key], v["seq"].encode())
        assert [int(x) for x in p] == v["positions"]
        assert [hex(int(x)) for x in h] == v["hashes"]


def test_tile_seams(oracle, dcn, monkeypatch):
    """Tiny tiles: every seam position / carry window / dedup across tiles is exercised."""
    rng = np.random.default_rng(21)
    reads = random_reads(rng, 40, 40, 1500, p_n=0.002) + [b"ACGT" * 200, b"A" * 500, b"GCATGCAT" * 100]
    idx = dcn.Index.from_keys(np.arange(1, 10, dtype=np.uint64), 31, 15)
    b, o = oracle.concat_reads(reads)
    for tw in ("16", "17", "64", "100"):
        monkeypatch.setenv("DCN_TILE_WINDOWS", tw)
        proc = dcn.FilterProcessor(idx, max_batch_bases=1 << 20, max_batch_reads=1024)
        off, h, p = proc.minimizer_hashes_batch(b, o)
        for r, s in enumerate(reads):
            wh, wp = oracle.minimizer_hashes_and_positions(s, 31, 15)
            lo, hi = int(off[r]), int(off[r + 1])
            assert p[lo:hi].tolist() == wp.tolist(), (tw, r)
            assert h[lo:hi].tolist() == wh.tolist(), (tw, r)


# ------------------------------------------------------------------------------------------------------
# the reference's behavioural tests, on the GPU
# ------------------------------------------------------------------------------------------------------
CASES = json.load(open(os.path.join(GOLDEN, "reference_cases.json")))["cases"]


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_reference_cases_on_gpu(oracle, dcn, case):
    oidx, gidx = make_index_pair(oracle, dcn, [s.encode() for s in case["ref"]], k=case["k"], w=case["w"])
    proc = dcn.FilterProcessor(gidx, abs_threshold=case["abs"], rel_threshold=case["rel"],
                               prefix_length=case.get("prefix_length", 0), deplete=case["deplete"],
                               max_batch_bases=1 << 16, max_batch_reads=64)
    reads, uid = [], []
    for u, unit in enumerate(case["units"]):
        for s in unit:
            reads.append(s.encode())
            uid.append(u)
    keep, hits, total = check_batch(oracle, proc, oidx, reads, np.array(uid, np.uint32))
    assert keep.tolist() == case["expect_keep"]


# ------------------------------------------------------------------------------------------------------
# whole path: decisions, hits, totals
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("deplete,abs_t,rel_t", [(False, 2, 0.01), (True, 2, 0.01), (False, 1, 0.0), (True, 5, 0.3)])
def test_filter_short_reads(oracle, dcn, genome, index_pair, deplete, abs_t, rel_t):
    oidx, gidx = index_pair
    rng = np.random.default_rng(31)
    reads = sample_reads(rng, genome, 20_000, 30, 250)
    reads[5] = b""
    reads[6] = b"ACGT"
    proc = dcn.FilterProcessor(gidx, abs_threshold=abs_t, rel_threshold=rel_t, deplete=deplete,
                               max_batch_bases=1 << 23, max_batch_reads=1 << 15)
    keep, hits, total = check_batch(oracle, proc, oidx, reads)
    assert 0.2 < keep.mean() < 0.8


def test_filter_paired(oracle, dcn, genome, index_pair):
    oidx, gidx = index_pair
    rng = np.random.default_rng(32)
    reads = sample_reads(rng, genome, 10_001, 100, 151)  # odd count: last unit is a single read
    uid = (np.arange(len(reads)) // 2).astype(np.uint32)
    # identical mates: shared minimizers must be counted once
    reads[10] = reads[11] = genome[5000:5150]
    for deplete in (False, True):
        proc = dcn.FilterProcessor(gidx, deplete=deplete, max_batch_bases=1 << 22, max_batch_reads=1 << 14)
        keep, hits, total = check_batch(oracle, proc, oidx, reads, uid)
        assert total[5] == 2 * len(oracle.minimizer_hashes_and_positions(reads[10], 31, 15)[0])
        assert hits[5] * 2 <= total[5] + 1
    # triples and ragged unit sizes are legal too
    uid3 = np.repeat(np.arange(4000, dtype=np.uint32), 3)[:len(reads)]
    proc = dcn.FilterProcessor(gidx, max_batch_bases=1 << 22, max_batch_reads=1 << 14)
    check_batch(oracle, proc, oidx, reads, uid3)


def test_filter_long_reads(oracle, dcn, genome, index_pair):
    oidx, gidx = index_pair
    rng = np.random.default_rng(33)
    lens = np.clip(rng.lognormal(8.0, 0.8, 150).astype(int), 200, 60_000)
    reads = []
    for ln in lens:
        if rng.random() < 0.5:
            s = int(rng.integers(0, len(genome) - ln))
            reads.append(mutate(rng, genome[s:s + ln], 0.05))
        else:
            reads.append(random_reads(rng, 1, ln, ln)[0])
    reads += sample_reads(rng, genome, 500, 100, 200)  # mixed with short reads in the same waves
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    proc = dcn.FilterProcessor(gidx, max_batch_bases=1 << 23, max_batch_reads=1 << 12)
    keep, hits, total = check_batch(oracle, proc, oidx, reads)
    assert hits.max() > 300


def test_one_very_long_read(oracle, dcn, genome, index_pair):
    """A chromosome-sized record among short reads: tens of thousands of tiles of one unit (planned by the whole
    workgroup), read positions far beyond 2^24, a relative threshold that asks for thousands of distinct hits."""
    oidx, gidx = index_pair
    rng = np.random.default_rng(42)
    big = bytearray(random_reads(rng, 1, 24_000_000, 24_000_000)[0])
    for s in range(0, len(big) - len(genome), 3_000_000):  # copies of the indexed genome inside it -> real hits, repeated
        big[s:s + len(genome)] = genome
    big[5_000_000:5_000_200] = b"N" * 200
    reads = sample_reads(rng, genome, 200, 100, 200) + [bytes(big)] + sample_reads(rng, genome, 200, 100, 200)
    for rel in (0.0, 0.002, 0.5):
        proc = dcn.FilterProcessor(gidx, abs_threshold=2, rel_threshold=rel, max_batch_bases=len(big) + (1 << 20),
                                   max_batch_reads=1 << 10)
        keep, hits, total = check_batch(oracle, proc, oidx, reads)
        assert total[200] > 2_000_000 and hits[200] > 10_000
    assert keep[200] == 0  # -r 0.5: half of its minimizers would have to hit


def test_very_long_reads_with_few_hits(oracle, dcn, genome, index_pair):
    """Reads of more than 1,024 tiles with a handful of hits (a foreign chromosome carrying a few k-mers of the indexed
    genome, some of them repeated): too few hits for a global set by their number, too many tiles for one wave to walk
    -- they take the multi-wave pass of the distinct count (plan.hip, DCN_LDS_WALK_MAX_TILES)."""
    oidx, gidx = index_pair
    rng = np.random.default_rng(43)
    reads = []
    for n, planted in ((300_000, 3), (700_000, 1), (1_500_000, 40), (290_000, 0)):
        big = bytearray(random_reads(rng, 1, n, n)[0])
        spots = rng.integers(0, n - 400, planted)
        for i, s in enumerate(spots):
            g0 = 5_000 if i % 2 else int(rng.integers(0, len(genome) - 200))  # every second copy is the same stretch
            big[s:s + 120] = genome[g0:g0 + 120]
        reads.append(bytes(big))
    reads += sample_reads(rng, genome, 100, 100, 5_000)
    proc = dcn.FilterProcessor(gidx, abs_threshold=2, rel_threshold=0.0, max_batch_bases=4 << 20, max_batch_reads=1 << 10)
    keep, hits, total = check_batch(oracle, proc, oidx, reads)
    assert 1 <= hits[0] < 40 and 1 <= hits[1] < 20 and hits[3] == 0 and total[2] > 100_000


def test_repeats_inside_long_reads_are_counted_once(oracle, dcn, genome, index_pair):
    oidx, gidx = index_pair
    seg = genome[1000:3000]
    reads = [seg * 4, seg + revcomp(seg) + seg, genome[0:8000] + genome[4000:12000]]
    proc = dcn.FilterProcessor(gidx, max_batch_bases=1 << 20, max_batch_reads=64)
    keep, hits, total = check_batch(oracle, proc, oidx, reads)
    assert total[0] > 3 * hits[0]


def test_low_complexity_and_list_overflow(oracle, dcn):
    """Periodic reads emit a position for nearly every window: per-lane lists and the LDS hit buffer overflow."""
    units = [b"ACGT" * 60, b"A" * 250, b"AT" * 120, b"GGGCCC" * 40, b"GCATGCAT" * 35, b"ACGTTGCA" * 30]
    oidx, gidx = make_index_pair(oracle, dcn, units)
    rng = np.random.default_rng(34)
    reads = []
    for _ in range(3000):
        u = units[int(rng.integers(0, len(units)))]
        s = int(rng.integers(0, 20))
        reads.append(u[s:s + int(rng.integers(60, 220))])
    proc = dcn.FilterProcessor(gidx, abs_threshold=1, max_batch_bases=1 << 21, max_batch_reads=1 << 12)
    keep, hits, total = check_batch(oracle, proc, oidx, reads)
    assert total.max() > 100


def test_all_host_reads_fill_the_hit_buffer(oracle, dcn, genome, index_pair):
    oidx, gidx = index_pair
    rng = np.random.default_rng(35)
    reads = sample_reads(rng, genome, 4096, 240, 250, host_frac=1.0, sub=0.0, p_n=0.0)
    proc = dcn.FilterProcessor(gidx, max_batch_bases=1 << 21, max_batch_reads=1 << 12)
    keep, hits, total = check_batch(oracle, proc, oidx, reads)
    assert hits.min() >= 10 and keep.all()


def test_record_scratch_grows(oracle, dcn, genome, index_pair, monkeypatch):
    oidx, gidx = index_pair
    # long host reads: ~1 hit per 8 bases, i.e. more than the 2800 the distinct pass counts in LDS: such units take
    # a global set, and the context is created with (almost) no room for those
    monkeypatch.setenv("DCN_RECORD_CAPACITY", "64")
    reads = [genome[i * 20_000:(i + 1) * 20_000 + 5000] for i in range(9)]
    proc = dcn.FilterProcessor(gidx, max_batch_bases=1 << 18, max_batch_reads=16)
    keep, hits, total = check_batch(oracle, proc, oidx, reads)
    assert hits.min() > 2800
    s = proc.stats()
    assert s["total_seqs"] == 3 * len(reads)  # check_batch makes three calls; the overflowed attempt is not counted


def test_empty_and_ragged_batches(oracle, dcn, index_pair):
    oidx, gidx = index_pair
    proc = dcn.FilterProcessor(gidx, max_batch_bases=1 << 16, max_batch_reads=256)
    keep, hits, total = proc.filter_reads([])
    assert len(keep) == 0
    reads = [b"", b"", b"A", b"ACGT" * 7, b"", b"ACGT" * 11, b"N" * 80, b""]
    for deplete in (False, True):
        proc.deplete = deplete
        keep, hits, total = check_batch(oracle, proc, oidx, reads)
        assert total.tolist()[:5] == [0] * 5 and keep.tolist() == [deplete] * len(reads)
    with pytest.raises(dcn.DeaconHipError):
        proc.filter_reads([b"A" * 100] * 300)  # more reads than the context allows
    with pytest.raises(dcn.DeaconHipError):
        proc.filter_batch(np.zeros(10, np.uint8), np.array([0, 5, 3], np.uint64))  # offsets not monotone


def test_stats_counters(oracle, dcn, genome, index_pair):
    oidx, gidx = index_pair
    rng = np.random.default_rng(36)
    reads = sample_reads(rng, genome, 3000, 50, 200)
    proc = dcn.FilterProcessor(gidx, deplete=True, max_batch_bases=1 << 21, max_batch_reads=1 << 12)
    keep1, _, _ = proc.filter_reads(reads)
    uid = (np.arange(len(reads)) // 2).astype(np.uint32)
    b, o = oracle.concat_reads(reads)
    keep2, _, _ = proc.filter_batch(b, o, uid)
    lens = np.array([len(r) for r in reads])
    plen = np.add.reduceat(lens, np.arange(0, len(reads), 2))
    s = proc.stats()
    assert s["total_seqs"] == 2 * len(reads)
    assert s["total_bp"] == 2 * lens.sum()
    assert s["output_bp"] == lens[keep1].sum() + plen[keep2].sum()
    assert s["filtered_bp"] == lens[~keep1].sum() + plen[~keep2].sum()
    assert s["filtered_seqs"] == (~keep1).sum() + 2 * (~keep2).sum()
    assert s["output_seq_counter"] == keep1.sum() + 2 * keep2.sum()
    summ = proc.summary(2.0)
    assert summ["seqs_in"] == s["total_seqs"] and summ["bp_per_second"] == s["total_bp"] // 2
    proc.reset_stats()
    assert sum(proc.stats().values()) == 0


def test_should_keep_hashes_seam(oracle, dcn, genome, index_pair):
    oidx, gidx = index_pair
    rng = np.random.default_rng(37)
    reads = sample_reads(rng, genome, 2000, 50, 3000)
    hs = [oracle.minimizer_hashes_and_positions(r, 31, 15)[0] for r in reads]
    off = np.concatenate([[0], np.cumsum([len(h) for h in hs])]).astype(np.uint64)
    flat = np.concatenate(hs)
    proc = dcn.FilterProcessor(gidx, max_batch_bases=1 << 16, max_batch_reads=1 << 12)
    for deplete in (False, True):
        want = oracle.should_keep_hashes(oidx, flat, off, 2, 0.01, deplete)
        proc.deplete = deplete
        got = proc.should_keep_hashes(flat, off)
        for g, w_ in zip(got, want):
            assert g.tolist() == w_.tolist()
    res = dcn.unpaired_should_keep(proc, hs[:50], 2, 0.01, False)
    want = oracle.should_keep_hashes(oidx, np.concatenate(hs[:50]), off[:51], 2, 0.01, False)
    assert [r[0] for r in res] == want[0].tolist() and [r[1] for r in res] == want[1].tolist()
    pairs = [np.concatenate([hs[2 * i], hs[2 * i + 1]]) for i in range(25)]
    res = dcn.paired_should_keep(proc, pairs, 2, 0.01, True)
    poff = np.concatenate([[0], np.cumsum([len(x) for x in pairs])]).astype(np.uint64)
    want = oracle.should_keep_hashes(oidx, np.concatenate(pairs), poff, 2, 0.01, True)
    assert [r[0] for r in res] == want[0].tolist() and [r[2] for r in res] == want[2].tolist()


def test_single_read_seam(oracle, dcn, genome, index_pair):
    oidx, gidx = index_pair
    proc = dcn.FilterProcessor(gidx, max_batch_bases=1 << 16, max_batch_reads=16)
    r = genome[7000:7150]
    keep, hits, total = proc.should_keep_sequence(r)
    assert (keep, hits) == (True, total) and total > 5
    assert proc.should_keep_sequence(revcomp(r)) == (keep, hits, total)
    kp, hp, tp = proc.should_keep_pair(r, revcomp(r))
    assert (kp, hp, tp) == (True, hits, 2 * total)


def test_device_resident_inputs(oracle, dcn, genome, index_pair):
    torch = pytest.importorskip("torch")
    oidx, gidx = index_pair
    rng = np.random.default_rng(38)
    reads = sample_reads(rng, genome, 5000, 100, 151)
    b, o = oracle.concat_reads(reads)
    uid = (np.arange(len(reads)) // 2).astype(np.uint32)
    proc = dcn.FilterProcessor(gidx, deplete=True, max_batch_bases=1 << 21, max_batch_reads=1 << 13)
    dev = torch.device("cuda:0")
    d_b = torch.from_numpy(b).to(dev)
    d_o = torch.from_numpy(o.view(np.int64)).to(dev)
    d_u = torch.from_numpy(uid.view(np.int32)).to(dev)
    for unit, n_units in ((None, len(reads)), (d_u, int(uid[-1]) + 1)):
        d_keep = torch.zeros(n_units, dtype=torch.uint8, device=dev)
        d_hits = torch.zeros(n_units, dtype=torch.int32, device=dev)
        d_total = torch.zeros(n_units, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        proc.filter_batch_device(d_b.data_ptr(), d_o.data_ptr(), len(reads), len(b), d_keep.data_ptr(),
                                 d_hits.data_ptr(), d_total.data_ptr(),
                                 d_unit_id=unit.data_ptr() if unit is not None else None, n_units=n_units)
        proc.synchronize()
        want = oracle.filter_batch(oidx, b, o, None if unit is None else uid, deplete=True, threads=4)
        assert d_keep.cpu().numpy().astype(bool).tolist() == want[0].tolist()
        assert d_hits.cpu().numpy().tolist() == want[1].tolist()
        assert d_total.cpu().numpy().tolist() == want[2].tolist()


# ------------------------------------------------------------------------------------------------------
# size-independent properties at a larger size (no oracle pass over the full input)
# ------------------------------------------------------------------------------------------------------
def test_page_locked_and_large_pageable_batches(oracle, dcn, genome, index_pair):
    """dcn_filter_batch copies page-locked buffers directly and splits large pageable ones over the host
    copy threads (several staging rounds): both must give the oracle's answers."""
    oidx, gidx = index_pair
    rng = np.random.default_rng(40)
    reads = sample_reads(rng, genome, 50_000, 140, 160) * 8  # ~60 MB: two staging rounds, above the pool threshold
    bases, offsets = oracle.concat_reads(reads)
    proc = dcn.FilterProcessor(gidx, max_batch_bases=len(bases), max_batch_reads=len(reads))
    want = oracle.filter_batch(oidx, bases, offsets, threads=8)
    got = proc.filter_batch(bases, offsets)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)
    pb = dcn.PinnedBuffer(len(bases), np.uint8)
    po = dcn.PinnedBuffer(len(offsets), np.uint64)
    pb.array[:] = bases
    po.array[:] = offsets
    got = proc.filter_batch(pb.array, po.array)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)
    # a slice of a page-locked allocation is still page-locked
    n = 1000
    got = proc.filter_batch(pb.array[:int(offsets[n])], po.array[:n + 1])
    assert np.array_equal(got[0], want[0][:n]) and np.array_equal(got[1], want[1][:n])
    pb.close()
    po.close()


@pytest.mark.parametrize("abs_t,rel_t,deplete", [(1, 0.0, False), (2, 0.01, False), (2, 0.01, True), (3, 0.05, True),
                                                 (4, 0.0, False), (5, 0.0, False), (2, 0.2, False), (1, 1.0, True)])
def test_decisions_only_mode(oracle, dcn, genome, index_pair, abs_t, rel_t, deplete):
    """dcn_filter_batch with hits = total = NULL: the early-out path (abs 1..4 and a relative threshold that cannot
    raise the requirement for a list of that length) and its fall-backs (abs 5, large -r, reads with N runs whose
    valid-minimizer total is smaller than the list, repeats whose hits are not distinct, pairs, long reads)."""
    oidx, gidx = index_pair
    rng = np.random.default_rng(41)
    reads = sample_reads(rng, genome, 6000, 31, 300, p_n=0.01)
    rep = genome[7000:7040]
    reads[10] = rep * 5                      # the same few minimizers again and again: distinct hits stay low
    reads[11] = rep + b"N" * 40 + rep        # hits on both sides of an invalid stretch
    reads[12] = genome[9000:9045]            # exactly one window
    reads[13] = b"N" * 100
    reads[14] = genome[100:5100]             # long read: several tiles, exact path
    proc = dcn.FilterProcessor(gidx, abs_threshold=abs_t, rel_threshold=rel_t, deplete=deplete,
                               max_batch_bases=1 << 22, max_batch_reads=1 << 13)
    b, o = oracle.concat_reads(reads)
    want = oracle.filter_batch(oidx, b, o, None, abs_t, rel_t, 0, deplete, threads=4)[0]
    assert proc.filter_batch(b, o, counts=False).tolist() == want.tolist()
    assert 0 < want.sum() < len(want)
    uid = (np.arange(len(reads)) // 2).astype(np.uint32)
    want = oracle.filter_batch(oidx, b, o, uid, abs_t, rel_t, 0, deplete, threads=4)[0]
    assert proc.filter_batch(b, o, uid, counts=False).tolist() == want.tolist()


def test_properties_at_scale(oracle, dcn, genome, index_pair):
    oidx, gidx = index_pair
    rng = np.random.default_rng(39)
    n = 400_000
    g = np.frombuffer(genome, dtype=np.uint8)
    starts = rng.integers(0, len(genome) - 150, n)
    host = rng.random(n) < 0.5
    mat = g[starts[:, None] + np.arange(150)[None, :]].copy()
    rnd = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 150))]
    mat[~host] = rnd[~host]
    bases = mat.reshape(-1)
    offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
    proc = dcn.FilterProcessor(gidx, max_batch_bases=n * 150, max_batch_reads=n)
    keep, hits, total = proc.filter_batch(bases, offsets)
    # idempotence
    keep2, hits2, total2 = proc.filter_batch(bases, offsets)
    if not ((keep == keep2).all() and (hits == hits2).all() and (total == total2).all()):
        full = oracle.filter_batch(oidx, bases, offsets, threads=8)
        msg = []
        for name, got in (("call1", (keep, hits, total)), ("call2", (keep2, hits2, total2))):
            for j, what in enumerate(("keep", "hits", "total")):
                d = np.nonzero(got[j] != full[j])[0]
                msg.append(f"{name}.{what}: {len(d)} differ from the oracle" + (f", units {d[0]}..{d[-1]}, got {got[j][d[:4]]} want {full[j][d[:4]]}" if len(d) else ""))
        raise AssertionError("two calls on the same batch disagree: " + "; ".join(msg))
    # strand symmetry: the reverse complement of every read has the same minimizer multiset
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    rc = comp[mat[:, ::-1]].reshape(-1)
    keep3, hits3, total3 = proc.filter_batch(rc, offsets)
    assert (hits == hits3).all() and (total == total3).all() and (keep == keep3).all()
    # batch-split invariance: any split of the batch gives the same per-read results
    cut = 123_457
    a = proc.filter_batch(bases[:cut * 150], offsets[:cut + 1])
    b = proc.filter_batch(bases[cut * 150:], offsets[cut:] - offsets[cut])
    assert (np.concatenate([a[1], b[1]]) == hits).all() and (np.concatenate([a[0], b[0]]) == keep).all()
    # pairing: a pair's total is the sum of its mates' totals, its hits at most the sum and at least the max
    uid = (np.arange(n) // 2).astype(np.uint32)
    kp, hp, tp = proc.filter_batch(bases, offsets, uid)
    assert (tp == total[0::2] + total[1::2]).all()
    assert (hp <= hits[0::2] + hits[1::2]).all() and (hp >= np.maximum(hits[0::2], hits[1::2])).all()
    # host reads are (almost all) hit, random reads are not; and a sampled slice matches the oracle exactly
    assert hits[host].mean() > 8 and hits[~host].max() <= 1
    sl = slice(200_000, 203_000)
    want = oracle.filter_batch(oidx, bases[sl.start * 150:sl.stop * 150], offsets[:3001], threads=4)
    assert want[1].tolist() == hits[sl].tolist() and want[2].tolist() == total[sl].tolist()


def test_mixed_long_and_short_stream_properties(oracle, dcn, genome, index_pair, monkeypatch):
    """BASELINE configs[4]'s stream shape: long and short reads interleaved in one batch without unit ids.  The planner
    puts the tiles of multi-tile reads ahead of the single-tile reads inside each planning block (round 3), so the tile
    order is no longer the read order: per-read results must not notice -- equal to filtering the long and the short
    reads as two separate batches, equal for another tile size, equal in decisions-only mode, and (on a slice) equal to
    the oracle."""
    oidx, gidx = index_pair
    rng = np.random.default_rng(61)
    reads, is_long = [], []
    while sum(len(r) for r in reads) < 40_000_000:
        ln = int(min(150_000, max(300, rng.lognormal(8.9, 0.8))))
        if rng.random() < 0.5:
            s = int(rng.integers(0, len(genome) - 1)) % max(1, len(genome) - ln) if ln < len(genome) else 0
            seg = (genome * (ln // len(genome) + 2))[s:s + ln]
            reads.append(mutate(rng, seg, 0.05))
        else:
            reads.append(random_reads(rng, 1, ln, ln)[0])
        is_long.append(True)
        short = sample_reads(rng, genome, max(1, ln // 150), 150, 150)
        reads += short
        is_long += [False] * len(short)
    is_long = np.array(is_long)
    b, o = oracle.concat_reads(reads)
    proc = dcn.FilterProcessor(gidx, max_batch_bases=len(b) + 64, max_batch_reads=len(reads) + 1)
    keep, hits, total = proc.filter_batch(b, o)
    assert keep[is_long].any() and keep[~is_long].any() and (~keep)[is_long].any()
    # the two kinds as separate batches
    for sel in (is_long, ~is_long):
        sub = [r for r, f in zip(reads, sel) if f]
        sb, so = oracle.concat_reads(sub)
        k2, h2, t2 = proc.filter_batch(sb, so)
        assert (k2 == keep[sel]).all() and (h2 == hits[sel]).all() and (t2 == total[sel]).all()
    assert (proc.filter_batch(b, o, counts=False) == keep).all()          # decisions only (the distinct pass may stop early)
    proc.close()
    monkeypatch.setenv("DCN_TILE_WINDOWS", "64")                           # many more multi-tile reads, other wave seams
    proc = dcn.FilterProcessor(gidx, max_batch_bases=len(b) + 64, max_batch_reads=len(reads) + 1)
    k3, h3, t3 = proc.filter_batch(b, o)
    assert (k3 == keep).all() and (h3 == hits).all() and (t3 == total).all()
    proc.close()
    n = int(np.searchsorted(o, 3_000_000))
    want = oracle.filter_batch(oidx, b[:int(o[n])], o[:n + 1], threads=4)
    assert want[0].tolist() == keep[:n].tolist() and want[1].tolist() == hits[:n].tolist() and want[2].tolist() == total[:n].tolist()


def test_units_of_many_reads_cut_by_a_planning_block(oracle, dcn, genome, index_pair):
    """unit_id may group any number of consecutive reads.  A unit whose reads fall into two planning blocks (256 reads
    each) has no contiguous tile range: the scan cannot finish it in-wave, and the distinct pass finds its tiles by
    sweeping all of them -- with an LDS-sized hit count (never: such units always take a global set) and with
    thousands of hits (long exact reads)."""
    oidx, gidx = index_pair
    rng = np.random.default_rng(60)
    reads = sample_reads(rng, genome, 254, 40, 200)
    uid = list(range(254))
    # unit 254: five long reads from the genome, reads 254..258 -> cut by the block boundary at read 256
    for j in range(5):
        reads.append(genome[j * 30_000:j * 30_000 + 25_000])
        uid.append(254)
    # unit 255: three short reads right after it; unit 256..: singles again, and one more cut unit at the next boundary
    more = sample_reads(rng, genome, 253 + 40, 40, 200)
    reads += more
    uid += [255, 255, 255] + list(range(256, 256 + 248)) + [600] * 6 + list(range(601, 601 + 36))
    uid = np.array(uid, dtype=np.uint32)
    uid = np.cumsum(np.concatenate([[0], (np.diff(uid) != 0).astype(np.uint32)])).astype(np.uint32)  # 0,1,2,... without gaps
    assert len(uid) == len(reads)
    for deplete, abs_t in ((False, 2), (True, 40)):
        proc = dcn.FilterProcessor(gidx, abs_threshold=abs_t, deplete=deplete, max_batch_bases=1 << 20, max_batch_reads=1 << 10)
        keep, hits, total = check_batch(oracle, proc, oidx, reads, uid)
        assert hits[254] > 5000  # the cut unit went through the global-set path
        proc.close()
